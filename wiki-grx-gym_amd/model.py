"""Robot model: URDF-derived link table (assets/*.model.json) -> lumped dynamics model.

Role of ``gym.load_asset`` + the asset options of the reference (legged_robot.py:947-966,
legged_robot_config.py:117-131): fixed joints are kept as named *frames* (collapse_fixed_joints
False keeps all 37 bodies addressable by name) but, dynamically, every fixed-joint subtree is
merged into its moving ancestor (exact for rigid attachments).  Primitive collision shapes become
sphere sets (DESIGN.md "contact geometry").

Everything is float64 numpy here; ``fill_model`` narrows to the float32 C struct once.
"""
import json
import os

import numpy as np

from . import _capi

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

# reference asset path -> asset key (legged_gym/resources/robots/<R>/urdf/<file>.urdf)
_URDF_TO_KEY = {
    "GR1T1_lower_limb.urdf": "gr1t1_lower_limb",
    "GR1T1.urdf": "gr1t1",
    "GR1T2_lower_limb.urdf": "gr1t2_lower_limb",
    "GR1T2.urdf": "gr1t2",
}


def asset_key_from_file(path):
    base = os.path.basename(path)
    if base not in _URDF_TO_KEY:
        raise ValueError(f"no model table for asset '{path}' (known: {sorted(_URDF_TO_KEY)})")
    return _URDF_TO_KEY[base]


def rpy_to_matrix(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _sym(i6):
    xx, xy, xz, yy, yz, zz = i6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], dtype=np.float64)


def _six(I):
    return np.array([I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]])


def combine(parts):
    """parts: list of (mass, com(3), Ic(3x3 about com)) in one frame -> merged (m, com, Ic)."""
    M = sum(p[0] for p in parts)
    if M <= 0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    c = sum(p[0] * p[1] for p in parts) / M
    I = np.zeros((3, 3))
    for m, ci, Ii in parts:
        d = ci - c
        I += Ii + m * (d @ d * np.eye(3) - np.outer(d, d))
    return M, c, I


class RobotModel:
    """Lumped model + named frames.  Attributes mirror what the reference reads from the asset:
    ``body_names`` (37), ``dof_names``, per-DOF limits, plus the merged dynamics tables."""

    def __init__(self, key):
        with open(os.path.join(ASSET_DIR, key + ".model.json")) as f:
            raw = json.load(f)
        self.key = key
        self.raw = raw
        links = raw["links"]
        self.body_names = raw["body_names"]
        self.dof_names = raw["dof_names"]
        self.num_links = len(links)
        self.num_dofs = len(self.dof_names)
        nb = 1 + self.num_dofs
        if nb > _capi.MAX_BODIES:
            raise ValueError("too many moving bodies")
        self.num_bodies = nb

        moving_of = [0] * self.num_links       # link -> moving body index
        R_rel = [np.eye(3)] * self.num_links   # link frame -> moving body frame
        p_rel = [np.zeros(3)] * self.num_links
        self.parent = [-1] * nb
        self.joint_axis = np.zeros((nb, 3))
        self.joint_rot0 = np.tile(np.eye(3), (nb, 1, 1))
        self.joint_pos = np.zeros((nb, 3))
        self.dof_lower = np.zeros(self.num_dofs)
        self.dof_upper = np.zeros(self.num_dofs)
        self.dof_vel_limit = np.zeros(self.num_dofs)
        self.dof_effort = np.zeros(self.num_dofs)
        parts = [[] for _ in range(nb)]
        base_link_part = None
        nbody = 1
        for li, L in enumerate(links):
            if L["joint_type"] == "floating":
                mb, R, p = 0, np.eye(3), np.zeros(3)
            else:
                pl = L["parent"]
                Rj = rpy_to_matrix(L["origin_rpy"])
                pj = np.array(L["origin_xyz"], dtype=np.float64)
                if L["joint_type"] == "fixed":
                    mb = moving_of[pl]
                    R = R_rel[pl] @ Rj
                    p = p_rel[pl] + R_rel[pl] @ pj
                else:
                    if L["joint_type"] not in ("revolute", "continuous"):
                        raise ValueError(f"unsupported joint type {L['joint_type']}")
                    mb = nbody
                    nbody += 1
                    self.parent[mb] = moving_of[pl]
                    self.joint_rot0[mb] = R_rel[pl] @ Rj
                    self.joint_pos[mb] = p_rel[pl] + R_rel[pl] @ pj
                    ax = np.array(L["axis"], dtype=np.float64)
                    self.joint_axis[mb] = ax / np.linalg.norm(ax)
                    lim = L.get("limit", {})
                    d = mb - 1
                    self.dof_lower[d] = lim.get("lower", -np.pi)
                    self.dof_upper[d] = lim.get("upper", np.pi)
                    self.dof_vel_limit[d] = lim.get("velocity", 100.0)
                    self.dof_effort[d] = lim.get("effort", 0.0)
                    R, p = np.eye(3), np.zeros(3)
            moving_of[li], R_rel[li], p_rel[li] = mb, R, p
            if L["mass"] > 0:
                Ri = R @ rpy_to_matrix(L["com_rpy"])
                part = (L["mass"], p + R @ np.array(L["com"]), Ri @ _sym(L["inertia"]) @ Ri.T)
                if li == 0:
                    base_link_part = part
                else:
                    parts[mb].append(part)
        assert nbody == nb
        self.link_body = moving_of
        self.link_rot = R_rel
        self.link_pos = p_rel

        self.base_link = base_link_part if base_link_part else (0.0, np.zeros(3), np.zeros((3, 3)))
        self.base_rest = combine(parts[0])
        parts[0] = parts[0] + ([base_link_part] if base_link_part else [])
        self.mass = np.zeros(nb)
        self.com = np.zeros((nb, 3))
        self.inertia = np.zeros((nb, 3, 3))
        for b in range(nb):
            self.mass[b], self.com[b], self.inertia[b] = combine(parts[b])

        # collision primitives -> spheres
        self.spheres = []  # (body, pos(3), radius, link index)
        for li, L in enumerate(links):
            for shp in L["collisions"]:
                Rs = R_rel[li] @ rpy_to_matrix(shp["rpy"])
                ps = p_rel[li] + R_rel[li] @ np.array(shp["xyz"])
                r = shp["radius"] if "radius" in shp else None
                if shp["type"] == "sphere":
                    self.spheres.append((moving_of[li], ps, r, li))
                elif shp["type"] == "cylinder":
                    half = 0.5 * shp["length"] - r
                    axis = Rs[:, 2]
                    if half <= 1e-6:
                        self.spheres.append((moving_of[li], ps, r, li))
                    else:
                        self.spheres.append((moving_of[li], ps - half * axis, r, li))
                        self.spheres.append((moving_of[li], ps + half * axis, r, li))
                else:
                    raise ValueError(f"unsupported collision primitive {shp['type']}")
        if len(self.spheres) > _capi.MAX_SPHERES:
            raise ValueError("too many collision spheres")
        # links that can touch each other (tools/self_collision_pairs.py: Monte-Carlo over the joint ranges)
        self.self_collision_link_pairs = [tuple(p) for p in raw.get("self_collision_link_pairs", [])]

    # ---- name queries, the way the reference builds its index sets (gr1t1.py:18-113, 127-279)
    def links_containing(self, sub):
        return [i for i, n in enumerate(self.body_names) if sub in n]

    def dofs_containing(self, sub):
        return [i for i, n in enumerate(self.dof_names) if sub in n]

    def total_mass(self):
        return float(self.mass.sum())


def _mask(idx):
    m = 0
    for i in idx:
        m |= 1 << i
    return m


def fill_model(cm, rm, foot_name, torso_name, forehead_name, terminate_names, penalise_names, damp_alpha=0.5, sim_dt=0.002, armature=0.0):
    """Write RobotModel ``rm`` into the ctypes ``_capi.Model`` ``cm``."""
    nb = rm.num_bodies
    cm.num_bodies = nb
    for b in range(nb):
        cm.parent[b] = rm.parent[b]
        for k in range(3):
            cm.joint_axis[b][k] = rm.joint_axis[b][k]
            cm.joint_pos[b][k] = rm.joint_pos[b][k]
            cm.com[b][k] = rm.com[b][k]
        for k in range(9):
            cm.joint_rot0[b][k] = rm.joint_rot0[b].reshape(-1)[k]
        cm.mass[b] = rm.mass[b]
        s6 = _six(rm.inertia[b])
        for k in range(6):
            cm.inertia[b][k] = s6[k]
    for tag, part in (("base_link", rm.base_link), ("base_rest", rm.base_rest)):
        setattr(cm, tag + "_mass", part[0])
        for k in range(3):
            getattr(cm, tag + "_com")[k] = part[1][k]
        s6 = _six(part[2])
        for k in range(6):
            getattr(cm, tag + "_inertia")[k] = s6[k]
    for d in range(rm.num_dofs):
        cm.dof_lower[d] = rm.dof_lower[d]
        cm.dof_upper[d] = rm.dof_upper[d]
        cm.dof_vel_limit[d] = rm.dof_vel_limit[d]
        cm.dof_effort[d] = rm.dof_effort[d]
        cm.dof_armature[d] = float(armature[d]) if hasattr(armature, "__len__") else float(armature)

    feet = rm.links_containing(foot_name)
    if len(feet) != 2:
        raise ValueError(f"expected 2 links containing '{foot_name}', found {len(feet)}")
    term_links = set()
    for n in terminate_names:
        term_links.update(rm.links_containing(n))
    pen_links = set()
    for n in penalise_names:
        pen_links.update(rm.links_containing(n))
    # spheres sorted by (body, link) so that per-link netting sees each link's shapes contiguously
    order = sorted(range(len(rm.spheres)), key=lambda i: (rm.spheres[i][0], rm.spheres[i][3], i))
    cm.num_spheres = len(order)
    for k, i in enumerate(order):
        body, pos, rad, li = rm.spheres[i]
        cm.sph_body[k] = body
        for a in range(3):
            cm.sph_pos[k][a] = pos[a]
        cm.sph_radius[k] = rad
        fl = 0
        if li == feet[0]:
            fl |= _capi.SPH_FOOT_LEFT
        if li == feet[1]:
            fl |= _capi.SPH_FOOT_RIGHT
        if li in term_links:
            fl |= _capi.SPH_TERMINATE
        if li in pen_links:
            fl |= _capi.SPH_PENALISE
        cm.sph_flags[k] = fl
        cm.sph_link[k] = li
        # effective mass of the carrying body at the sphere centre along body z (the sole normal)
        arm = np.cross(pos - rm.com[body], np.array([0.0, 0.0, 1.0]))
        inv_meff = 1.0 / rm.mass[body] + arm @ np.linalg.solve(rm.inertia[body], arm)
        cm.sph_damp_max[k] = damp_alpha / inv_meff / sim_dt
    # self-collision: every sphere pair of every link pair that can touch
    link_of = [rm.spheres[i][3] for i in order]
    pairs = [(a, b) for a in range(len(order)) for b in range(a + 1, len(order))
             if (min(link_of[a], link_of[b]), max(link_of[a], link_of[b])) in set(rm.self_collision_link_pairs)]
    if len(pairs) > _capi.MAX_PAIRS:
        raise ValueError(f"{len(pairs)} self-collision sphere pairs exceed GRX_MAX_PAIRS")
    cm.num_pairs = len(pairs)
    for k, (a, b) in enumerate(pairs):
        cm.pair_a[k], cm.pair_b[k] = a, b
    for f in range(2):
        cm.foot_body[f] = rm.link_body[feet[f]]
        for a in range(3):
            cm.foot_pos[f][a] = rm.link_pos[feet[f]][a]

    def frame(name, body_attr, rot_attr):
        ls = rm.links_containing(name) if name else []
        if ls:
            setattr(cm, body_attr, rm.link_body[ls[0]])
            R = rm.link_rot[ls[0]].reshape(-1)
        else:
            setattr(cm, body_attr, -1)
            R = np.eye(3).reshape(-1)
        for k in range(9):
            getattr(cm, rot_attr)[k] = R[k]

    frame(torso_name, "torso_body", "torso_rot")
    frame(forehead_name, "forehead_body", "forehead_rot")
    # every URDF link frame (GRX_T_RIGID_BODY_STATES rows)
    if rm.num_links > _capi.MAX_LINKS:
        raise ValueError(f"{rm.num_links} links > GRX_MAX_LINKS")
    cm.num_links = rm.num_links
    for li in range(rm.num_links):
        cm.link_body[li] = int(rm.link_body[li])
        for a in range(3):
            cm.link_pos[li][a] = float(rm.link_pos[li][a])
        R = np.asarray(rm.link_rot[li]).reshape(-1)
        for a in range(9):
            cm.link_rot[li][a] = float(R[a])
    return {"feet_links": feet, "termination_links": sorted(term_links), "penalised_links": sorted(pen_links)}
