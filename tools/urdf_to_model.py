#!/usr/bin/env python3
"""Extract a compact robot model table (JSON) from the GR1T1 / GR1T2 URDF files.

Run ONCE in the build container (the reference tree does not exist on the GPU box):

    python tools/urdf_to_model.py            # reads /root/reference/.../urdf/*.urdf
                                             # writes wiki-grx-gym_amd/assets/*.model.json

The JSON is *data* (names, parents, joint frames, limits, inertials, primitive collision
shapes) -- the robot description the reference loads through gym.load_asset
(legged_robot.py:947-966).  Meshes are visual-only in these URDFs (every mesh <collision>
is commented out) and are ignored.

Body order = depth-first from the root link with siblings visited in ASCII name order
("left leg, right leg, waist, head, left arm, right arm" in
gr1t1_config.py:284-299 is reproduced by exactly that order).  DOF order = order of the
non-fixed joints in that traversal.
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

REF = "/root/reference/legged_gym/resources/robots"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "wiki-grx-gym_amd", "assets")

MODELS = {
    "gr1t1_lower_limb": "GR1T1/urdf/GR1T1_lower_limb.urdf",
    "gr1t1": "GR1T1/urdf/GR1T1.urdf",
    "gr1t2_lower_limb": "GR1T2/urdf/GR1T2_lower_limb.urdf",
    "gr1t2": "GR1T2/urdf/GR1T2.urdf",
}


def _vec(s, n=3):
    if s is None:
        return [0.0] * n
    v = [float(x) for x in s.split()]
    assert len(v) == n, s
    return v


def parse(path):
    root = ET.parse(path).getroot()
    links = {}
    for l in root.findall("link"):
        name = l.get("name")
        rec = {"name": name, "mass": 0.0, "com": [0.0] * 3, "com_rpy": [0.0] * 3,
               "inertia": [0.0] * 6, "collisions": []}
        ine = l.find("inertial")
        if ine is not None:
            o = ine.find("origin")
            if o is not None:
                rec["com"] = _vec(o.get("xyz"))
                rec["com_rpy"] = _vec(o.get("rpy"))
            rec["mass"] = float(ine.find("mass").get("value"))
            I = ine.find("inertia")
            rec["inertia"] = [float(I.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")]
        for c in l.findall("collision"):
            g = list(c.find("geometry"))[0]
            o = c.find("origin")
            shape = {"type": g.tag,
                     "xyz": _vec(o.get("xyz")) if o is not None else [0.0] * 3,
                     "rpy": _vec(o.get("rpy")) if o is not None else [0.0] * 3}
            if g.tag == "cylinder":
                shape["radius"] = float(g.get("radius"))
                shape["length"] = float(g.get("length"))
            elif g.tag == "sphere":
                shape["radius"] = float(g.get("radius"))
            elif g.tag == "box":
                shape["size"] = _vec(g.get("size"))
            else:  # mesh collision: none are active in these URDFs
                continue
            rec["collisions"].append(shape)
        links[name] = rec

    children = {}
    has_parent = set()
    for j in root.findall("joint"):
        o = j.find("origin")
        ax = j.find("axis")
        lim = j.find("limit")
        p = j.find("parent").get("link")
        c = j.find("child").get("link")
        jr = {"joint_name": j.get("name"), "joint_type": j.get("type"),
              "origin_xyz": _vec(o.get("xyz")) if o is not None else [0.0] * 3,
              "origin_rpy": _vec(o.get("rpy")) if o is not None else [0.0] * 3,
              "axis": _vec(ax.get("xyz")) if ax is not None else [0.0, 0.0, 0.0]}
        if lim is not None:
            jr["limit"] = {k: float(lim.get(k)) for k in ("lower", "upper", "effort", "velocity")
                           if lim.get(k) is not None}
        children.setdefault(p, []).append((c, jr))
        has_parent.add(c)
    roots = [n for n in links if n not in has_parent]
    assert len(roots) == 1, roots

    order = []

    def visit(name, parent_idx, jr):
        rec = dict(links[name])
        rec["parent"] = parent_idx
        if jr is None:
            rec.update({"joint_name": None, "joint_type": "floating",
                        "origin_xyz": [0.0] * 3, "origin_rpy": [0.0] * 3, "axis": [0.0] * 3})
        else:
            rec.update(jr)
        idx = len(order)
        order.append(rec)
        for c, cj in sorted(children.get(name, []), key=lambda t: t[0]):
            visit(c, idx, cj)

    visit(roots[0], -1, None)
    dof_names = [r["joint_name"] for r in order if r["joint_type"] in ("revolute", "continuous", "prismatic")]
    return {"robot": root.get("name"), "source": os.path.relpath(path, REF),
            "num_bodies": len(order), "num_dofs": len(dof_names),
            "body_names": [r["name"] for r in order], "dof_names": dof_names, "links": order}


def main():
    os.makedirs(OUT, exist_ok=True)
    for key, rel in MODELS.items():
        m = parse(os.path.join(REF, rel))
        with open(os.path.join(OUT, key + ".model.json"), "w") as f:
            json.dump(m, f, indent=1)
        print(key, m["num_bodies"], "bodies", m["num_dofs"], "dofs; total mass",
              round(sum(l["mass"] for l in m["links"]), 4))
        print("  dofs:", m["dof_names"])


if __name__ == "__main__":
    sys.exit(main())
