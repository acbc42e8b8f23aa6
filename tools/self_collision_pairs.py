#!/usr/bin/env python3
"""Which URDF links can touch each other?  (self_collisions = 0 = enabled in the reference: legged_robot_config.py:121,
passed to create_actor at legged_robot.py:1022-1028; PhysX filters out pairs of links joined by a joint.)

Monte-Carlo over the joint ranges of every model table in wiki-grx-gym_amd/assets: a pair of links on different,
non-adjacent MOVING bodies is kept when two of their collision spheres come within reach of each other anywhere inside
the joint limits.  Writes the list ("self_collision_link_pairs": [[link_a, link_b], ...], link indices, a < b) into the
model JSON; model.py expands it to sphere pairs for grx_model.pair_a / pair_b.

    python tools/self_collision_pairs.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wiki_grx_gym_amd.model import ASSET_DIR, RobotModel  # noqa: E402


def axang(ax, q):
    ax = ax / np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * K @ K


def link_pairs(key, samples=30000, margin=0.0, seed=0):
    rm = RobotModel(key)
    nb, sph = rm.num_bodies, rm.spheres
    ns = len(sph)
    rng = np.random.default_rng(seed)
    body = np.array([s[0] for s in sph]); link = [s[3] for s in sph]
    rad = np.array([s[2] for s in sph])
    mind = np.full((ns, ns), 1e9)
    for _ in range(samples):
        q = rm.dof_lower + (rm.dof_upper - rm.dof_lower) * rng.random(rm.num_dofs)
        R = [np.eye(3)] * nb; p = [np.zeros(3)] * nb
        for b in range(1, nb):
            par = rm.parent[b]
            R[b] = R[par] @ rm.joint_rot0[b] @ axang(rm.joint_axis[b], q[b - 1]); p[b] = p[par] + R[par] @ rm.joint_pos[b]
        c = np.array([p[s[0]] + R[s[0]] @ s[1] for s in sph])
        d = np.linalg.norm(c[:, None] - c[None], axis=2) - (rad[:, None] + rad[None])
        mind = np.minimum(mind, d)
    pairs = set()
    for i in range(ns):
        for j in range(i + 1, ns):
            bi, bj = body[i], body[j]
            if bi == bj or rm.parent[bi] == bj or rm.parent[bj] == bi:
                continue
            if mind[i, j] < margin:
                pairs.add((min(link[i], link[j]), max(link[i], link[j])))
    return sorted(pairs), rm


def main():
    for f in sorted(os.listdir(ASSET_DIR)):
        if not f.endswith(".model.json"):
            continue
        key = f[:-len(".model.json")]
        pairs, rm = link_pairs(key)
        path = os.path.join(ASSET_DIR, f)
        raw = json.load(open(path))
        raw["self_collision_link_pairs"] = [list(p) for p in pairs]
        json.dump(raw, open(path, "w"), indent=1)
        nsp = sum(1 for a, b in pairs for i in rm.spheres for j in rm.spheres if i[3] == a and j[3] == b)
        print(key, len(pairs), "link pairs ->", nsp, "sphere pairs:", [(rm.body_names[a], rm.body_names[b]) for a, b in pairs])


if __name__ == "__main__":
    main()
