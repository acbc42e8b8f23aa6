"""Terrain raster and _get_heights against the reference (tests/golden/terrain.npz, SURVEY G-7/G-8)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import make_cfg, make_sims
from wiki_grx_gym_amd.envs import config
from wiki_grx_gym_amd.utils.terrain import Terrain

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _terrain():
    tcfg = config.LeggedRobotCfg.terrain()
    tcfg.mesh_type = "heightfield"
    return Terrain(tcfg, 64, seed=1), tcfg


def test_heightfield_matches_reference():
    d = np.load(os.path.join(G, "terrain.npz"))
    ter, tcfg = _terrain()
    ref = d["heightsamples"]
    assert ter.heightsamples.shape == ref.shape == (1300, 2100) and ter.heightsamples.dtype == np.int16
    diff = np.abs(ter.heightsamples.astype(np.int32) - ref.astype(np.int32))
    b, px = ter.border, ter.tile_pixels
    # deterministic tiles (smooth slopes cols 0-1, stairs cols 4-15) and the RNG-driven discrete
    # obstacles (cols 16-19, same RandomState call order) must be bit-exact
    for col in list(range(0, 2)) + list(range(4, 20)):
        assert diff[:, b + col * px: b + (col + 1) * px].max() == 0, f"column {col} differs"
    # rough slopes (cols 2-3): bilinear upsampling + rint; the reference used scipy's removed interp2d
    rough = diff[:, b + 2 * px: b + 4 * px]
    assert rough.max() <= 1 and (rough > 0).mean() < 0.01
    np.testing.assert_allclose(ter.env_origins, d["env_origins"], atol=0.0051)
    cols_exact = [c for c in range(20) if c not in (2, 3)]
    np.testing.assert_array_equal(ter.env_origins[:, cols_exact], d["env_origins"][:, cols_exact])


def test_get_heights_matches_reference():
    """Oracle measure_heights on the REFERENCE raster (incl. border clamps / negative coordinates)."""
    from oracle.binding import OracleSim, PipelineState
    from wiki_grx_gym_amd.envs import build_config
    d = np.load(os.path.join(G, "terrain.npz"))
    ter, _ = _terrain()
    ter.heightsamples = ter.height_field_raw = d["heightsamples"].copy()
    cfg = make_cfg(terrain="heightfield")
    N = d["root"].shape[0]
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, N, terrain=ter)
    for prec, tol in (("f64", 1e-6), ("f32", 1e-6)):
        sim = OracleSim(c, prec, keep)
        mism = 0
        for i in range(N):
            ps = PipelineState()
            for k in range(13):
                ps.root[k] = float(d["root"][i][k])
            ps.torso_R[0] = ps.torso_R[4] = ps.torso_R[8] = 1.0
            sim.post_physics(i, ps, apply_reset=False)
        got = sim.tensor("MEASURED_HEIGHTS").numpy()
        # heights are quantised (0.005 m): a sample falling within rounding of a cell edge may pick
        # the neighbouring cell in fp32; allow a handful of such points, none in fp64
        bad = np.abs(got - d["heights"]) > tol
        assert bad.mean() <= (0.0 if prec == "f64" else 2e-3), f"{prec}: {bad.sum()} of {bad.size} height samples differ"


def test_all_tile_families_and_trimesh_match_reference():
    """Stepping stones / gap / pit tiles, a 3 x 10 curriculum grid and the raster -> trimesh conversion with the
    slope-threshold correction (tests/golden/terrain_all_tiles.npz from the reference, seed 5)."""
    d = np.load(os.path.join(G, "terrain_all_tiles.npz"))
    tcfg = config.LeggedRobotCfg.terrain()
    tcfg.mesh_type = "trimesh"
    tcfg.num_rows, tcfg.num_cols, tcfg.border_size = 3, 10, 5
    tcfg.terrain_proportions = [0.1, 0.1, 0.2, 0.2, 0.1, 0.1, 0.1, 0.1]
    ter = Terrain(tcfg, 30, seed=5)
    ref = d["heightsamples"]
    assert ter.heightsamples.shape == ref.shape
    diff = np.abs(ter.heightsamples.astype(np.int32) - ref.astype(np.int32))
    b, px = ter.border, ter.tile_pixels
    for col in range(10):
        blk = diff[:, b + col * px: b + (col + 1) * px]
        if col == 1:      # rough slope: bilinear upsampling stand-in for scipy's removed interp2d
            assert blk.max() <= 1 and (blk > 0).mean() < 0.01
        else:             # smooth slope, stairs, discrete obstacles, stepping stones, gap, pit: bit-exact
            assert blk.max() == 0, f"column {col} differs"
    assert ref.min() == -2000 and (ref == -1000).any()      # stepping-stone pits and the gap really are in the grid
    np.testing.assert_allclose(ter.env_origins, d["env_origins"], atol=0.0051)
    # trimesh: identical topology; vertices identical wherever the raster is (the one rough tile may move a few)
    assert len(ter.vertices) == int(d["n_vert"]) and len(ter.triangles) == int(d["n_tri"])
    np.testing.assert_array_equal(ter.triangles[::997], d["tri_sample"])
    np.testing.assert_array_equal(ter.triangles.astype(np.int64).sum(0), d["tri_sum"])
    dv = np.abs(ter.vertices[::997] - d["vert_sample"])
    assert (dv.max(axis=1) > 1e-6).mean() < 0.01
    # with the reference's raster as input the conversion itself is exact
    from wiki_grx_gym_amd.utils.terrain import heightfield_to_trimesh
    v, t = heightfield_to_trimesh(ref, tcfg.horizontal_scale, tcfg.vertical_scale, tcfg.slope_treshold)
    np.testing.assert_array_equal(v[::997], d["vert_sample"])
    np.testing.assert_allclose(v.astype(np.float64).sum(0), d["vert_sum"], rtol=1e-9)


def test_selected_terrain_builds_every_tile_from_one_generator():
    """cfg.selected: the reference's selected_terrain is broken (AttributeError at terrain.py:103); this build
    implements its evident intent.  Pinned by construction: same generator + RNG stream as building the tiles by hand."""
    from wiki_grx_gym_amd.utils.terrain import Tile, stepping_stones
    tcfg = config.LeggedRobotCfg.terrain()
    tcfg.mesh_type = "heightfield"
    tcfg.curriculum, tcfg.selected = False, True
    tcfg.num_rows, tcfg.num_cols, tcfg.border_size = 2, 2, 5
    tcfg.terrain_kwargs = {"type": "terrain_utils.stepping_stones_terrain",
                           "terrain_kwargs": dict(stone_size=0.8, stone_distance=0.2, max_height=0.05, platform_size=2.0)}
    ter = Terrain(tcfg, 4, seed=7)
    rng = np.random.RandomState(7)
    b, px = ter.border, ter.tile_pixels
    for k in range(4):
        i, j = np.unravel_index(k, (2, 2))
        tile = Tile(px, tcfg.horizontal_scale, tcfg.vertical_scale)
        stepping_stones(tile, rng, 0.8, 0.2, 0.05, 2.0)
        np.testing.assert_array_equal(ter.heightsamples[b + i * px: b + (i + 1) * px, b + j * px: b + (j + 1) * px], tile.height_field_raw)
    assert ter.heightsamples.min() == -2000


# ---- mesh_type 'trimesh': the physics terrain surface against the reference's slope-corrected triangle mesh ----
def _mesh_height(v, t, pts):
    """Height of the TOP surface of the triangle mesh (v, t) under the (n, 2) points: a vertical ray, the highest hit; triangles whose
    projection is degenerate (the corrected, vertical faces) do not count."""
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    den = (b[:, 1] - c[:, 1]) * (a[:, 0] - c[:, 0]) + (c[:, 0] - b[:, 0]) * (a[:, 1] - c[:, 1])
    ok = np.abs(den) > 1e-9
    a, b, c, den = a[ok], b[ok], c[ok], den[ok]
    out = np.full(len(pts), -1e9)
    for i0 in range(0, len(pts), 256):
        p = pts[i0:i0 + 256]
        l1 = ((b[None, :, 1] - c[None, :, 1]) * (p[:, None, 0] - c[None, :, 0]) + (c[None, :, 0] - b[None, :, 0]) * (p[:, None, 1] - c[None, :, 1])) / den[None]
        l2 = ((c[None, :, 1] - a[None, :, 1]) * (p[:, None, 0] - c[None, :, 0]) + (a[None, :, 0] - c[None, :, 0]) * (p[:, None, 1] - c[None, :, 1])) / den[None]
        l3 = 1 - l1 - l2
        z = np.where((l1 >= -1e-9) & (l2 >= -1e-9) & (l3 >= -1e-9), l1 * a[None, :, 2] + l2 * b[None, :, 2] + l3 * c[None, :, 2], -1e9)
        out[i0:i0 + 256] = z.max(1)
    return out


def _trimesh_tile(name):
    """(cfg, terrain namespace, fixture arrays) of one tile of tests/golden/trimesh_tiles.npz as a 1 x 1 terrain without border."""
    import types
    d = np.load(os.path.join(G, "trimesh_tiles.npz"))
    cfg = make_cfg(terrain="trimesh", curriculum=False)
    cfg.terrain.border_size = 0.0
    cfg.terrain.num_rows = cfg.terrain.num_cols = 1
    assert cfg.terrain.horizontal_scale == float(d["horizontal_scale"]) and cfg.terrain.vertical_scale == float(d["vertical_scale"]) and cfg.terrain.slope_treshold == float(d["slope_threshold"])
    blk = d[name + "_raster"]
    ter = types.SimpleNamespace(heightsamples=blk, env_origins=np.zeros((1, 1, 3), np.float32))
    return cfg, ter, blk, d[name + "_vertices"].astype(np.float64), d[name + "_triangles"]


def check_trimesh_surface(query, name, tol):
    """`query((n, 2) points) -> heights` of the build's physics terrain (oracle or HIP) against a vertical ray cast onto the REFERENCE's mesh of the
    same raster (isaacgym terrain_utils.py:286-350 with slope_threshold 0.75, the call of legged_robot.py:903-921).  Round 6: the build holds that
    mesh itself, a plane per triangle half of every raster cell (oracle trimesh_build / csrc/grx_capi.cpp build_trimesh_tables), so
      * the two surfaces are the same to `tol` EVERYWHERE a cell's half lies under one triangle of the mesh -- all of the pyramid-stairs tile, whose
        treads are three cells deep and 57 % of whose cells hold a corrected edge or a moved vertex (round 5's band, a quarter-cell ramp then), the
        concave corners' split diagonals included;
      * what is left: cells next to a point where THREE levels meet (discrete obstacles of different heights touching), whose halves the vertices the
        reference slides along a face cut once more -- at most 0.5 % of the obstacle tile's points, bounded by the raster's local step."""
    cfg, ter, blk, v, t = _trimesh_tile(name)
    hs, vs = cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale
    H = blk.astype(np.int32)
    n = H.shape[0]
    T = cfg.terrain.slope_treshold * hs / vs
    gx, gy = np.meshgrid(np.arange(n) * hs, np.arange(n) * hs, indexing="ij")
    moved = (np.abs(v[:, 0].reshape(n, n) - gx) > 1e-6) | (np.abs(v[:, 1].reshape(n, n) - gy) > 1e-6)
    sx, sy = np.abs(np.diff(H, axis=0)) > T, np.abs(np.diff(H, axis=1)) > T
    cell_sx, cell_sy = sx[:, :-1] | sx[:, 1:], sy[:-1, :] | sy[1:, :]
    band = cell_sx | cell_sy | moved[:-1, :-1] | moved[1:, :-1] | moved[:-1, 1:] | moved[1:, 1:]       # the cells the correction touches
    assert moved.sum() > 400 and 0.1 < band.mean() < 0.7, (name, moved.sum(), band.mean())
    rng = np.random.default_rng(3)
    g = np.arange(0.513, 7.4, 0.0731)                      # 95 x 95 points, incommensurate with the 0.1 m raster, + 12000 random ones
    pts = np.concatenate([np.array([(x, y) for x in g for y in g]), rng.uniform(0.11, 7.85, (12000, 2))])
    f = pts / hs - np.floor(pts / hs)
    pts = pts[(np.minimum(f, 1 - f).min(1) > 1e-4) & (np.abs(f[:, 0] - f[:, 1]) > 1e-4)]   # (a point ON a grid line or a cell diagonal sits on a face: either level)
    hm = _mesh_height(v, t, pts)
    hq = np.asarray(query(pts), dtype=np.float64)
    dev = np.abs(hm - hq)
    ij = np.floor(pts / hs).astype(int)
    inb = band[ij[:, 0], ij[:, 1]]
    assert inb.sum() > 2000 and (~inb).sum() > 2000
    off = dev > tol
    assert not off[~inb].any(), (name, dev[~inb].max())
    assert off.mean() <= (0.0 if name == "stairs" else 5e-3), (name, int(off.sum()), len(off))   # (observed: 0 of 21 000 on the stairs, 0.15 % on the obstacles)
    # bounded by the local step of the raster (5 x 5 vertices around the cell: a vertex moves by one cell)
    pad = np.pad(H, 2, mode="edge")
    loc = np.stack([pad[2 + di:2 + di + n, 2 + dj:2 + dj + n] for di in (-2, -1, 0, 1, 2) for dj in (-2, -1, 0, 1, 2)])
    step = (loc.max(0) - loc.min(0)) * vs
    cell_step = np.maximum.reduce([step[:-1, :-1], step[1:, :-1], step[:-1, 1:], step[1:, 1:]])
    assert (dev <= cell_step[ij[:, 0], ij[:, 1]] + tol).all()
    return float(inb.mean()), float(off.mean())


def _closest_point_on_triangles(p, a, b, c):
    """closest point of each triangle (a, b, c: (n, 3)) to the point p (3,): Ericson's region walk, vectorised over the triangles"""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(1), (ac * ap).sum(1)
    bp = p - b; d3, d4 = (ab * bp).sum(1), (ac * bp).sum(1)
    cp = p - c; d5, d6 = (ab * cp).sum(1), (ac * cp).sum(1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    out, done = np.zeros_like(a), np.zeros(len(a), bool)

    def take(m, val):
        m = m & ~done
        out[m] = val[m]
        done[m] = True
    with np.errstate(all="ignore"):
        take((d1 <= 0) & (d2 <= 0), a)
        take((d3 >= 0) & (d4 <= d3), b)
        take((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab)
        take((d6 >= 0) & (d5 <= d6), c)
        take((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
        take((va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0), b + ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None] * (c - b))
        den = 1 / (va + vb + vc)
        take(np.ones(len(a), bool), a + ab * (vb * den)[:, None] + ac * (vc * den)[:, None])
    return out


def check_trimesh_walls(ground, wall, name, tol, n_points=4000):
    """`wall(x, y, z, r) -> (overlap, nx, ny, nz)` of the build (oracle or HIP) against the REFERENCE mesh's vertical faces: the triangles of
    tests/golden/trimesh_tiles.npz whose projection is a segment.  For spheres of radius 4 cm resting up to 12 cm above the build's ground at random
    places, the true overlap is r - (distance to the closest such triangle whose closest point stands above the ground under the centre).
    The build keeps, per raster cell, the faces on the cell's four sides as rectangles and the ends of faces at its four corners as posts
    (oracle trimesh_build / csrc/grx_capi.cpp build_trimesh_tables); what that cannot hold -- the notch the corrected mesh leaves along a cell's
    diagonal at a concave corner, the triangular fins where three levels meet -- is a bounded fraction of the contacts, counted here."""
    cfg, ter, blk, v, t = _trimesh_tile(name)
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    den = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (c[:, 0] - a[:, 0]) * (b[:, 1] - a[:, 1])
    vert = (np.abs(den) < 1e-9) & (np.linalg.norm(np.cross(b - a, c - a), axis=1) > 1e-9)
    A, B, Cc = a[vert], b[vert], c[vert]
    assert vert.sum() > 500
    rng = np.random.default_rng(2)
    pts = rng.uniform(0.3, 7.7, (n_points, 2))
    lift = rng.uniform(0.0, 0.12, n_points)
    r = 0.04
    g = np.asarray(ground(pts), dtype=np.float64)
    got = np.asarray(wall(np.column_stack([pts, g + lift, np.full(n_points, r)])), dtype=np.float64)
    touching = both = wrong = 0
    for k, (x, y) in enumerate(pts):
        p = np.array([x, y, g[k] + lift[k]])
        near = (np.abs(A[:, 0] - x) < 0.25) & (np.abs(A[:, 1] - y) < 0.25)
        d_ref, n_ref = 0.0, np.zeros(3)
        if near.any():
            q = _closest_point_on_triangles(p, A[near], B[near], Cc[near])
            dist = np.linalg.norm(q - p, axis=1)
            ok = (q[:, 2] > g[k] + 1e-4) & (dist > 1e-9)
            if ok.any() and r - dist[ok].min() > 0:
                m = np.argmin(np.where(ok, dist, 1e9))
                d_ref, n_ref = r - dist[m], (p - q[m]) / dist[m]
        if d_ref > 0 or got[k, 0] > 0:
            touching += 1
            if abs(got[k, 0] - d_ref) <= tol and (d_ref == 0 or np.abs(got[k, 1:] - n_ref).max() <= 1e-3 + 50 * tol / r):
                both += 1
            else:
                wrong += 1
    return touching, both, wrong


@pytest.mark.parametrize("name", ["stairs", "obstacles"])
def test_trimesh_surface_against_the_reference_mesh(name):
    from oracle.binding import OracleSim
    from wiki_grx_gym_amd.envs import build_config
    cfg, ter, _, _, _ = _trimesh_tile(name)
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 1, terrain=ter)
    assert c.vertical_faces == 1
    for prec, tol in (("f64", 1e-6), ("f32", 2e-5)):
        ora = OracleSim(c, prec, keep)
        check_trimesh_surface(lambda pts: [ora.terrain(x, y)[0] for x, y in pts], name, tol)


def test_trimesh_surface_of_every_tile_family():
    """The same ray cast over a 3 x 10 curriculum grid that holds EVERY tile family (the grid of test_all_tile_families_and_trimesh_match_reference, which
    pins this build's mesh conversion to the reference's vertices and triangles): smooth and rough slopes, stairs up and down, discrete obstacles, gap and
    pit -- the oracle's ground IS the mesh; stepping stones (not in the reference's default terrain_proportions) -- stones one or two cells apart over a
    5 m drop, where the reference's correction leaves needles and overlapping sheets inside the gaps: the stones' tops are exact, the gaps are not."""
    from oracle.binding import OracleSim
    from wiki_grx_gym_amd.envs import build_config
    tcfg = config.LeggedRobotCfg.terrain()
    tcfg.mesh_type = "trimesh"
    tcfg.num_rows, tcfg.num_cols, tcfg.border_size = 3, 10, 5
    tcfg.terrain_proportions = [0.1, 0.1, 0.2, 0.2, 0.1, 0.1, 0.1, 0.1]
    ter = Terrain(tcfg, 30, seed=5)
    v, t = np.asarray(ter.vertices, dtype=np.float64), np.asarray(ter.triangles)
    cfg = make_cfg(terrain="trimesh")
    cfg.terrain = tcfg
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 30, terrain=ter)
    ora = OracleSim(c, "f64", keep)
    hs, b, px = tcfg.horizontal_scale, tcfg.border_size, ter.tile_pixels
    rng = np.random.default_rng(0)
    family = ["smooth slope", "rough slope", "stairs", "stairs", "stairs down", "stairs down", "obstacles", "stepping stones", "gap", "pit"]
    allowed = {"rough slope": (5e-3, 5e-3), "obstacles": (8e-3, 8e-3), "stepping stones": (0.25, 0.06)}       # (fraction off by > 2e-5 m, by > 1 mm); observed 0.0013 / 0.0027 / 0.154, 0.030
    for col in range(10):
        pts = np.column_stack([rng.uniform(ter.border + 1, ter.border + 3 * px - 1, 600) * hs, rng.uniform(ter.border + col * px + 1, ter.border + (col + 1) * px - 1, 600) * hs])
        f = pts / hs - np.floor(pts / hs)
        pts = pts[(np.minimum(f, 1 - f).min(1) > 1e-4) & (np.abs(f[:, 0] - f[:, 1]) > 1e-4)]
        near = (v[t[:, 0], 1] > pts[:, 1].min() - 0.5) & (v[t[:, 0], 1] < pts[:, 1].max() + 0.5)
        dev = np.abs(_mesh_height(v, t[near], pts) - np.array([ora.terrain(x - b, y - b)[0] for x, y in pts]))
        lim = allowed.get(family[col], (0.0, 0.0))
        assert (dev > 2e-5).mean() <= lim[0] and (dev > 1e-3).mean() <= lim[1], (col, family[col], (dev > 2e-5).mean(), (dev > 1e-3).mean())


def test_trimesh_tables_of_the_library_equal_the_oracles():
    """The PRODUCT's host side of mesh_type 'trimesh' on the CPU tier: the per-cell tables grx_create derives from a raster (csrc/grx_capi.cpp
    build_trimesh_tables, reached without a device through the test-only grx_debug_trimesh_tables) are, entry for entry, the ones the oracle builds for
    itself (trimesh_build) -- on the two reference tiles and on the 3 x 10 grid with every tile family.  (What the kernels do with them: the gpu tier.)"""
    import ctypes as C
    from oracle.binding import OracleSim
    from wiki_grx_gym_amd import sim
    from wiki_grx_gym_amd.envs import build_config
    api = sim.load_hip_library()
    cases = []
    for name in ("stairs", "obstacles"):
        cfg, ter, _, _, _ = _trimesh_tile(name)
        cases.append((name, cfg, ter, 1))
    tcfg = config.LeggedRobotCfg.terrain()
    tcfg.mesh_type = "trimesh"
    tcfg.num_rows, tcfg.num_cols, tcfg.border_size = 3, 10, 5
    tcfg.terrain_proportions = [0.1, 0.1, 0.2, 0.2, 0.1, 0.1, 0.1, 0.1]
    cfg = make_cfg(terrain="trimesh")
    cfg.terrain = tcfg
    cases.append(("every family", cfg, Terrain(tcfg, 30, seed=5), 30))
    for name, cfg, ter, n_env in cases:
        c, keep, _ = build_config.build(cfg, cfg.sim.dt, n_env, terrain=ter)
        rows, cols = int(c.hf_rows), int(c.hf_cols)
        g = np.zeros((rows * cols, 6), np.int16); w = np.zeros((rows * cols, 8), np.int16)
        rc = api["debug_trimesh_tables"](C.byref(c), g.ctypes.data_as(C.POINTER(C.c_int16)), w.ctypes.data_as(C.POINTER(C.c_int16)))
        assert rc == 0, api["last_error"]()
        og, ow = OracleSim(c, "f32", keep).trimesh_tables(rows, cols)
        assert np.array_equal(g, og) and np.array_equal(w, ow), (name, int((g != og).sum()), int((w != ow).sum()))
        faces = (w != np.iinfo(np.int16).min).any(1).mean()
        assert 0.005 < faces < 0.5, (name, faces)        # (cells that hold a face or a post: a few per cent of a curriculum grid, a third of the stairs tile)


@pytest.mark.parametrize("name", ["stairs", "obstacles"])
def test_trimesh_vertical_faces_against_the_reference_mesh(name):
    """The second half of the corrected mesh: its vertical faces as contacts (VERDICT r5 missing #2).  A sphere against the oracle's per-cell
    sides and posts = the same sphere against the mesh's own vertical triangles."""
    from oracle.binding import OracleSim
    from wiki_grx_gym_amd.envs import build_config
    cfg, ter, _, _, _ = _trimesh_tile(name)
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 1, terrain=ter)
    for prec, tol in (("f64", 1e-6), ("f32", 2e-5)):
        ora = OracleSim(c, prec, keep)
        touching, same, wrong = check_trimesh_walls(lambda pts: [ora.terrain(x, y)[0] for x, y in pts], lambda q: [ora.wall(*row) for row in q], name, tol)
        # (observed: stairs 453 of 456 the same, obstacles 129 of 133; the rest are the notch cells and fins of check_trimesh_walls' docstring)
        assert touching > 100 and wrong <= 0.04 * touching, (name, prec, touching, same, wrong)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["stairs", "obstacles"])
def test_trimesh_surface_of_the_hip_kernels_against_the_reference_mesh(name):
    """The same checks on the step kernels' own terrain query and wall contact (grx_debug_terrain launches terrain_height<true>, grx_debug_wall
    wall_gather / wall_contact), and HIP = oracle point by point."""
    from oracle.binding import OracleSim
    from wiki_grx_gym_amd.envs import build_config
    from wiki_grx_gym_amd.sim import HipSim
    cfg, ter, _, _, _ = _trimesh_tile(name)
    c, keep, _ = build_config.build(cfg, cfg.sim.dt, 16, terrain=ter)
    sim = HipSim(c, "cuda:0", keep)
    band_frac, off_frac = check_trimesh_surface(lambda pts: sim.debug_terrain(pts)[:, 0], name, 2e-5)
    print(name, "cells the correction touches", band_frac, "points off the reference mesh", off_frac)

    def hip_wall(q):
        f = sim.debug_wall(q).astype(np.float64)
        d = np.linalg.norm(f, axis=1)
        return np.column_stack([d, f / np.maximum(d, 1e-30)[:, None]])
    touching, same, wrong = check_trimesh_walls(lambda pts: sim.debug_terrain(pts)[:, 0], hip_wall, name, 2e-5)
    assert touching > 100 and wrong <= 0.04 * touching, (name, touching, same, wrong)
    c2, keep2, _ = build_config.build(cfg, cfg.sim.dt, 16, terrain=ter)
    ora = OracleSim(c2, "f32", keep2)
    rng = np.random.default_rng(0)
    pts = rng.uniform(0.05, 7.85, size=(2000, 2))
    a = sim.debug_terrain(pts)
    b = np.array([ora.terrain(x, y) for x, y in pts.astype(np.float32)])
    # (a point within rounding of a cell edge or of a cell's diagonal may sit on the other side in one of the two)
    close = (np.abs(a[:, 0] - b[:, 0]) <= 2e-5) & (np.abs(a[:, 1:] - b[:, 1:]) <= 1e-3 + 1e-3 * np.abs(b[:, 1:])).all(1)
    assert close.mean() >= 0.995, close.mean()
    z = b[:, 0] + rng.uniform(0.0, 0.1, len(pts))
    q = np.column_stack([pts, z, np.full(len(pts), 0.04)]).astype(np.float32)
    wa = sim.debug_wall(q)
    wb = np.array([(lambda o: o[0] * o[1:])(ora.wall(*row)) for row in q])
    assert (np.abs(wb).sum(1) > 0).sum() > 50 and (np.abs(wa - wb).max(1) <= 2e-5).mean() >= 0.995
    sim.close()
