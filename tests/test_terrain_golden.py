"""Terrain raster and _get_heights against the reference (tests/golden/terrain.npz, SURVEY G-7/G-8)."""
import os

import numpy as np
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
