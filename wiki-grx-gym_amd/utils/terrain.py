"""Procedural heightfield terrain (host side, one-off at env creation).

Restates what the reference builds in ``legged_gym/utils/terrain.py:38-164`` on top of the
generators of ``isaacgym/terrain_utils.py:17-283``: an int16 height raster
(horizontal_scale 0.1 m, vertical_scale 0.005 m) of ``num_rows x num_cols`` square tiles
surrounded by a flat border, one terrain *type* per column and one *difficulty* per row
(curriculum layout), plus the spawn origin of every tile.

Tile generators are written against a small ``Tile`` record (raw int16 raster + scales).  Random
tiles draw from a ``numpy.random.RandomState`` in the same call order as the reference draws from
the global numpy generator, so a reference run seeded with ``np.random.seed(s)`` and
``Terrain(cfg, n, seed=s)`` produce the same raster (tests/test_terrain_golden.py; the
rough-slope tiles go through a bilinear upsampling that the reference delegates to the removed
``scipy.interpolate.interp2d`` -- pinned within +-1 raster unit).

Not implemented ("next"): stepping stones / gap / pit tiles, which the default
``terrain_proportions`` never select, ``selected`` terrains and the trimesh conversion -- the
physics of this build queries the height raster directly for 'heightfield' and 'trimesh' alike.
"""
import numpy as np


class Tile:
    def __init__(self, pixels, horizontal_scale, vertical_scale):
        self.width = self.length = int(pixels)
        self.horizontal_scale = horizontal_scale
        self.vertical_scale = vertical_scale
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def _bilinear_resample(coarse, out_rows, out_cols):
    """Linear interpolation of ``coarse`` (r, c) sampled on linspace(0, L) grids onto
    (out_rows, out_cols) linspace grids over the same extent (== interp2d(kind='linear'))."""
    r, c = coarse.shape
    fi = np.linspace(0.0, r - 1.0, out_rows)
    fj = np.linspace(0.0, c - 1.0, out_cols)
    i0 = np.clip(np.floor(fi).astype(int), 0, r - 2)
    j0 = np.clip(np.floor(fj).astype(int), 0, c - 2)
    ti = (fi - i0)[:, None]
    tj = (fj - j0)[None, :]
    a = coarse[np.ix_(i0, j0)]
    b = coarse[np.ix_(i0 + 1, j0)]
    cc = coarse[np.ix_(i0, j0 + 1)]
    d = coarse[np.ix_(i0 + 1, j0 + 1)]
    return (a * (1 - ti) + b * ti) * (1 - tj) + (cc * (1 - ti) + d * ti) * tj


def pyramid_slope(tile, slope, platform_size):
    """terrain_utils.py:74-106: pyramid whose flanks rise with ``slope``, flat top platform."""
    n = tile.width
    centre = int(n / 2)
    ramp = (centre - np.abs(centre - np.arange(n))) / centre
    peak = int(slope * (tile.horizontal_scale / tile.vertical_scale) * (n / 2))
    tile.height_field_raw += (peak * ramp[:, None] * ramp[None, :]).astype(np.int16)
    half = int(platform_size / tile.horizontal_scale / 2)
    ref = tile.height_field_raw[n // 2 - half, n // 2 - half]
    tile.height_field_raw = np.clip(tile.height_field_raw, min(ref, 0), max(ref, 0))


def uniform_noise(tile, rng, min_height, max_height, step, downsampled_scale):
    """terrain_utils.py:17-51: coarse random heights, bilinearly upsampled, added to the raster."""
    lo, hi, st = int(min_height / tile.vertical_scale), int(max_height / tile.vertical_scale), int(step / tile.vertical_scale)
    levels = np.arange(lo, hi + st, st)
    nr = int(tile.width * tile.horizontal_scale / downsampled_scale)
    nc = int(tile.length * tile.horizontal_scale / downsampled_scale)
    coarse = rng.choice(levels, (nr, nc))
    fine = np.rint(_bilinear_resample(coarse.astype(np.float64), tile.width, tile.length))
    tile.height_field_raw += fine.astype(np.int16)


def pyramid_stairs(tile, step_width, step_height, platform_size):
    """terrain_utils.py:196-227: concentric square steps towards a central platform."""
    sw = int(step_width / tile.horizontal_scale)
    sh = int(step_height / tile.vertical_scale)
    plat = int(platform_size / tile.horizontal_scale)
    lo, hi, h = 0, tile.width, 0
    while (hi - lo) > plat:
        lo += sw
        hi -= sw
        h += sh
        tile.height_field_raw[lo:hi, lo:hi] = h


def discrete_obstacles(tile, rng, max_height, min_size, max_size, num_rects, platform_size):
    """terrain_utils.py:109-150: random rectangular blocks/pits, flat platform in the centre."""
    mh = int(max_height / tile.vertical_scale)
    lo, hi = int(min_size / tile.horizontal_scale), int(max_size / tile.horizontal_scale)
    plat = int(platform_size / tile.horizontal_scale)
    n, m = tile.height_field_raw.shape
    heights = [-mh, -mh // 2, mh // 2, mh]
    sizes = range(lo, hi, 4)
    for _ in range(num_rects):
        w = rng.choice(sizes)
        l = rng.choice(sizes)
        i = rng.choice(range(0, n - w, 4))
        j = rng.choice(range(0, m - l, 4))
        tile.height_field_raw[i:i + w, j:j + l] = rng.choice(heights)
    a, b = (tile.width - plat) // 2, (tile.width + plat) // 2
    tile.height_field_raw[a:b, a:b] = 0


class Terrain:
    """Attributes used by the env (same names as the reference): ``heightsamples`` (int16
    (tot_rows, tot_cols)), ``env_origins`` ((num_rows, num_cols, 3) float), ``tot_rows``,
    ``tot_cols``, ``env_length``, ``env_width``, ``border``, ``cfg``."""

    def __init__(self, cfg, num_robots, seed=None):
        self.cfg = cfg
        self.num_robots = num_robots
        self.type = cfg.mesh_type
        if self.type in ("none", "plane"):
            return
        self.rng = np.random.RandomState(seed) if seed is not None else np.random
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [float(np.sum(cfg.terrain_proportions[:i + 1])) for i in range(len(cfg.terrain_proportions))]
        self.tile_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.width_per_env_pixels = self.tile_pixels
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        if cfg.curriculum:
            for j in range(cfg.num_cols):            # terrain.py:85-92: column-major tile order
                for i in range(cfg.num_rows):
                    self._place(self._make_tile(j / cfg.num_cols + 0.001, i / cfg.num_rows), i, j)
        elif getattr(cfg, "selected", False):
            raise NotImplementedError("terrain.selected is not implemented (SURVEY 8f)")
        else:                                        # terrain.py:74-83 randomized_terrain
            for k in range(cfg.num_rows * cfg.num_cols):
                i, j = np.unravel_index(k, (cfg.num_rows, cfg.num_cols))
                choice = self.rng.uniform(0, 1)
                difficulty = self.rng.choice([0.5, 0.75, 0.9])
                self._place(self._make_tile(choice, difficulty), i, j)
        self.heightsamples = self.height_field_raw

    def _make_tile(self, choice, difficulty):
        """terrain.py:109-145 make_terrain"""
        cfg = self.cfg
        tile = Tile(self.tile_pixels, cfg.horizontal_scale, cfg.vertical_scale)
        slope = difficulty * 0.4
        step_height = 0.05 + 0.18 * difficulty
        obstacle_height = 0.05 + difficulty * 0.2
        p = self.proportions
        if choice < p[0]:
            pyramid_slope(tile, -slope if choice < p[0] / 2 else slope, 3.0)
        elif choice < p[1]:
            pyramid_slope(tile, slope, 3.0)
            uniform_noise(tile, self.rng, -0.05, 0.05, 0.005, 0.2)
        elif choice < p[3]:
            pyramid_stairs(tile, 0.31, -step_height if choice < p[2] else step_height, 3.0)
        elif choice < p[4]:
            discrete_obstacles(tile, self.rng, obstacle_height, 1.0, 2.0, 20, 3.0)
        else:
            raise NotImplementedError("stepping-stone / gap / pit tiles are not implemented (SURVEY 8f)")
        return tile

    def _place(self, tile, i, j):
        """terrain.py:147-164 add_terrain_to_map"""
        x0 = self.border + i * self.length_per_env_pixels
        y0 = self.border + j * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = tile.height_field_raw
        x1 = int((self.env_length / 2.0 - 1) / tile.horizontal_scale)
        x2 = int((self.env_length / 2.0 + 1) / tile.horizontal_scale)
        y1 = int((self.env_width / 2.0 - 1) / tile.horizontal_scale)
        y2 = int((self.env_width / 2.0 + 1) / tile.horizontal_scale)
        z = np.max(tile.height_field_raw[x1:x2, y1:y2]) * tile.vertical_scale
        self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width, z]
