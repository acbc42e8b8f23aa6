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

All seven tile families of ``make_terrain`` are available (the default ``terrain_proportions`` only select the
first five), so are ``selected`` terrains (``terrain_kwargs = {"type": <generator name>, ...}``) and the
raster -> triangle-mesh conversion with the vertical-face correction (``slope_treshold``, default 0.75) that the
reference applies for ``mesh_type = 'trimesh'``.  The physics of this build queries the height raster directly for
'heightfield' and 'trimesh' alike; ``vertices`` / ``triangles`` are produced for API completeness (viewers, export).
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


def stepping_stones(tile, rng, stone_size, stone_distance, max_height, platform_size, depth=-10.0):
    """terrain_utils.py:227-283: rows of square stones over a deep pit, each row shifted by a random offset, every
    stone at a random height in [-max_height, max_height); flat platform in the centre.  Same RNG call order."""
    n = tile.width
    ss, sd = int(stone_size / tile.horizontal_scale), int(stone_distance / tile.horizontal_scale)
    mh = int(max_height / tile.vertical_scale)
    plat = int(platform_size / tile.horizontal_scale)
    levels = np.arange(-mh - 1, mh, step=1)
    H = tile.height_field_raw
    H[:, :] = int(depth / tile.vertical_scale)
    y = 0
    while y < n:                                   # the tile is square: the reference's length >= width branch
        y_end = min(n, y + ss)
        x = rng.randint(0, ss)
        H[0:max(0, x - sd), y:y_end] = rng.choice(levels)      # partial stone before the first hole
        while x < n:
            H[x:min(n, x + ss), y:y_end] = rng.choice(levels)
            x += ss + sd
        y += ss + sd
    a, b = (n - plat) // 2, (n + plat) // 2
    H[a:b, a:b] = 0


def gap(tile, gap_size, platform_size):
    """terrain.py:166-178 gap_terrain: a bottomless square ring (raw -1000) around the central platform."""
    n = tile.width
    g, plat = int(gap_size / tile.horizontal_scale), int(platform_size / tile.horizontal_scale)
    c = n // 2
    inner = (n - plat) // 2
    outer = inner + g
    tile.height_field_raw[c - outer:c + outer, c - outer:c + outer] = -1000
    tile.height_field_raw[c - inner:c + inner, c - inner:c + inner] = 0


def pit(tile, depth, platform_size):
    """terrain.py:180-187 pit_terrain: the central square lowered by `depth`."""
    n = tile.width
    d, half = int(depth / tile.vertical_scale), int(platform_size / tile.horizontal_scale / 2)
    tile.height_field_raw[n // 2 - half:n // 2 + half, n // 2 - half:n // 2 + half] = -d


def heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """terrain_utils.py:286-350: one vertex per raster sample, two triangles per cell.  With a slope threshold, the
    low side of every step steeper than the threshold is moved one cell towards the high side, which turns ramps
    of one cell into vertical faces (what a foot edge actually meets on stairs)."""
    hf = np.asarray(height_field_raw)
    R, Cn = hf.shape
    xx, yy = np.meshgrid(np.linspace(0, (R - 1) * horizontal_scale, R), np.linspace(0, (Cn - 1) * horizontal_scale, Cn), indexing="ij")
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        h = hf.astype(np.int64)
        mx, my, mc = np.zeros((R, Cn)), np.zeros((R, Cn)), np.zeros((R, Cn))
        mx[:-1, :] += (h[1:, :] - h[:-1, :] > thr)
        mx[1:, :] -= (h[:-1, :] - h[1:, :] > thr)
        my[:, :-1] += (h[:, 1:] - h[:, :-1] > thr)
        my[:, 1:] -= (h[:, :-1] - h[:, 1:] > thr)
        mc[:-1, :-1] += (h[1:, 1:] - h[:-1, :-1] > thr)
        mc[1:, 1:] -= (h[:-1, :-1] - h[1:, 1:] > thr)
        xx = xx + (mx + mc * (mx == 0)) * horizontal_scale
        yy = yy + (my + mc * (my == 0)) * horizontal_scale
    vertices = np.stack([xx.ravel(), yy.ravel(), hf.ravel() * vertical_scale], axis=1).astype(np.float32)
    i0 = (np.arange(R - 1)[:, None] * Cn + np.arange(Cn - 1)[None, :]).ravel()      # cell corner (i, j)
    i1, i2, i3 = i0 + 1, i0 + Cn, i0 + Cn + 1
    triangles = np.empty((2 * i0.size, 3), dtype=np.uint32)
    triangles[0::2] = np.stack([i0, i3, i1], axis=1)
    triangles[1::2] = np.stack([i0, i2, i3], axis=1)
    return vertices, triangles


TILE_GENERATORS = {   # names accepted by cfg.terrain_kwargs["type"] (the reference eval()s the name)
    "pyramid_sloped_terrain": lambda tile, rng, **kw: pyramid_slope(tile, kw.get("slope", 1), kw.get("platform_size", 1.0)),
    "random_uniform_terrain": lambda tile, rng, **kw: uniform_noise(tile, rng, kw["min_height"], kw["max_height"], kw.get("step", 1), kw.get("downsampled_scale")),
    "pyramid_stairs_terrain": lambda tile, rng, **kw: pyramid_stairs(tile, kw["step_width"], kw["step_height"], kw.get("platform_size", 1.0)),
    "discrete_obstacles_terrain": lambda tile, rng, **kw: discrete_obstacles(tile, rng, kw["max_height"], kw["min_size"], kw["max_size"], kw["num_rects"], kw.get("platform_size", 1.0)),
    "stepping_stones_terrain": lambda tile, rng, **kw: stepping_stones(tile, rng, kw["stone_size"], kw["stone_distance"], kw["max_height"], kw.get("platform_size", 1.0), kw.get("depth", -10.0)),
    "gap_terrain": lambda tile, rng, **kw: gap(tile, kw["gap_size"], kw.get("platform_size", 1.0)),
    "pit_terrain": lambda tile, rng, **kw: pit(tile, kw["depth"], kw.get("platform_size", 1.0)),
}


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
        elif getattr(cfg, "selected", False):        # terrain.py:94-106 selected_terrain: one generator for every tile
            kw = dict(cfg.terrain_kwargs)
            gen = TILE_GENERATORS[str(kw.pop("type")).split(".")[-1]]
            kw = kw.get("terrain_kwargs", kw)
            for k in range(cfg.num_rows * cfg.num_cols):
                i, j = np.unravel_index(k, (cfg.num_rows, cfg.num_cols))
                tile = Tile(self.tile_pixels, cfg.horizontal_scale, cfg.vertical_scale)
                gen(tile, self.rng, **kw)
                self._place(tile, i, j)
        else:                                        # terrain.py:74-83 randomized_terrain
            for k in range(cfg.num_rows * cfg.num_cols):
                i, j = np.unravel_index(k, (cfg.num_rows, cfg.num_cols))
                choice = self.rng.uniform(0, 1)
                difficulty = self.rng.choice([0.5, 0.75, 0.9])
                self._place(self._make_tile(choice, difficulty), i, j)
        self.heightsamples = self.height_field_raw
        if self.type == "trimesh":                   # terrain.py:68-72
            self.vertices, self.triangles = heightfield_to_trimesh(self.height_field_raw, cfg.horizontal_scale, cfg.vertical_scale,
                                                                   getattr(cfg, "slope_treshold", None))

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
        elif choice < p[5]:
            stepping_stones(tile, self.rng, 1.5 * (1.05 - difficulty), 0.05 if difficulty == 0 else 0.1, 0.0, 4.0)
        elif choice < p[6]:
            gap(tile, 1.0 * difficulty, 3.0)
        else:
            pit(tile, 1.0 * difficulty, 4.0)
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
