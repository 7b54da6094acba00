"""The uniform grid of sphere-only scenes (csrc/rt_grid.h build_sphere_grid) on the CPU, through rt_debug_grid_build — host code of
the GPU library, no GPU.  What the device walk relies on is checked against a float64 restatement of the construction rule:
every sphere that is not "large" is listed in every cell its surface grown by pad / 2 can touch (so whatever cell the fp32
walk believes a hit point to lie in, within its rounding distance, lists the sphere); the lists hold nothing twice and
nothing that is far from the cell; large spheres are the ones the rule names; and a float64 DDA over the cells finds, for
random rays, every sphere the ray hits."""
import numpy as np
import pytest


def _spheres(scene):
    a = scene.arrays()
    return np.stack([a["sph_cx"], a["sph_cy"], a["sph_cz"]], 1).astype(np.float64), np.abs(a["sph_r"].astype(np.float64))


def _check_lists(g, c, r):
    cells, refs = g["cells"], g["refs"]
    cnt, off = cells & 0xFFF, cells >> 12
    flat_off, flat_cnt = off.ravel(), cnt.ravel()
    assert np.array_equal(flat_off, np.concatenate([[0], np.cumsum(flat_cnt)[:-1]])) and int(flat_cnt.sum()) == len(refs)
    org, cs, pad = g["origin"].astype(np.float64), g["cell"].astype(np.float64), g["pad"]
    nz, ny, nx = cells.shape
    small = np.setdiff1d(np.arange(len(r)), g["large"])
    # the grid's box holds every small sphere with room to spare
    hi = org + cs * np.array([nx, ny, nz])
    assert np.all(c[small] - r[small, None] - pad > org - 1e-9 * np.abs(org)) and np.all(c[small] + r[small, None] + pad < hi + 1e-9 * np.abs(hi))
    listed = {}
    for z in range(nz):
        for y in range(ny):
            for x in range(nx):
                ids = refs[off[z, y, x]: off[z, y, x] + cnt[z, y, x]]
                assert len(set(ids.tolist())) == len(ids) and np.all(np.diff(ids.astype(int)) > 0)  # once each, in index order
                for s in ids:
                    listed.setdefault(int(s), []).append((x, y, z))
    assert set(listed) <= set(small.tolist())
    for s in small:
        lo = np.floor((c[s] - r[s] - 0.5 * pad - org) / cs).astype(int)
        hi_ = np.floor((c[s] + r[s] + 0.5 * pad - org) / cs).astype(int)
        have = set(listed.get(int(s), []))
        for z in range(max(lo[2], 0), min(hi_[2], nz - 1) + 1):
            for y in range(max(lo[1], 0), min(hi_[1], ny - 1) + 1):
                for x in range(max(lo[0], 0), min(hi_[0], nx - 1) + 1):
                    mn = org + cs * np.array([x, y, z])
                    e = np.maximum(np.maximum(mn - c[s], c[s] - (mn + cs)), 0.0)
                    if e @ e <= (r[s] + 0.5 * pad) ** 2:
                        assert (x, y, z) in have, (s, (x, y, z))  # reachable within pad / 2 -> listed
        for (x, y, z) in have:  # and nothing from far away
            mn = org + cs * np.array([x, y, z])
            e = np.maximum(np.maximum(mn - c[s], c[s] - (mn + cs)), 0.0)
            assert e @ e <= (r[s] + 1.01 * pad) ** 2, (s, (x, y, z))


def _dda_finds_every_hit(g, c, r, rng, n_rays=400):
    """float64 3D-DDA over the cells (the walk of k_intersect_grid without its rounding): the spheres listed in the cells it
    visits, plus the large ones, contain every sphere the ray hits in front of the first hit."""
    cells, refs = g["cells"], g["refs"]
    cnt, off = cells & 0xFFF, cells >> 12
    org, cs = g["origin"].astype(np.float64), g["cell"].astype(np.float64)
    dims = np.array(g["dims"])
    hi = org + cs * dims
    for _ in range(n_rays):
        s0 = rng.integers(0, len(r))
        o = c[s0] + rng.normal(size=3) * (r[s0] * rng.choice([1.0, 3.0, 30.0]))
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        oc = o - c
        hb = oc @ d
        disc = hb * hb - (np.einsum("ij,ij->i", oc, oc) - r * r)
        t = np.where(disc >= 0, -hb - np.sqrt(np.maximum(disc, 0)), np.inf)
        t = np.where(t < 1e-3, np.where(disc >= 0, -hb + np.sqrt(np.maximum(disc, 0)), np.inf), t)
        t = np.where(t < 1e-3, np.inf, t)
        if not np.isfinite(t.min()):
            continue
        winner = int(np.argmin(t))
        if winner in g["large"]:
            continue
        with np.errstate(divide="ignore"):
            inv = 1.0 / d
        t0, t1 = (org - o) * inv, (hi - o) * inv
        tn, tf = max(np.minimum(t0, t1).max(), 0.0), np.maximum(t0, t1).min()
        assert tn <= tf and tn <= t[winner]  # a hit sphere lies inside the grid's box
        p = o + d * tn
        ix = np.clip(np.floor((p - org) / cs).astype(int), 0, dims - 1)
        seen = set()
        for _step in range(int(dims.sum()) + 2):
            seen.update(refs[off[ix[2], ix[1], ix[0]]: off[ix[2], ix[1], ix[0]] + cnt[ix[2], ix[1], ix[0]]].tolist())
            nxt = org + (ix + (d > 0)) * cs
            tm = (nxt - o) * inv
            k = int(np.argmin(tm))
            if tm[k] > t[winner] or winner in seen:
                break
            ix[k] += 1 if d[k] > 0 else -1
            if ix[k] < 0 or ix[k] >= dims[k]:
                break
        assert winner in seen, (o, d, winner)


@pytest.mark.parametrize("name,large", [("sphere_scene", 4), ("pbr_sweep_scene", 1)])
def test_grid_of_the_headline_scenes(rt, name, large):
    rt.register_default_images()
    scene = rt.Scene.build(name, 16 / 9)
    c, r = _spheres(scene)
    g = rt.grid_build(scene)
    assert g is not None and len(g["large"]) == large and min(g["dims"]) == 1  # one layer of cells for a layer of spheres
    assert g["pad"] == pytest.approx(float(np.median(2 * r)) * 1.4 / 128, rel=1e-5) and g["max_coord"] == pytest.approx(g["pad"] * 2 ** 20)
    # "large" = a box of more than 64 nominal cells, (floor((2 r + 2 pad) / cell) + 2)^3 > 64: the ground (and sphere_scene's three r = 1 spheres)
    nominal = g["pad"] * 128
    assert sorted(g["large"]) == sorted(np.flatnonzero((np.floor((2 * r + 2 * g["pad"]) / nominal) + 2) ** 3 > 64).tolist())
    _check_lists(g, c, r)
    _dda_finds_every_hit(g, c, r, np.random.default_rng(1))
    # other cell sizes: same invariants; a budget too small for any grid: none
    others = 0
    for pm in (1000, 2000, 2500):
        g2 = rt.grid_build(scene, cell_per_mille=pm)
        if g2 is None:  # (a cell size the scene does not admit: too few cells or too many references per cell)
            continue
        assert g2["dims"] != g["dims"]
        _check_lists(g2, c, r)
        others += 1
    assert others >= 1
    assert rt.grid_build(scene, lds_budget=4096) is None


def test_grid_of_random_clouds(rt):
    f = rt._ffi
    rng = np.random.default_rng(3)
    built = 0
    for trial in range(12):
        n = int(rng.choice([16, 60, 300, 900]))
        ext = np.array([(8, 0.3, 8), (4, 4, 4), (9, 9, 0.4)][trial % 3])
        scale = float(rng.choice([1.0, 1e-2, 100.0]))
        centre = rng.normal(size=3) * (0.0 if trial % 2 else 50.0)
        s = rt.Scene.new()
        m = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.5, 0.5, 0.5)))
        rad = (np.full(n, 0.2) if trial % 4 else np.exp(rng.normal(np.log(0.2), 0.4, n))) * scale
        for ci, ri in zip((rng.uniform(-1, 1, (n, 3)) * ext + centre) * scale, rad):
            s.sphere(tuple(float(x) for x in ci), float(ri * (-1 if rng.random() < 0.1 else 1)), m, "s")
        if trial % 3 == 0:
            s.sphere(tuple(float(x) for x in (centre + np.array([0, -1000.5, 0])) * scale), 1000.0 * scale, m, "ground")
        s.set_camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 1.0)
        s.finish()
        g = rt.grid_build(s)
        if g is None:
            continue
        built += 1
        c, r = _spheres(s)
        assert len(g["large"]) <= 4 and int(np.prod(g["dims"])) >= 64 and len(g["refs"]) <= 4 * int(np.prod(g["dims"]))
        _check_lists(g, c, r)
        _dda_finds_every_hit(g, c, r, rng, n_rays=150)
    assert built >= 6
    # no grid: too few spheres, a general scene, spheres at coordinates fp32 cannot resolve against their size
    assert rt.grid_build(rt.Scene.build("test_sphere", 2.0)) is None and rt.grid_build(rt.Scene.build("cornell_box", 1.0)) is None
    s = rt.Scene.new()
    m = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.5, 0.5, 0.5)))
    for ci in rng.uniform(-5, 5, (200, 3)) + 1e6:
        s.sphere(tuple(float(x) for x in ci), 0.1, m, "far")
    s.set_camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 1.0)
    s.finish()
    assert rt.grid_build(s) is None
