"""What-if, on the CPU, before any kernel is written: how many 3D-DDA cell steps and exact sphere tests would a uniform grid
over the spheres cost per SECONDARY ray of a sphere-only scene, against the BVH4's node visits and leaf tests?

Rays: the oracle's own bounce hook (test infrastructure) iterated from the primary rays of a coarse frame, so the
distribution is the one k_intersect sees at depths 1..D.  The grid: spheres whose box covers more than `big_cells` cells
go to an always-tested list (the ground sphere of sphere_scene); the grid spans the union box of the rest.

    python scripts/whatif_grid_sim.py [scene] [--cells 0.5 ...]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import ray_tracing_in_one_weekend_amd as rt  # noqa: E402
from oracle import binding as orc  # noqa: E402
from test_gpu_parity import _primary_rays  # noqa: E402


def secondary_rays(scene, nx, ny, spp, depths):
    p = rt.make_params(nx, ny, spp, max_depth=50, seed=95)
    jj, ii, ss = np.meshgrid(np.arange(ny), np.arange(nx), np.arange(spp), indexing="ij")
    o, d, keys = _primary_rays(scene, p, ii.ravel(), jj.ravel(), ss.ravel())
    out = []
    for depth in range(depths):
        r = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=depth, accel=orc.ACCEL_LIST)
        alive = r["alive"] != 0
        o, d, keys = r["o"][alive], r["d"][alive], keys[alive]
        out.append((o.copy(), d.copy()))
        if len(o) == 0:
            break
    return out


def build_grid(c, r, cell, big_cells):
    lo, hi = c - r[:, None], c + r[:, None]
    ext = hi - lo
    ncell_cover = np.prod(np.ceil(ext / cell), axis=1)
    big = ncell_cover > big_cells
    g0 = lo[~big].min(axis=0)
    g1 = hi[~big].max(axis=0)
    dims = np.maximum(np.ceil((g1 - g0) / cell).astype(int), 1)
    cells = {}
    for s in np.nonzero(~big)[0]:
        a = np.clip(np.floor((lo[s] - g0) / cell - 1e-3).astype(int), 0, dims - 1)
        b = np.clip(np.floor((hi[s] - g0) / cell + 1e-3).astype(int), 0, dims - 1)
        for x in range(a[0], b[0] + 1):
            for y in range(a[1], b[1] + 1):
                for z in range(a[2], b[2] + 1):
                    # exact sphere/box overlap would prune corners; keep the box test (what a simple builder does)
                    cells.setdefault((x, y, z), []).append(s)
    return g0, dims, cells, np.nonzero(big)[0]


def sphere_t(o, d, c, r, tmax):
    oc = o - c
    a = d @ d
    hb = oc @ d
    cc = oc @ oc - r * r
    disc = hb * hb - a * cc
    if disc < 0:
        return None, False
    sq = np.sqrt(disc)
    t = (-hb - sq) / a
    if t < 1e-3 or t > tmax:
        t = (-hb + sq) / a
        if t < 1e-3 or t > tmax:
            return None, True
    return t, True


def simulate(o, d, c, r, g0, dims, cells, big, cell):
    """-> per ray: dda steps (cells visited), sphere tests in cells, tests that got past the discriminant"""
    n = len(o)
    steps = np.zeros(n, int)
    tests = np.zeros(n, int)
    rep1 = np.zeros(n, int)   # tests of the sphere tested last (a one-entry mailbox would skip them)
    rep2 = np.zeros(n, int)   # ... of one of the two tested last
    repall = np.zeros(n, int) # ... of any sphere this ray has tested before
    g1 = g0 + dims * cell
    for i in range(n):
        oo, dd = o[i].astype(np.float64), d[i].astype(np.float64)
        tbest = np.inf
        for s in big:
            t, _ = sphere_t(oo, dd, c[s], r[s], tbest)
            if t is not None:
                tbest = t
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / dd
            t0 = (g0 - oo) * inv
            t1 = (g1 - oo) * inv
        tn = np.nanmax(np.minimum(t0, t1))
        tf = np.nanmin(np.maximum(t0, t1))
        tn = max(tn, 0.0)
        if not (tn <= tf) or tn > tbest:
            continue
        p = oo + dd * tn
        ix = np.clip(np.floor((p - g0) / cell).astype(int), 0, dims - 1)
        step = np.where(dd > 0, 1, -1)
        nextp = g0 + (ix + (dd > 0)) * cell
        with np.errstate(divide="ignore", invalid="ignore"):
            tmax = np.where(dd != 0, (nextp - oo) * inv, np.inf)
            tdel = np.where(dd != 0, cell * np.abs(inv), np.inf)
        seen, last = set(), [-1, -1]
        while True:
            steps[i] += 1
            for s in cells.get((ix[0], ix[1], ix[2]), ()):
                tests[i] += 1
                rep1[i] += s == last[0]
                rep2[i] += s in last
                repall[i] += s in seen
                seen.add(s)
                if s != last[0]:
                    last = [s, last[0]]
                t, _ = sphere_t(oo, dd, c[s], r[s], tbest)
                if t is not None:
                    tbest = t
            k = int(np.argmin(tmax))
            if tbest <= tmax[k]:
                break
            ix[k] += step[k]
            if ix[k] < 0 or ix[k] >= dims[k]:
                break
            tmax[k] += tdel[k]
    simulate.repeats = (rep1, rep2, repall)
    return steps, tests


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="sphere_scene")
    ap.add_argument("--cells", type=float, nargs="*", default=[0.35, 0.5, 0.7, 1.0])
    ap.add_argument("--nx", type=int, default=96)
    ap.add_argument("--ny", type=int, default=54)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--depths", type=int, default=4)
    ap.add_argument("--big", type=int, default=512)
    ap.add_argument("--max-rays", type=int, default=3000)
    a = ap.parse_args()
    rt.register_default_images()
    scene = rt.Scene.build(a.scene, a.nx / a.ny)
    fs = scene.flat
    ns = fs.n_spheres
    c = np.stack([np.ctypeslib.as_array(fs.sph_cx, (ns,)), np.ctypeslib.as_array(fs.sph_cy, (ns,)),
                  np.ctypeslib.as_array(fs.sph_cz, (ns,))], axis=1).astype(np.float64)
    r = np.abs(np.ctypeslib.as_array(fs.sph_r, (ns,)).astype(np.float64))
    rays = secondary_rays(scene, a.nx, a.ny, a.spp, a.depths)
    o = np.concatenate([x[0] for x in rays])
    d = np.concatenate([x[1] for x in rays])
    print(f"{a.scene}: {ns} spheres, {len(o)} secondary rays of depths 1..{len(rays)} ({[len(x[0]) for x in rays]})")
    rng = np.random.default_rng(1)
    if len(o) > a.max_rays:
        sel = rng.choice(len(o), a.max_rays, replace=False)
        o, d = o[sel], d[sel]
    for cell in a.cells:
        g0, dims, cells, big = build_grid(c, r, cell, a.big)
        refs = sum(len(v) for v in cells.values())
        steps, tests = simulate(o, d, c, r, g0, dims, cells, big, cell)
        # a wave of 64 consecutive rays makes max-over-lanes trips if it does not refill: the naive lane utilisation
        nw = len(o) // 64
        ws = steps[:nw * 64].reshape(nw, 64)
        print(f"cell {cell:5.2f}: grid {tuple(dims)} = {int(np.prod(dims))} cells, {refs} refs ({refs / max(1, ns - len(big)):.2f} per sphere), "
              f"{len(big)} always-tested | per ray: {steps.mean():.2f} cells (p50 {np.median(steps):.0f}, p90 {np.percentile(steps, 90):.0f}, "
              f"max {steps.max()}), {tests.mean():.2f} sphere tests in cells (p90 {np.percentile(tests, 90):.0f}, max {tests.max()}) | "
              f"rays that never enter the grid {np.mean(steps == 0):.2f} | 64-ray blocks: mean/max steps {ws.mean() / max(ws.max(axis=1).mean(), 1e-9):.2f}")
        r1, r2, ra = simulate.repeats
        print(f"            repeated tests per ray (the same sphere listed in several cells of the ray's path): {ra.mean():.2f} of {tests.mean():.2f}; "
              f"a mailbox of the sphere tested last would skip {r1.mean():.2f}, of the last two {r2.mean():.2f}")


if __name__ == "__main__":
    main()
