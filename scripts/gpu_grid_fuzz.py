"""Fuzz of the uniform-grid closest hit (csrc/rt_grid.h) against the tree and the list walk, on the GPU box:
    python scripts/gpu_grid_fuzz.py [n_scenes] [first_seed]
Random sphere-only scenes — 16 to 3 000 spheres, equal or log-normal radii, flat layers / cubes / thin slabs, with and
without one or two huge spheres, centred at the origin or far from it, unit or tiny or large scale — each uploaded with a
random cell size.  Where upload builds a grid: (1) 20 000 rays through the production kernels (random, axis-parallel,
near-axis, origins inside / on / far from spheres) — grid == list walk bit for bit unless the list walk's hit is a miss in
exact geometry, in which case the grid may return what the list walk returns without that sphere; (2) a 96 x 64 x 4 spp
frame at depth 12 — grid == tree == list walk, bits and ray counts per depth.  Prints one line per scene and a summary."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ray_tracing_in_one_weekend_amd as rt  # noqa: E402
from helpers import grid_fuzz_scene, path_keys  # noqa: E402



def explain(s, p, seed, fg, ft, fl, sg, st, sl, sc_, sr_):
    """RTOW_FUZZ_EXPLAIN=1: a frame that differs between the three searches — which one, where, and the first bounce of the pixel's
    paths at which grid, tree and list walk part (every sample re-traced through rt_debug_bounce's production kernels)."""
    from test_gpu_parity import _primary_rays
    u = np.uint32
    print(f"  seed {seed}: grid == list {np.array_equal(fg.view(u), fl.view(u))}, tree == list {np.array_equal(ft.view(u), fl.view(u))}")
    for nm, x in (("grid", sg), ("tree", st), ("list", sl)):
        print(f"  rays per depth, {nm}: {list(x.rays_per_depth)[:14]}")
    q = rt.make_params(p.nx, p.ny, p.spp, max_depth=p.max_depth, seed=int(p.seed))
    jj, ii = np.nonzero(np.any((fg.view(u) != fl.view(u)) | (ft.view(u) != fl.view(u)), axis=2))
    print(f"  {len(jj)} pixels differ: {list(zip(ii.tolist(), jj.tolist()))[:10]}")
    for pi, pj in list(zip(ii, jj))[:6]:
        print(f"  pixel ({pi},{pj}): grid {fg[pj, pi].tolist()} tree {ft[pj, pi].tolist()} list {fl[pj, pi].tolist()}")
    # every path of the frame, bounce by bounce, once per search (each following its own results): rays per depth as the frames report them?
    n = q.spp
    pj_all, pi_all = np.divmod(np.arange(q.nx * q.ny), q.nx)
    o0, d0, keys0 = _primary_rays(s, q, np.repeat(pi_all, n), np.repeat(pj_all, n), np.tile(np.arange(n), q.nx * q.ny))
    traces = {}
    # (rt_debug_bounce's own kernels take the path keys as given — the production-kernel hook derives them from the slot — and know
    # the tree and the list walk; the frames say grid == tree)
    for nm, grid_opt, fl_ in (("tree", 1, 0), ("list", 0, f.FLAG_BRUTE_FORCE)):
        r.set_option("grid", grid_opt)
        o, d, live = o0.copy(), d0.copy(), np.ones(len(o0), dtype=bool)
        counts, hits = [], []
        for depth in range(q.max_depth + 1):
            idx = np.flatnonzero(live)
            counts.append(len(idx))
            h = np.full(len(o0), -2, dtype=np.int64)
            if len(idx):
                g = r.debug_bounce(o[idx], d[idx], keys0[idx], depth=depth, flags=fl_)
                h[idx] = g["hit"]
                if nm == "list" and depth >= 1:
                    # the same rays through the PRODUCTION closest-hit kernels (their hit records do not depend on the keys)
                    for pn, go in (("grid", 0), ("tree", 1)):
                        r.set_option("grid", go)
                        gp_ = r.debug_bounce(o[idx], d[idx], keys0[idx], depth=depth, flags=prod)
                        for k in np.flatnonzero((gp_["hit"] != g["hit"]) | (gp_["t"].view(u) != g["t"].view(u))):
                            oo, dd = o[idx[k]].astype(np.float64), d[idx[k]].astype(np.float64)
                            kk = int(idx[k])
                            print(f"  depth {depth}, path {kk} (pixel {kk // n % q.nx}, {kk // n // q.nx}, sample {kk % n}): production {pn} hit {int(gp_['hit'][k])} t {float(gp_['t'][k])!r}, "
                                  f"list hit {int(g['hit'][k])} t {float(g['t'][k])!r}")
                            print(f"    o {o[idx[k]].tolist()} d {d[idx[k]].tolist()}")
                            for hh in sorted({int(gp_['hit'][k]), int(g['hit'][k])} - {-1}):
                                oc = oo - sc_[hh]
                                miss2 = oc @ oc - (oc @ dd) ** 2 / (dd @ dd)
                                print(f"    sphere {hh}: centre {sc_[hh].tolist()} r {sr_[hh]!r}: (closest approach / r)^2 = {miss2 / sr_[hh] ** 2:.12f} (> 1: the exact ray misses it); "
                                      f"origin at {np.sqrt(oc @ oc) / sr_[hh]:.9f} r from the centre")
                    r.set_option("grid", 0)
                alive = g["alive"].astype(bool)
                o[idx[alive]], d[idx[alive]] = g["o"][alive], g["d"][alive]
                live[idx[~alive]] = False
            hits.append(h)
        traces[nm] = (counts, np.stack(hits))
        print(f"  re-traced, {nm}: {counts}")
    r.set_option("grid", 0)
    dv = np.flatnonzero((traces["tree"][1] != traces["list"][1]).any(axis=0))
    print(f"  paths whose re-traced hit sequences differ between tree and list (pixel x, y, sample): {[(int(k // n % q.nx), int(k // n // q.nx), int(k % n)) for k in dv[:8]]}")
    for k in dv[:4]:
        print(f"    path {k}: tree {traces['tree'][1][:, k].tolist()}")
        print(f"    path {k}: list {traces['list'][1][:, k].tolist()}")
        # the bounce where they part, in float64
        dep = int(np.flatnonzero(traces["tree"][1][:, k] != traces["list"][1][:, k])[0])
        o, d = o0[k:k + 1].copy(), d0[k:k + 1].copy()
        for depth in range(dep):
            g = r.debug_bounce(o, d, keys0[k:k + 1], depth=depth, flags=0)
            o, d = g["o"], g["d"]
        gt = r.debug_bounce(o, d, keys0[k:k + 1], depth=dep, flags=0)
        gl = r.debug_bounce(o, d, keys0[k:k + 1], depth=dep, flags=f.FLAG_BRUTE_FORCE)
        oo, dd = o[0].astype(np.float64), d[0].astype(np.float64)
        print(f"    depth {dep}: o {o[0].tolist()} d {d[0].tolist()}: tree hit {int(gt['hit'][0])} t {float(gt['t'][0])}, list hit {int(gl['hit'][0])} t {float(gl['t'][0])}")
        for h in sorted({int(gt['hit'][0]), int(gl['hit'][0])} - {-1}):
            oc = oo - sc_[h]
            miss2 = oc @ oc - (oc @ dd) ** 2 / (dd @ dd)
            print(f"    sphere {h}: centre {sc_[h].tolist()} r {sr_[h]}: (closest approach / r)^2 = {miss2 / sr_[h] ** 2:.9f} (> 1: the exact ray misses it)")
    # depth 0 of the differing pixels: the frames of grid and tree take the closest hit from the pixel's candidate list
    for pi, pj in list(zip(ii, jj))[:6]:
        o, d, keys = _primary_rays(s, q, np.full(n, pi), np.full(n, pj), np.arange(n))
        gl = r.debug_bounce(o, d, keys, depth=0, flags=f.FLAG_BRUTE_FORCE)
        for k in range(n):
            oo, dd = o[k].astype(np.float64), d[k].astype(np.float64)
            oc = oo - sc_
            b_ = oc @ dd
            miss2 = (oc * oc).sum(axis=1) - b_ * b_ / (dd @ dd)
            near = np.argsort(miss2 / sr_ ** 2)[:3]
            print(f"  pixel ({pi},{pj}) sample {k}: list walk hit {int(gl['hit'][k])} t {float(gl['t'][k])!r}; spheres by (closest approach / r)^2: "
                  + ", ".join(f"{int(h)}: {miss2[h] / sr_[h] ** 2:.9f}" for h in near))
    return
    for order in ("as stored", "rows flipped"):
        found = 0
        for pi, pj in list(zip(ii, jj))[:6]:
            pjj = pj if order == "as stored" else p.ny - 1 - pj
            n = q.spp
            o, d, keys = _primary_rays(s, q, np.full(n, pi), np.full(n, pjj), np.arange(n))
            live = np.ones(n, dtype=bool)
            for depth in range(q.max_depth + 1):
                idx = np.flatnonzero(live)
                if not len(idx):
                    break
                res = {}
                for nm, grid_opt, fl_ in (("grid", 0, prod), ("tree", 1, prod), ("list", 0, prod | f.FLAG_BRUTE_FORCE)):
                    r.set_option("grid", grid_opt)
                    res[nm] = r.debug_bounce(o[idx], d[idx], keys[idx], depth=depth, flags=fl_)
                r.set_option("grid", 0)
                g, b = res["grid"], res["list"]
                if os.environ.get("RTOW_FUZZ_EXPLAIN") == "2":
                    print(f"    depth {depth}: samples {idx.tolist()} hit {[res[nm]['hit'].tolist() for nm in ('grid', 'tree', 'list')]} alive {g['alive'].tolist()} t {g['t'].tolist()}")
                for k in range(len(idx)):
                    hs = [int(res[nm]["hit"][k]) for nm in ("grid", "tree", "list")]
                    ts = [res[nm]["t"][k] for nm in ("grid", "tree", "list")]
                    if len(set(hs)) > 1 or len({x.view(u) for x in ts}) > 1:
                        found += 1
                        oo, dd = o[idx[k]].astype(np.float64), d[idx[k]].astype(np.float64)
                        print(f"  pixel ({pi},{pjj}) sample {idx[k]} depth {depth}: hit grid/tree/list {hs}, t {[float(x) for x in ts]}")
                        print(f"    o {o[idx[k]].tolist()} d {d[idx[k]].tolist()}")
                        for h in sorted(set(x for x in hs if x >= 0)):
                            oc = oo - sc_[h]
                            miss2 = oc @ oc - (oc @ dd) ** 2 / (dd @ dd)
                            print(f"    sphere {h}: centre {sc_[h].tolist()} r {sr_[h]}, closest approach^2 / r^2 = {miss2 / sr_[h] ** 2:.9f} (> 1: the exact ray misses it)")
                        live[idx[k]] = False
                alive = g["alive"].astype(bool) & live[idx]
                o[idx[alive]], d[idx[alive]] = g["o"][alive], g["d"][alive]
                live[idx[~alive]] = False
        print(f"  rows {order}: {found} diverging bounces found")
        if found:
            break


n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = rt._ffi
r = rt.Renderer(0)
prod = f.FLAG_PRODUCTION_KERNELS
n_grid = n_bad = n_fp = 0
for seed in range(seed0, seed0 + n_scenes):
    s, rng, n, scale, shape, ext, centre, c, rad, n_huge = grid_fuzz_scene(rt, seed)
    r.set_option("grid", 0)
    r.set_option("grid_cell", int(rng.choice([0, 0, 700, 1000, 2000, 3500])))
    r.upload(s)
    info = r.scene_info()
    if not info["grid"]:
        print(f"seed {seed}: {n} spheres ({shape}, scale {scale:g}, {n_huge} huge): no grid", flush=True)
        continue
    n_grid += 1
    # rays
    m = 20000
    allc = np.concatenate([c, c[:1]])
    o = (rng.uniform(-1.3, 1.3, (m, 3)) * np.array(ext) * 1.2 + centre) * scale
    d = rng.normal(size=(m, 3))
    k4 = m // 4
    pick = rng.integers(0, n, k4)
    o[:k4] = c[pick] + d[:k4] / np.linalg.norm(d[:k4], axis=1, keepdims=True) * np.abs(rad[pick])[:, None]   # on a surface
    o[k4:k4 + 500] = c[rng.integers(0, n, 500)]                                                               # at a centre
    axes = np.eye(3)[rng.integers(0, 3, 1500)] * rng.choice([-1.0, 1.0], 1500)[:, None]
    d[k4 + 500:k4 + 2000] = axes                                                                              # axis-parallel
    d[k4 + 2000:k4 + 3000] = axes[:1000] + rng.normal(size=(1000, 3)) * 10.0 ** -rng.integers(3, 9, (1000, 1))  # nearly so
    o[k4 + 3000:k4 + 3300] = (centre + rng.normal(size=(300, 3)) * 3e5) * scale                               # far away
    o, d = o.astype(np.float32), d.astype(np.float32)
    d = (d * (np.float32(1) / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32))[:, None]).astype(np.float32)
    keys = path_keys(0, np.arange(m), np.zeros(m, dtype=np.uint64))
    g = r.debug_bounce(o, d, keys, depth=2, flags=prod)
    b = r.debug_bounce(o, d, keys, depth=2, flags=prod | f.FLAG_BRUTE_FORCE)
    a = s.arrays()
    sc_ = np.stack([a["sph_cx"], a["sph_cy"], a["sph_cz"]], 1).astype(np.float64)
    sr_ = np.abs(a["sph_r"].astype(np.float64))
    diff = np.flatnonzero((g["hit"] != b["hit"]) | (g["t"].view(np.uint32) != b["t"].view(np.uint32)))
    bad = 0
    for i in diff:
        h = b["hit"][i]
        oc = o[i].astype(np.float64) - sc_[h] if h >= 0 else None
        dd = d[i].astype(np.float64)
        miss_exact = h >= 0 and (oc @ oc - (oc @ dd) ** 2 / (dd @ dd)) > (sr_[h] * (1 + 1e-6)) ** 2
        if not (miss_exact and g["hit"][i] != h and (g["hit"][i] < 0 or g["t"][i] >= b["t"][i])):
            bad += 1
    n_fp += len(diff) - bad
    # frame
    p = rt.make_params(96, 64, 4, max_depth=12, seed=seed)
    fg, _, sg = r.render(s.camera, p)
    r.set_option("grid", 1)
    ft, _, st = r.render(s.camera, p)
    r.set_option("grid", 0)
    p.flags = f.FLAG_BRUTE_FORCE
    fl, _, sl = r.render(s.camera, p)
    same = (np.array_equal(fg.view(np.uint32), fl.view(np.uint32)) and np.array_equal(ft.view(np.uint32), fl.view(np.uint32))
            and list(sg.rays_per_depth) == list(sl.rays_per_depth) == list(st.rays_per_depth))
    ok = bad == 0 and same
    if not same and os.environ.get("RTOW_FUZZ_EXPLAIN"):
        explain(s, p, seed, fg, ft, fl, sg, st, sl, sc_, sr_)
    n_bad += 0 if ok else 1
    print(f"seed {seed}: {n} spheres ({shape}, scale {scale:g}, {n_huge} huge), grid {info['grid_cells']} refs {info['grid_refs']} large {info['grid_always']} "
          f"lds {info['grid_lds_bytes']}: rays differing {len(diff)} (unexplained {bad}), frame {'==' if same else 'DIFFERS'} ({sg.n_rays} rays)"
          + ("" if ok else "   <-- FAIL"), flush=True)
print(f"{n_scenes} scenes, {n_grid} with a grid, {n_bad} failures, {n_fp} rays where the list walk reports an fp32 false positive that the grid culls")
sys.exit(1 if n_bad else 0)
