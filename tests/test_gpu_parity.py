"""GPU parity tests: the HIP wavefront path (through the C-ABI) against the CPU oracle on the
same flattened scene, the same counter-based RNG keys and the same decoded texels.

Tolerances.  With -ffp-contract=off every +,-,*,/ and sqrt on the GPU is IEEE-exact, so ray
geometry and all branch decisions are bit-identical to the oracle; only libm-class functions
(sin/cos/acos/atan2/log) differ in the last ulp.  Hence: hit index, t, scattered origin and
direction and ray counts must match EXACTLY; colours (textures, BRDF weights) to 1e-5 relative;
images to RMSE <= 2e-4 in display units (gamma-2, clamped [0,1]) against the recursive oracle,
whose product chain is associated differently from the wavefront's T *= a.
"""
import os
import time

import numpy as np
import pytest

from helpers import ctr_draw, display, path_keys, rays_on_scene, rmse_display

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RMSE_TOL = 2e-4


def _accel_for(orc, scene):
    """The oracle's search for a frame comparison: its list walk for scenes with rectangles, wrappers or media (the reference's
    own BvhNode boxes are unpadded and lose grazing rectangle hits, which the oracle's BVH mode mirrors; the device's padded tree
    equals the list walk bit for bit), the reference's BVH for spheres."""
    return orc.ACCEL_LIST if (scene.flat.n_rects or scene.flat.n_media or scene.flat.n_xforms) else orc.ACCEL_BVH


def _oracle(orc, scene, params, **kw):
    kw.setdefault("rng_mode", orc.RNG_COUNTER)
    kw.setdefault("accel", _accel_for(orc, scene))
    return orc.render(scene.flat_ptr, scene.camera, params, orc.options(**kw))


def _check_bounce(rt, orc, renderer, scene, n=20000, depth=0, seed=1):
    o, d, keys = rays_on_scene(n, seed)
    renderer.upload(scene)
    g = renderer.debug_bounce(o, d, keys, depth=depth)  # LDS BVH traversal
    b = renderer.debug_bounce(o, d, keys, depth=depth, flags=rt._ffi.FLAG_BRUTE_FORCE)  # list walk
    for k in g:
        assert np.array_equal(g[k].view(np.uint8), b[k].view(np.uint8)), k  # BVH culling never changes the result
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=depth, accel=orc.ACCEL_LIST)
    assert np.array_equal(g["hit"], c["hit"])
    assert np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    assert np.array_equal(g["alive"], c["alive"])
    assert np.array_equal(g["o"].view(np.uint32), c["o"].view(np.uint32))
    assert np.array_equal(g["d"].view(np.uint32), c["d"].view(np.uint32))
    for k in ("radiance", "attenuation"):
        a, b = g[k].astype(np.float64), c[k].astype(np.float64)
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin), k
        assert np.allclose(a[fin], b[fin], rtol=2e-5, atol=1e-6), (k, np.abs(a[fin] - b[fin]).max())
    _check_production_kernels(rt, orc, renderer, scene, o, d, depth)
    return g


def _check_production_kernels(rt, orc, renderer, scene, o, d, depth):
    """The same rays through the ray queue and the kernels rt_render launches for a depth >= 1 (persistent-lane
    k_intersect of the scene's instantiation, class-sorting k_shade, wave64 compaction), per ray against the oracle.
    The renderer derives the RNG key from the slot: ray i is pixel i, sample 0, seed 0."""
    keys = path_keys(0, np.arange(len(o)), np.zeros(len(o), dtype=np.uint64))
    p = renderer.debug_bounce(o, d, keys, depth=depth, flags=rt._ffi.FLAG_PRODUCTION_KERNELS)
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=depth, accel=orc.ACCEL_LIST)
    assert np.array_equal(p["hit"], c["hit"])
    assert np.array_equal(p["t"].view(np.uint32), c["t"].view(np.uint32))
    assert np.array_equal(p["alive"], c["alive"])
    live = c["alive"].astype(bool)
    assert np.array_equal(p["o"][live].view(np.uint32), c["o"][live].view(np.uint32))
    assert np.array_equal(p["d"][live].view(np.uint32), c["d"][live].view(np.uint32))
    for k, sel in (("attenuation", live), ("radiance", ~live)):  # a finished path keeps no ray, a survivor emits nothing
        a, b = p[k][sel].astype(np.float64), c[k][sel].astype(np.float64)
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin), k
        assert np.allclose(a[fin], b[fin], rtol=2e-5, atol=1e-6), (k, np.abs(a[fin] - b[fin]).max())
    assert not p["radiance"][live].any()  # emitted = 0 for everything that scatters (material.rs:12-14)


def _false_positive_of_fp32_sphere_hit(o, d, c, r, sph):
    """True when the ray misses sphere `sph` in exact (float64) geometry: a root that fp32 Sphere::hit (hitable.rs:79-83)
    reports only because |oc|^2 - r^2 cancelled.  A box test — the reference's own AABB::hit included — may cull such a hit."""
    oc = o.astype(np.float64) - c[sph]
    dd = d.astype(np.float64)
    dist2 = oc @ oc - (oc @ dd) ** 2 / (dd @ dd)
    return dist2 > (r[sph] * (1 + 1e-6)) ** 2


@pytest.mark.parametrize("name", ["sphere_scene", "pbr_sweep_scene"])
def test_grid_walk_equals_tree_and_list_walk(rt, orc, renderer, name):
    """Sphere-only scenes answer depth >= 1 by a 3D-DDA walk over a uniform grid in LDS (csrc/rt_grid.h).  Through the
    production kernels, on adversarial rays (axis-parallel, origins inside / on spheres, grazing, 1e6 away, near-axis) and on
    the rays of the scene: grid == list walk == oracle bit for bit; grid == tree except where the list walk's hit is a
    false positive of fp32 Sphere::hit that a box test culls.  Every cell size gives the same records."""
    scene = rt.Scene.build(name, 16 / 9)
    a = scene.arrays()
    rng = np.random.default_rng(11)
    c = np.stack([a["sph_cx"], a["sph_cy"], a["sph_cz"]], 1).astype(np.float64)
    r = np.abs(a["sph_r"].astype(np.float64))
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)
    o_list, d_list = [], []
    for k in range(6000):
        s_ = rng.integers(0, len(r))
        ax = axes[rng.integers(0, 6)]
        kind = k % 6
        if kind == 0:    # axis-aligned ray through a sphere centre from outside
            o_list.append(c[s_] - ax * (r[s_] * 3 + 1)); d_list.append(ax)
        elif kind == 1:  # origin at the centre of a sphere
            o_list.append(c[s_]); d_list.append(ax)
        elif kind == 2:  # grazing: offset by exactly r perpendicular to the direction
            perp = axes[(np.argmax(np.abs(ax)) * 2 + 2) % 6]
            o_list.append(c[s_] + perp * r[s_] - ax * 5); d_list.append(ax)
        elif kind == 3:  # very far origin: beyond the grid's coordinate limit, every sphere is tested
            o_list.append(c[s_] - ax * 1e6); d_list.append(ax)
        elif kind == 4:  # random direction from a point on the sphere surface
            v = rng.normal(size=3); v /= np.linalg.norm(v)
            o_list.append(c[s_] + v * r[s_]); w = rng.normal(size=3).astype(np.float32)
            d_list.append(w * (np.float32(1) / np.sqrt(np.float32(w[0] * w[0] + w[1] * w[1]) + np.float32(w[2] * w[2]))))
        else:            # a long, nearly axis-parallel ray skimming the layer of small spheres
            w = ax.astype(np.float32) + rng.normal(size=3).astype(np.float32) * np.float32(10.0 ** -rng.integers(2, 9))
            o_list.append(c[s_] - ax * 30 + rng.normal(size=3) * 0.1); d_list.append(w)
    o = np.asarray(o_list, np.float32)
    d = np.asarray(d_list, np.float32)
    o2, d2, _ = rays_on_scene(20000, 5)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    keys = path_keys(0, np.arange(len(o)), np.zeros(len(o), dtype=np.uint64))
    prod = rt._ffi.FLAG_PRODUCTION_KERNELS
    renderer.upload(scene)
    info = renderer.scene_info()
    assert info["grid"] and info["grid_lds_bytes"] <= 80 * 1024 and info["grid_refs"] > 0, info
    g = renderer.debug_bounce(o, d, keys, depth=3, flags=prod)
    renderer.set_option("grid", 1)
    t = renderer.debug_bounce(o, d, keys, depth=3, flags=prod)
    renderer.set_option("grid", 0)
    b = renderer.debug_bounce(o, d, keys, depth=3, flags=rt._ffi.FLAG_BRUTE_FORCE | prod)
    ref = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=3, accel=orc.ACCEL_LIST)
    assert np.array_equal(b["hit"], ref["hit"]) and np.array_equal(b["t"].view(np.uint32), ref["t"].view(np.uint32))
    for label, x in (("grid", g), ("tree", t)):
        same = (x["hit"] == b["hit"]) & (x["t"].view(np.uint32) == b["t"].view(np.uint32))
        for i in np.flatnonzero(~same):  # a culled hit must be a miss in exact geometry
            assert b["hit"][i] >= 0 and _false_positive_of_fp32_sphere_hit(o[i], d[i], c, r, b["hit"][i]), (label, i)
            # ... and what the search returns instead is what the list walk returns once that sphere is gone: another
            # accepted root behind it, or a miss
            assert x["hit"][i] != b["hit"][i] and (x["hit"][i] < 0 or x["t"][i] >= b["t"][i]), (label, i)
        far = np.arange(len(o)) < 6000
        assert same[~far].mean() > 0.9999, label
    kind3 = (np.arange(len(o)) % 6 == 3) & (np.arange(len(o)) < 6000)
    assert np.array_equal(g["hit"][kind3], b["hit"][kind3])  # far origins: the grid kernel tests every sphere, like the list walk
    assert (g["hit"] >= 0).mean() > 0.5
    # every cell size gives the same records; the cells are what was asked for
    n_other = 0
    for per_mille in (700, 1000, 2000, 3000):
        renderer.set_option("grid_cell", per_mille)
        renderer.upload(scene)
        info2 = renderer.scene_info()
        if not info2["grid"]:  # (a cell size the scene does not admit: too many references per cell, csrc/rt_grid.h)
            continue
        assert info2["grid_cells"] != info["grid_cells"], (per_mille, info2)
        n_other += 1
        g2 = renderer.debug_bounce(o, d, keys, depth=3, flags=prod)
        near = ~kind3
        differ = (g2["hit"] != g["hit"]) | (g2["t"].view(np.uint32) != g["t"].view(np.uint32))
        for i in np.flatnonzero(differ & near):  # only fp32 false positives may depend on which cells a ray visits
            w = g2 if g2["hit"][i] >= 0 and (g["hit"][i] < 0 or g2["t"][i] < g["t"][i]) else g
            assert _false_positive_of_fp32_sphere_hit(o[i], d[i], c, r, w["hit"][i]), (per_mille, i)
        assert differ.mean() < 1e-4
    assert n_other >= 2
    renderer.set_option("grid_cell", 0)
    renderer.upload(scene)
    # whole frames (queues, refills, deep bounces): grid == tree == list walk, bit for bit, ray counts included
    p = rt.make_params(240, 135, 6, max_depth=50, seed=9)
    f_grid, _, s_grid = renderer.render(scene.camera, p)
    renderer.set_option("grid", 1)
    f_tree, _, s_tree = renderer.render(scene.camera, p)
    renderer.set_option("grid", 0)
    p.flags = rt._ffi.FLAG_BRUTE_FORCE
    f_list, _, s_list = renderer.render(scene.camera, p)
    assert np.array_equal(f_grid.view(np.uint32), f_list.view(np.uint32)) and np.array_equal(f_tree.view(np.uint32), f_list.view(np.uint32))
    assert list(s_grid.rays_per_depth) == list(s_list.rays_per_depth) == list(s_tree.rays_per_depth)


def test_grid_is_built_only_where_it_suits(rt, renderer):
    """rt_scene_upload builds the grid for sphere-only scenes of similar spheres and leaves the others to the tree: too few
    spheres, general scenes, spheres of wildly different sizes or coordinates the fp32 walk could not resolve."""
    f = rt._ffi
    for name, want in (("sphere_scene", True), ("pbr_sweep_scene", True), ("test_sphere", False), ("earth_env_scene", False),
                       ("cornell_box", False), ("final_scene", False), ("simple_light_scene", False)):
        renderer.upload(rt.Scene.build(name, 16 / 9))
        assert bool(renderer.scene_info()["grid"]) == want, name
    rng = np.random.default_rng(2)

    def cloud(centres, radii):
        s = rt.Scene.new()
        m = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.6, 0.6, 0.6)))
        for c_, r_ in zip(centres, radii):
            s.sphere(tuple(float(x) for x in c_), float(r_), m, "s")
        s.set_camera((0, 2, 30), (0, 0, 0), (0, 1, 0), 40, 1.5)
        s.finish()
        return s
    # 300 equal spheres far from the origin: cells of 0.2 at coordinates of 1e6 are below what fp32 resolves -> tree
    far = cloud(rng.uniform(-5, 5, (300, 3)) + 1e6, np.full(300, 0.1))
    renderer.upload(far)
    assert not renderer.scene_info()["grid"]
    # radii over four decades: more than four "large" spheres at every cell size tried, or too many references -> tree
    renderer.upload(cloud(rng.uniform(-5, 5, (300, 3)), 10.0 ** rng.uniform(-3, 1, 300)))
    assert not renderer.scene_info()["grid"]
    # a plain cloud gets one, negative radii (hollow glass, as in the book) and zero radii included, and renders like the tree
    radii = np.full(400, 0.25)
    radii[::7] = -0.25
    radii[3::50] = 0.0
    s = cloud(rng.uniform(-6, 6, (400, 3)) * np.array([1, 0.2, 1]), radii)
    renderer.upload(s)
    assert renderer.scene_info()["grid"]
    p = rt.make_params(160, 120, 4, max_depth=10)
    a_, _, sa = renderer.render(s.camera, p)
    renderer.set_option("grid", 1)
    b_, _, sb = renderer.render(s.camera, p)
    assert np.array_equal(a_.view(np.uint32), b_.view(np.uint32)) and list(sa.rays_per_depth) == list(sb.rays_per_depth)


def test_bounce_sphere_scene_all_depth_blocks(rt, orc, renderer):
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    for depth in (0, 1, 7, 50):
        g = _check_bounce(rt, orc, renderer, scene, depth=depth, seed=depth + 1)
    assert (g["hit"] >= 0).mean() > 0.5 and g["alive"].mean() > 0.3


def test_bounce_pbr_and_env_scenes(rt, orc, renderer):
    for name in ("pbr_sweep_scene", "earth_env_scene", "test_sphere", "simple_light_scene"):
        scene = rt.Scene.build(name, 16 / 9)
        _check_bounce(rt, orc, renderer, scene, n=30000, seed=7)


def _single_material_scene(rt, mat_type, **kw):
    s = rt.Scene.new()
    tex = {"const": s.constant_tex((0.7, 0.5, 0.3)), "checker": s.checker_tex((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)),
           "perlin": s.perlin_tex(4.0), "image": s.image_tex("res/earthmap.jpg")}[kw.pop("tex", "const")]
    tex1 = s.constant_tex((0.2, 0.6, 0.9))
    m = s.material(mat_type, tex0=tex, tex1=tex1, color=(0.8, 0.6, 0.2), p=kw.pop("p", (0.5, 0.5, 0.25, 0.0)))
    g = s.material(rt._ffi.MAT_DIFFUSE, tex0=tex1)
    s.sphere((0, -1000, 0), 1000.0, g, "ground")
    for i in range(-3, 4):
        for j in range(-3, 4):
            s.sphere((1.7 * i, 0.7, 1.7 * j), 0.7, m, f"s{i},{j}")
    s.set_sky(kw.pop("sky", rt._ffi.SKY_GRADIENT), kw.pop("env", None))
    s.set_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20, 16 / 9)
    return s.finish()


@pytest.mark.parametrize("mat", list(range(13)))
def test_bounce_every_material(rt, orc, renderer, mat):
    p = {rt._ffi.MAT_METAL: (0.3,), rt._ffi.MAT_DIELECTRIC: (1.5,), rt._ffi.MAT_ROUGH_PLASTIC: (0.3, 1.5),
         rt._ffi.MAT_DISNEY_METAL: (0.4, 0.6, 0.125), rt._ffi.MAT_DISNEY_CLEARCOAT: (0.7,)}.get(mat, (0.5, 0.5, 0.25, 0.0))
    scene = _single_material_scene(rt, mat, p=p)
    _check_bounce(rt, orc, renderer, scene, n=20000, seed=100 + mat)


@pytest.mark.parametrize("tex", ["const", "checker", "perlin", "image"])
def test_bounce_every_texture_and_sky(rt, orc, renderer, tex):
    scene = _single_material_scene(rt, rt._ffi.MAT_DIFFUSE, tex=tex, sky=rt._ffi.SKY_ENV, env="res/newport_loft.jpg")
    _check_bounce(rt, orc, renderer, scene, n=20000, seed=5)
    scene = _single_material_scene(rt, rt._ffi.MAT_EMISSION, tex=tex, sky=rt._ffi.SKY_BLACK)
    _check_bounce(rt, orc, renderer, scene, n=20000, seed=6)


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("where,radius", [(5e7, 1000.0), (1e10, 1e6), (-3e9, 4e5), (3e19, 4e15), (-3e19, 4e15)])
def test_perlin_lattice_index_far_from_the_origin(rt, orc, renderer, where, radius, axis):
    """`p.x.floor() as isize` then rem_euclid(256) (texture.rs:126-138) for coordinates beyond 2^31 — reached by the seventh octave
    (p * 64) of a hit 3.4e7 units out, by every octave at 1e10, and beyond isize at 3e19 — where a 32-bit conversion saturates to an
    odd index and the reference's 64-bit one lands on a multiple of 256.  One axis far, the other two small enough to have
    fractional parts, so that the gradients of the far axis' lattice index do contribute; colours against the oracle."""
    s = rt.Scene.new()
    marble = s.material(rt._ffi.MAT_EMISSION, tex0=s.perlin_tex(4.0))  # emitted = the texture itself, no random number involved
    centre = np.zeros(3)
    centre[axis] = where
    s.sphere(tuple(float(x) for x in centre), radius, marble, "far marble")
    s.set_sky(rt._ffi.SKY_BLACK, None)
    s.set_camera((0, 0, 0), tuple(float(x) for x in centre), (0, 1, 0) if axis != 1 else (1, 0, 0), 20, 1.0)
    scene = s.finish()
    renderer.upload(scene)
    rng = np.random.default_rng(17 + axis)
    n = 6000
    target = centre[None, :] + rng.uniform(-0.7, 0.7, (n, 3)) * radius
    o = (target * (1.0 - 4.0 * radius / abs(where)) + rng.uniform(-0.1, 0.1, (n, 3)) * radius).astype(np.float32)
    d = (target - o.astype(np.float64))
    d = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d * (np.float32(1) / ln)[:, None]).astype(np.float32)
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    g = renderer.debug_bounce(o, d, keys, depth=1)
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=1, accel=orc.ACCEL_LIST)
    assert np.array_equal(g["hit"], c["hit"]) and (g["hit"] >= 0).mean() > 0.5
    assert np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))
    a, b = g["radiance"].astype(np.float64), c["radiance"].astype(np.float64)
    assert np.isfinite(b).all() and np.allclose(a, b, rtol=2e-5, atol=2e-5), np.abs(a - b).max()
    assert len(np.unique(np.round(b[g["hit"] >= 0, 0], 3))) > 50  # a marble, not a constant: the lookups matter
    _check_production_kernels(rt, orc, renderer, scene, o, d, 1)


def test_bvh_equals_brute_force_on_adversarial_rays(rt, orc, renderer):
    """Axis-aligned directions (zero components -> infinite slab reciprocals), origins inside
    spheres, on sphere surfaces, far away, and grazing rays: BVH == list walk == oracle, bit for bit."""
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    a = scene.arrays()
    rng = np.random.default_rng(3)
    c = np.stack([a["sph_cx"], a["sph_cy"], a["sph_cz"]], 1).astype(np.float64)
    r = a["sph_r"].astype(np.float64)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)
    o_list, d_list = [], []
    for k in range(4000):
        s = rng.integers(0, len(r))
        ax = axes[rng.integers(0, 6)]
        kind = k % 5
        if kind == 0:    # axis-aligned ray through a sphere centre from outside
            o_list.append(c[s] - ax * (r[s] * 3 + 1)); d_list.append(ax)
        elif kind == 1:  # origin at the centre of a sphere
            o_list.append(c[s]); d_list.append(ax)
        elif kind == 2:  # grazing: offset by exactly r perpendicular to the direction
            perp = axes[(np.argmax(np.abs(ax)) * 2 + 2) % 6]
            o_list.append(c[s] + perp * r[s] - ax * 5); d_list.append(ax)
        elif kind == 3:  # very far origin
            o_list.append(c[s] - ax * 1e6); d_list.append(ax)
        else:            # random direction from a point on the sphere surface
            v = rng.normal(size=3); v /= np.linalg.norm(v)
            o_list.append(c[s] + v * r[s]); w = rng.normal(size=3).astype(np.float32)
            w = w * (np.float32(1) / np.sqrt(np.float32(w[0] * w[0] + w[1] * w[1]) + np.float32(w[2] * w[2])))
            d_list.append(w)
    o = np.asarray(o_list, np.float32)
    d = np.asarray(d_list, np.float32)
    keys = rng.integers(0, 2**32, size=(len(o), 2), dtype=np.uint64).astype(np.uint32)
    g = renderer.debug_bounce(o, d, keys)
    b = renderer.debug_bounce(o, d, keys, flags=rt._ffi.FLAG_BRUTE_FORCE)
    ref = orc.debug_bounce(scene.flat_ptr, o, d, keys, accel=orc.ACCEL_LIST)
    # the GPU list walk is the oracle's HitableList::hit, bit for bit, on every ray
    assert np.array_equal(b["hit"], ref["hit"]) and np.array_equal(b["t"].view(np.uint32), ref["t"].view(np.uint32))
    kind = np.arange(len(o)) % 5
    same = (g["hit"] == b["hit"]) & (g["t"].view(np.uint32) == b["t"].view(np.uint32))
    # kinds 0,1,2,4: the BVH search returns exactly the list-walk result
    assert same[kind != 3].all(), np.flatnonzero(~same & (kind != 3))[:10]
    # kind 3 (origin 1e6 away): hitable.rs:79-80 cancels catastrophically (|oc|^2 ~ 1e12, ulp 65536), so the
    # list walk reports "hits" on spheres the ray misses by up to hundreds of units; any box test culls them
    # (the reference's own AABB test does too).  Where the two differ, the list-walk hit must be such a
    # false positive in exact (float64) geometry.
    for i in np.flatnonzero(~same):
        sph = b["hit"][i]
        assert sph >= 0
        oc = o[i].astype(np.float64) - c[sph]
        dd = d[i].astype(np.float64)
        dist2 = oc @ oc - (oc @ dd) ** 2 / (dd @ dd)
        assert dist2 > (r[sph] * (1 + 1e-6)) ** 2, (i, sph, dist2, r[sph])
    assert (g["hit"] >= 0).mean() > 0.6
    # whole renders agree bit for bit as well
    p = rt.make_params(160, 90, 4, max_depth=50)
    i1, _, s1 = renderer.render(scene.camera, p)
    for flags in (rt._ffi.FLAG_BRUTE_FORCE,):  # list walk
        p.flags = flags
        i2, _, s2 = renderer.render(scene.camera, p)
        assert np.array_equal(i1.view(np.uint32), i2.view(np.uint32)) and s1.n_rays == s2.n_rays


def test_render_config1_random_spheres(rt, orc, renderer):
    """BASELINE.json config 1: sphere_scene 400x225, 8 spp, depth 8."""
    scene = rt.Scene.build("sphere_scene", 400 / 225)
    p = rt.make_params(400, 225, 8, max_depth=8, seed=95)
    renderer.upload(scene)
    img, rgb8, st = renderer.render(scene.camera, p, want_rgb8=True)
    ref, ref8, so = orc.render(scene.flat_ptr, scene.camera, p, orc.options(rng_mode=orc.RNG_COUNTER), want_rgb8=True)
    assert st.n_paths == so.n_paths == 400 * 225 * 8
    assert st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    assert st.n_texture_fetches == so.n_texture_fetches and st.n_bad_dir == so.n_bad_dir == 0
    _compare_frames(orc, scene, p, img, ref, "config 1", rt, renderer)  # (the scene holds an image texture: texel-edge lookups are re-traced)
    diff8 = np.abs(rgb8.astype(int) - ref8.astype(int))
    assert diff8.max() <= 1 and (diff8 > 0).mean() < 1e-3
    # and the reference-order stream mode agrees statistically (different random numbers)
    stv, _, _ = orc.render(scene.flat_ptr, scene.camera, p, orc.options(rng_mode=orc.RNG_STREAM))
    assert abs(display(img).mean() - display(stv).mean()) < 5e-3


def test_render_full_depth_and_slicing_invariance(rt, orc, renderer):
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    p = rt.make_params(256, 144, 6, max_depth=50)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p)
    assert st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    _compare_frames(orc, scene, p, img, ref, "sphere_scene 256x144x6", rt, renderer)
    # slices of 1, 4 and 6 samples: bit-identical framebuffers (sample order is preserved)
    for s in (1, 4):
        ps = rt.make_params(256, 144, 6, max_depth=50, spp_slice=s)
        im2, _, st2 = renderer.render(scene.camera, ps)
        assert st2.n_slices == (6 + s - 1) // s and st2.n_rays == st.n_rays
        assert np.array_equal(im2.view(np.uint32), img.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name,nx,ny", [("sphere_scene", 96, 54), ("cornell_box", 48, 48)])
def test_many_samples_and_uneven_slices(rt, orc, renderer, name, nx, ny):
    """150 samples per pixel against the oracle, then in slices of 70 + 70 + 10, 64 + 64 + 22 and 128 + 22 samples: the
    sample sum is taken in sample order whatever the slicing, so the frames are bit-identical (the layouts of the path
    slots that were tried on top of this test: scripts/experiments/pixel_major_sample_blocks.patch)."""
    scene = rt.Scene.build(name, nx / ny)
    renderer.upload(scene)
    p = rt.make_params(nx, ny, 150, max_depth=12)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p)
    _rays_agree(st, so, scene, p)
    _compare_frames(orc, scene, p, img, ref, name, rt, renderer)
    for s in (70, 64, 128):
        im2, _, st2 = renderer.render(scene.camera, rt.make_params(nx, ny, 150, max_depth=12, spp_slice=s))
        assert st2.n_slices == (150 + s - 1) // s and st2.n_rays == st.n_rays
        assert np.array_equal(im2.view(np.uint32), img.view(np.uint32)), s


def test_texel_pool_rgba8_and_float_fallback(rt, orc, renderer):
    """An image whose texels are all k/255 (any decoded 8-bit file, texture.rs:176-177) is kept as RGBA8 on the device and
    (float)k / 255 is taken at the lookup: the same frame, bit for bit, as with the float4 pool (RT_OPT_TEXEL_POOL = 1).
    An image with other values takes the float4 pool.  Both against the oracle."""
    rng = np.random.default_rng(7)
    for name, pixels in (("test/eight_bit.img", (rng.integers(0, 256, (32, 64, 3)).astype(np.float32) / np.float32(255.0))),
                         ("test/float.img", rng.random((32, 64, 3), dtype=np.float32))):
        rt.register_image(name, pixels)
        s = rt.Scene.new()
        f = rt._ffi
        img = s.image_tex(name)
        s.sphere((0, 0, -1), 0.5, s.material(f.MAT_LAMBERT, tex0=img), "textured")
        s.sphere((0, -100.5, -1), 100.0, s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.8, 0.8, 0.0))), "ground")
        s.sphere((1.1, 0, -1), 0.5, s.material(f.MAT_EMISSION, tex0=img), "lamp")
        s.set_camera((0, 0.3, 1.5), (0, 0, -1), (0, 1, 0), 50, 2.0)
        s.finish()
        p = rt.make_params(128, 64, 16, max_depth=8)
        renderer.upload(s)
        img_a, _, st = renderer.render(s.camera, p)
        ref, _, so = _oracle(orc, s, p)
        assert st.n_rays == so.n_rays and st.n_texture_fetches == so.n_texture_fetches > 0
        _compare_frames(orc, s, p, img_a, ref, name, rt, renderer)
        renderer.set_option("texel_pool", 1)
        renderer.upload(s)
        img_b, _, _ = renderer.render(s.camera, p)
        renderer.set_option("texel_pool", 0)
        assert np.array_equal(img_a.view(np.uint32), img_b.view(np.uint32)), name


def test_tall_narrow_frame_up_to_the_row_limit(rt, orc, renderer):
    """slot -> (sample, row, column) uses a float-reciprocal division that is exact for quotients below 2^21
    (rt_kernels.h udiv_inv): a shard of 2^21 - 1 rows renders like the oracle (the row index reaches the limit, every
    pixel keyed by its own (row, column)), one of 2^21 rows is refused, and sharding brings it back under the limit."""
    scene = rt.Scene.build("test_sphere", 1.0)
    renderer.upload(scene)
    ny = (1 << 21) - 1
    p = rt.make_params(1, ny, 1, max_depth=4, seed=5)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p)
    assert st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    _compare_frames(orc, scene, p, img, ref, "test_sphere 1 x (2^21 - 1)", rt, renderer)
    with pytest.raises(rt.RtError, match="2\\^21"):
        renderer.render(scene.camera, rt.make_params(1, 1 << 21, 1, max_depth=4))
    part, _, _ = renderer.render(scene.camera, rt.make_params(1, 1 << 21, 1, max_depth=4, seed=5, shard_band=8, shard_count=2, shard_id=1))
    assert part.shape[0] == 1 << 20


def test_tile_order_of_path_slots_does_not_change_images(rt, renderer):
    """Path slots enumerate the pixels of a shard in 8 x 8 tiles (one wave of depth 0 = one tile) when nx is a multiple of 8 —
    the last rows % 8 rows and other widths in rows (rt_kernels.h GenParams::tiles_per_row).  Keys, candidate lists, the sample sum and
    the output image are functions of the pixel, not of its slot: frames, RGB8 and ray counts are the same bit for bit either
    way — full frame, sharded (whole tiles per band and not), with slices, on a sphere-only and on a general scene."""
    for name, nx, ny in (("sphere_scene", 320, 181), ("cornell_box", 96, 96)):  # 181 rows: 22 tile rows + 5 rows
        scene = rt.Scene.build(name, nx / ny)
        renderer.upload(scene)
        for kw in ({}, {"spp_slice": 3}, {"shard_band": 8, "shard_count": 3, "shard_id": 1}, {"shard_band": 4, "shard_count": 2, "shard_id": 0}):
            p = rt.make_params(nx, ny, 6, max_depth=12, seed=3, **kw)
            renderer.set_option("pixel_order", 1)
            rows, rows8, sr = renderer.render(scene.camera, p, want_rgb8=True)
            renderer.set_option("pixel_order", 2)
            tiles, tiles8, st = renderer.render(scene.camera, p, want_rgb8=True)
            renderer.set_option("pixel_order", 0)
            assert np.array_equal(rows.view(np.uint32), tiles.view(np.uint32)) and np.array_equal(rows8, tiles8), (name, kw)
            assert sr.n_rays == st.n_rays and list(sr.rays_per_depth) == list(st.rays_per_depth)


def test_render_sharding_is_bit_invariant(rt, renderer):
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    from ray_tracing_in_one_weekend_amd import shard
    nx, ny = 192, 108
    full, _, st = renderer.render(scene.camera, rt.make_params(nx, ny, 4, max_depth=20))
    for world, band in ((2, 8), (3, 5), (8, 8)):
        parts, rays = [], 0
        for r in range(world):
            im, _, s = renderer.render(scene.camera, rt.make_params(nx, ny, 4, max_depth=20, shard_band=band, shard_count=world, shard_id=r))
            assert im.shape[0] == len(shard.shard_rows(ny, band, world, r))
            parts.append(im)
            rays += s.n_rays
        out = shard.deinterleave(parts, ny, band, world)
        assert np.array_equal(out.view(np.uint32), full.view(np.uint32))
        assert rays == st.n_rays


@pytest.mark.parametrize("name,spp,depth", [("test_sphere", 16, 50), ("earth_env_scene", 8, 12), ("pbr_sweep_scene", 8, 6),
                                            ("simple_light_scene", 16, 50)])
def test_render_other_scenes(rt, orc, renderer, name, spp, depth):
    scene = rt.Scene.build(name, 2.0)
    renderer.upload(scene)
    p = rt.make_params(200, 100, spp, max_depth=depth)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p)
    assert st.n_rays == so.n_rays
    assert st.n_texture_fetches == so.n_texture_fetches
    _compare_frames(orc, scene, p, img, ref, name, rt, renderer)


def test_the_workload_the_reference_ships(rt, orc, renderer):
    """main.rs:63-74 as checked in: test_sphere (demo_scene.rs:229-244), 800 x 400, 128 samples per pixel, MAX_DEPTH 50 — at its
    own size against the oracle: ray counts per depth exact, no pixel off by more than 1e-4, RGB8 within one level."""
    scene = rt.Scene.build("test_sphere", 800 / 400)
    renderer.upload(scene)
    p = rt.make_params(800, 400, 128, max_depth=50, seed=95)
    img, rgb8, st = renderer.render(scene.camera, p, want_rgb8=True)
    ref, ref8, so = orc.render(scene.flat_ptr, scene.camera, p, orc.options(rng_mode=orc.RNG_COUNTER), want_rgb8=True)
    assert st.n_paths == 800 * 400 * 128 and st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    assert 1.7 < st.n_rays / st.n_paths < 1.8
    _compare_frames(orc, scene, p, img, ref, "test_sphere 800x400x128", rt, renderer)
    assert np.abs(rgb8.astype(np.int16) - ref8.astype(np.int16)).max() <= 1


def test_c_host_through_the_headers(rt, renderer, tmp_path):
    """examples/host_main.c — C99, nothing but include/rtow_host.h and include/rtow_mi355x.h, the stand-in for a host in the
    reference's own language (main.rs:62-129) — run as a child process: scene function, context, upload, rt_render into
    rt_host_alloc'ed memory, PNG.  Its RGB8 image equals the ctypes path's byte for byte, on the reference's shipped workload
    (small) and on sphere_scene, whose image texture it registers from a decoded PPM."""
    import subprocess
    from PIL import Image
    from ray_tracing_in_one_weekend_amd import images
    exe = os.path.join(ROOT, "build", "host_main")
    assert os.path.exists(exe), "build/host_main is missing: python -c 'import __graft_entry__ as g; g.build()'"
    for rel in images.DEFAULT_IMAGES:  # the same decode the ctypes path registers (PIL), handed over as 8-bit P6
        with Image.open(os.path.join(images.ASSET_DIR, rel)) as im:
            a = np.asarray(im.convert("RGB"), dtype=np.uint8)
        with open(tmp_path / (os.path.splitext(os.path.basename(rel))[0] + ".ppm"), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]) + a.tobytes())
    for name, nx, ny, spp, depth in (("test_sphere", 200, 100, 16, 50), ("sphere_scene", 320, 180, 8, 50)):
        png, raw = tmp_path / f"{name}.png", tmp_path / f"{name}.rgb8"
        r = subprocess.run([exe, name, str(nx), str(ny), str(spp), str(depth), str(png), str(raw), str(tmp_path)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "Mray/s" in r.stdout and f"saved {png}" in r.stdout
        scene = rt.Scene.build(name, nx / ny)
        renderer.upload(scene)
        _, rgb8, st = renderer.render(scene.camera, rt.make_params(nx, ny, spp, max_depth=depth, seed=95), want_rgb8=True)
        got = np.fromfile(raw, dtype=np.uint8).reshape(ny, nx, 3)
        assert np.array_equal(got, rgb8), name
        assert f"{st.n_rays} rays" in r.stdout
        with Image.open(png) as im:
            assert np.array_equal(np.asarray(im.convert("RGB")), rgb8)


def _primary_rays(scene, p, pix_i, pix_j, samp):
    """main.rs:89-94 + camera.rs:40-46 for the given (pixel, sample) pairs in numpy float32, every operation rounded once
    (glam order: dot = (xx + yy) + zz, normalize = v * (1 / len)) — the very rays the renderer and the oracle trace."""
    f = np.float32
    keys = path_keys(int(p.seed), pix_j.astype(np.uint64) * p.nx + pix_i.astype(np.uint64), samp.astype(np.uint64))
    k0, k1 = keys[:, 0].astype(np.uint64), keys[:, 1].astype(np.uint64)

    def draw(ctr):
        return ((ctr_draw(k0, k1, ctr) >> 8).astype(np.float32) * f(1.0 / 16777216.0)).astype(f)
    u = ((pix_i.astype(f) + draw(0)) / f(p.nx)).astype(f)
    v = ((pix_j.astype(f) + draw(1)) / f(p.ny)).astype(f)
    cam = scene.camera
    org, H, V, llc = (np.array(list(x), dtype=f) for x in (cam.origin, cam.horizontal, cam.vertical, cam.lower_left_corner))
    dirs = np.empty((len(u), 3), dtype=f)
    for k in range(3):
        dirs[:, k] = (((llc[k] + (u * H[k]).astype(f)).astype(f) + (v * V[k]).astype(f)).astype(f) - org[k]).astype(f)
    len2 = (((dirs[:, 0] * dirs[:, 0]).astype(f) + (dirs[:, 1] * dirs[:, 1]).astype(f)).astype(f) + (dirs[:, 2] * dirs[:, 2]).astype(f)).astype(f)
    inv = (f(1.0) / np.sqrt(len2).astype(f)).astype(f)
    dirs = (dirs * inv[:, None]).astype(f)
    return np.tile(org, (len(u), 1)).astype(f), dirs, keys


def _texel_edge_distance(scene, hit, o, d, t):
    """For one segment that samples an image (texture.rs:183-193 through an ImageTex of the hit sphere, or tex_sky_color
    demo_scene.rs:22-26 on a miss): distance of (u, v) to the nearest texel edge in units of 2^-23 (the f32 spacing of a
    texture coordinate near 1: u and v come out of acos / atan2, a division by pi and, for the sky, 1 - u, each good to an
    ulp or two of THAT scale, whatever the size of u W), evaluated in float64 from the f32 hit record; None when the segment
    samples no image."""
    fs = scene.flat
    if hit >= 0:
        if hit >= fs.n_spheres:
            return None  # a rectangle's uv is (p - min) / (max - min), plain arithmetic: exact on both sides
        m = fs.sph_mat[hit]
        imgs = []
        if fs.mat_type[m] in (0, 1, 2, 5, 6, 7, 8, 9, 10, 11) and fs.tex_type[fs.mat_tex0[m]] == 3:  # RT_TEX_IMAGE as tex0
            imgs.append(fs.tex_aux[fs.mat_tex0[m]])
        if fs.mat_type[m] == 8 and fs.tex_type[fs.mat_tex1[m]] == 3:                                  # RoughPlastic's diff_color, pbr.rs:172
            imgs.append(fs.tex_aux[fs.mat_tex1[m]])
        if not imgs:
            return None
        f = np.float32
        o, d = _object_space_ray(fs, hit, o.astype(f), d.astype(f))                 # below its wrappers, hitable.rs:411, 483-492
        pnt = (o + (d * f(t)).astype(f)).astype(f)                                   # Ray::at, math.rs:64
        c = np.array([fs.sph_cx[hit], fs.sph_cy[hit], fs.sph_cz[hit]], dtype=f)
        n = ((pnt - c).astype(f) / f(fs.sph_r[hit])).astype(f).astype(np.float64)  # hitable.rs:95
        flip_u = False
    else:
        if fs.sky_type != 2:
            return None
        imgs, n, flip_u = [fs.sky_image], d.astype(np.float64), True
    theta, phi = np.arccos(-n[1]), np.arctan2(-n[2], n[0]) + np.pi                   # hitable.rs:65-71
    u, v = phi / (2 * np.pi), theta / np.pi
    if flip_u:
        u = 1.0 - u                                                                  # demo_scene.rs:24
    best = None
    for img in imgs:
        x = min(max(u, 0.0), 1.0) * fs.img_w[img]
        y = (1.0 - min(max(v, 0.0), 1.0)) * fs.img_h[img]
        dist = min(abs(x - round(x)) / fs.img_w[img], abs(y - round(y)) / fs.img_h[img]) * 2.0 ** 23
        best = dist if best is None else min(best, dist)
    return best


def _object_space_ray(fs, sphere, o, d):
    """The ray Sphere::hit receives below the sphere's Translate / RotateY wrappers (outermost first: hitable.rs:411 moves the origin,
    :483-492 rotates origin and direction), in float32 like both implementations."""
    f = np.float32
    chain, x = [], (fs.sph_xform[sphere] if fs.n_xforms and fs.sph_xform else rt_ffi_no_xform())
    while x != rt_ffi_no_xform():
        chain.append(x)
        x = fs.xf_parent[x]
    for x in reversed(chain):
        q = [f(fs.xf_param[4 * x + k]) for k in range(4)]
        if fs.xf_type[x] == 0:
            o = (o - np.array(q[:3], dtype=f)).astype(f)
        else:
            sn, cs = q[0], q[1]
            o = np.array([f(f(cs * o[0]) - f(sn * o[2])), o[1], f(f(sn * o[0]) + f(cs * o[2]))], dtype=f)
            d = np.array([f(f(cs * d[0]) - f(sn * d[2])), d[1], f(f(sn * d[0]) + f(cs * d[2]))], dtype=f)
    return o, d


def rt_ffi_no_xform():
    return 0xFFFFFFFF


def _explain_outliers(rt, orc, renderer, scene, p, img, it, name, px_tol=1e-4, edge_ulps=2.0):
    """Every pixel that differs from the iterative oracle by more than `px_tol` (display units) must be EXPLAINED by one of the
    two places where device and host libm differ in the last ulp; a pixel that is not fails the test.  All samples of such
    a pixel are re-traced bounce by bounce through rt_debug_bounce on BOTH sides (primary rays rebuilt in numpy float32, bit for
    bit); hit, alive flag and scattered direction must agree exactly at every bounce, t and the scattered origin too — except
      * a medium scatter (hitable.rs:560-570: `neg_inv_density * ln(rand)`): t may differ in the last bits (2e-6).  The scatter
        POINT is then an ulp apart, every later comparison on that path sees different inputs and the two paths may part for
        good (cornell_box: about one path in 350 000), so the path is followed no further;
      * an image lookup (texture.rs:183-193 / hitable.rs:65-71: `(u * W) as u32` of a uv that went through acos and atan2):
        where the colours of a segment part, (u W, v H) must lie within `edge_ulps` * 2^-23 of a texel edge in (u, v) — the
        neighbouring texel on one side, a visible, isolated difference that no tolerance on the arithmetic can cover.
    Returns (outlier mask, diverged medium paths, texel-edge lookups)."""
    fin = np.isfinite(it) & np.isfinite(img)
    diff = np.abs(display(np.where(fin, img, 0)) - display(np.where(fin, it, 0))).max(axis=2)
    out = diff > px_tol
    jj, ii = np.nonzero(out)
    assert len(jj) <= max(64, int(1e-3 * out.size)), (name, len(jj))  # isolated pixels, not a region
    if not len(jj):
        return out, 0, 0
    n_prims = scene.flat.n_spheres + scene.flat.n_rects
    pi_, pj_ = np.repeat(ii, p.spp), np.repeat(jj, p.spp)
    o, d, keys = _primary_rays(scene, p, pi_, pj_, np.tile(np.arange(p.spp), len(ii)))
    live = np.ones(len(o), dtype=bool)
    medium = np.zeros(len(o), dtype=bool)
    texel = np.zeros(len(o), dtype=bool)
    worst = 0.0
    for depth in range(p.max_depth + 1):
        idx = np.nonzero(live)[0]
        if not len(idx):
            break
        g = renderer.debug_bounce(o[idx], d[idx], keys[idx], depth=depth)
        c = orc.debug_bounce(scene.flat_ptr, o[idx], d[idx], keys[idx], depth=depth, accel=orc.ACCEL_LIST)
        for k in ("hit", "alive"):
            assert np.array_equal(g[k], c[k]), (name, depth, k)
        assert np.array_equal(g["d"].view(np.uint32), c["d"].view(np.uint32)), (name, depth, "d")
        exact = (g["t"].view(np.uint32) == c["t"].view(np.uint32)) & (g["o"].view(np.uint32) == c["o"].view(np.uint32)).all(axis=1)
        for r in np.nonzero(~exact)[0]:  # an ulp apart: only a medium scatter may be
            assert g["hit"][r] >= n_prims and np.isclose(g["t"][r], c["t"][r], rtol=2e-6), \
                (name, "pixel", int(pi_[idx[r]]), int(pj_[idx[r]]), "depth", depth, int(g["hit"][r]), g["t"][r], c["t"][r])
            medium[idx[r]] = True
        for r in np.nonzero(exact)[0]:
            a_ = np.concatenate([g["radiance"][r], g["attenuation"][r]]).astype(np.float64)
            b_ = np.concatenate([c["radiance"][r], c["attenuation"][r]]).astype(np.float64)
            if np.allclose(a_, b_, rtol=2e-5, atol=1e-6, equal_nan=True):
                continue
            dist = _texel_edge_distance(scene, int(g["hit"][r]), o[idx[r]], d[idx[r]], g["t"][r])
            assert dist is not None and dist <= edge_ulps, (name, "pixel", int(pi_[idx[r]]), int(pj_[idx[r]]), "depth", depth,
                                                            "colours differ away from a texel edge", dist, a_, b_)
            texel[idx[r]] = True
            worst = max(worst, dist)
        alive = g["alive"].astype(bool) & exact
        o[idx[alive]], d[idx[alive]] = g["o"][alive], g["d"][alive]
        live[idx[~alive]] = False
    per_pixel = (medium | texel).reshape(len(ii), p.spp).any(axis=1)
    assert per_pixel.all(), (name, "pixels that differ although none of their paths holds a medium scatter an ulp apart or a texel-edge lookup",
                             list(zip(ii[~per_pixel], jj[~per_pixel]))[:8])
    print(f"{name}: {len(ii)} of {out.size} pixels differ by more than {px_tol}: {int(medium.sum())} paths with a medium scatter point an ulp "
          f"apart, {int(texel.sum())} lookups within {worst:.2f} x 2^-23 of a texel edge")
    return out, int(medium.sum()), int(texel.sum())


def _rays_agree(st, so, scene, p):
    """Ray counts per depth: exact — except in a scene with media, where a path whose scatter point is an ulp apart may part from
    the oracle's (_explain_outliers): there to 1e-4 of the rays, as test_constant_medium_and_cornell_box states it."""
    if st.n_bad_dir or so.n_bad_dir:
        # directions the reference would panic on (main.rs:39: fp32 far from the origin, scripts/gpu_random_scene_sweep.py): the
        # oracle drops such a ray before it counts it, the device counts the closest-hit query it made for it before k_shade drops it
        assert st.n_bad_dir == so.n_bad_dir
        assert abs(int(st.n_rays) - int(st.n_bad_dir) - int(so.n_rays)) <= (max(16, 1e-4 * so.n_rays) if scene.flat.n_media else 0), (st.n_rays, st.n_bad_dir, so.n_rays)
    elif not scene.flat.n_media:
        assert st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    else:
        assert abs(int(st.n_rays) - int(so.n_rays)) <= max(16, 1e-4 * so.n_rays), (st.n_rays, so.n_rays)


def _compare_frames(orc, scene, p, img, ref, name, rt=None, renderer=None):
    """Frame against the oracle when pixels may be non-finite (pbr.rs: a grazing n_dot_i -> 0 divides by ~0, the
    attenuation overflows and inf * 0 = NaN poisons the pixel in the reference's arithmetic too).
    Against the oracle in the wavefront's own product order (EST_ITERATIVE; list walk for scenes with rectangles, whose
    unpadded reference boxes lose grazing hits) the non-finite pixels must be THE SAME pixels and the finite ones agree to
    2e-5 RMSE with no pixel off by more than 1e-4 — in a scene with media or image textures every pixel beyond that is first
    re-traced and explained (_explain_outliers), in any other scene there is none; against the reference's recursive order an
    overflow can strike at a different factor of the chain, so there only the pixels finite in both are compared (RMSE_TOL)
    and the two masks may differ in a 1e-4 fraction of the pixels (one pixel of a small frame) at most."""
    it, _, _ = orc.render(scene.flat_ptr, scene.camera, p, orc.options(rng_mode=orc.RNG_COUNTER, estimator=orc.EST_ITERATIVE, accel=_accel_for(orc, scene)))
    assert np.array_equal(np.isfinite(img), np.isfinite(it)), name
    fin = np.isfinite(it)
    if scene.flat.n_media or scene.flat.n_images:
        assert renderer is not None, "scenes with media or image textures need the renderer to re-trace their outliers"
        fin = fin & ~_explain_outliers(rt, orc, renderer, scene, p, img, it, name)[0][:, :, None]
        ref = np.where(fin, ref, img)  # (the same pixels are set aside against the recursive order below: a neighbouring texel in a
                                       # 96 x 64 frame is 2.6e-4 of RMSE on its own — random scene 1034 of scripts/gpu_random_scene_sweep.py)
    else:
        worst = np.abs(display(np.where(fin, img, 0)) - display(np.where(fin, it, 0))).max(initial=0.0)
        assert worst <= 1e-4, (name, worst)  # nothing to explain them with: no pixel may be off
    e_it = rmse_display(np.where(fin, img, 0), np.where(fin, it, 0))
    assert e_it <= 2e-5, (name, e_it)
    both = np.isfinite(ref) & np.isfinite(img)
    # (one pixel of a 96 x 64 frame is 1.6e-4 of it: random scene 8937 of scripts/gpu_random_scene_sweep.py — the masks are identical
    # against the iterative order above; the recursive order overflows one factor later in one path)
    assert (np.isfinite(ref) != np.isfinite(img)).any(axis=2).sum() <= max(1, int(1e-4 * ref.shape[0] * ref.shape[1])), name
    e = rmse_display(np.where(both, img, 0), np.where(both, ref, 0))
    assert e <= RMSE_TOL, (name, e)


def test_analytic_images(rt, renderer):
    # empty world + gradient sky: closed form of the jittered camera ray
    s = rt.Scene.new()
    s.set_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90, 2.0)
    s.finish()
    renderer.upload(s)
    img, _, st = renderer.render(s.camera, rt.make_params(64, 32, 4, max_depth=5))
    assert st.n_rays == st.n_paths == 64 * 32 * 4
    assert img[-1, :, 2].mean() == pytest.approx(1.0) and np.all(np.diff(img[:, 32, 0]) <= 1e-6)  # bluer towards the top
    assert np.all(img[:, :, 0] <= img[:, :, 1] + 1e-6) and np.all(img[:, :, 1] <= img[:, :, 2] + 1e-6)
    # white furnace: albedo-1 Lambert... use Diffuse albedo 1 inside a constant-1 env: radiance 1 up to depth truncation
    s = rt.Scene.new()
    one = s.constant_tex((1, 1, 1))
    m = s.material(rt._ffi.MAT_EMISSION, tex0=one)
    s.sphere((0, 0, -1), 0.5, m, "lamp")
    s.set_sky(rt._ffi.SKY_BLACK)
    s.set_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90, 2.0)
    s.finish()
    renderer.upload(s)
    img, _, _ = renderer.render(s.camera, rt.make_params(64, 32, 16, max_depth=5))
    assert np.all(img[16, 30:34] == 1.0) and np.all(img[0] == 0.0) and set(np.unique(img[:, :, 0] == img[:, :, 1])) == {True}


def test_full_size_properties_config2_resolution(rt, renderer):
    """1920x1080 (config 2 resolution) at reduced spp: size-independent properties."""
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    p = rt.make_params(1920, 1080, 2, max_depth=50)
    a, _, sa = renderer.render(scene.camera, p)
    b, _, sb = renderer.render(scene.camera, p)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # run-to-run bit reproducible despite atomics in the queue
    assert sa.n_rays == sb.n_rays == sum(sa.rays_per_depth) and sa.rays_per_depth[0] == sa.n_paths == 1920 * 1080 * 2
    assert all(sa.rays_per_depth[d + 1] <= sa.rays_per_depth[d] for d in range(50))
    assert np.isfinite(a).all() and a.min() >= 0.0
    assert sa.bytes_algorithmic == 96 * sa.n_rays + 24 * sa.n_paths + 12 * sa.n_texture_fetches
    # energy: every surface has albedo <= 1 and the only emitter/sky are <= 1
    assert a.max() <= 1.0 + 1e-5


def test_error_behaviour(rt):
    r = rt.Renderer(0)
    scene = rt.Scene.build("test_sphere", 2.0)
    with pytest.raises(rt.RtError, match="no scene"):
        r.render(scene.camera, rt.make_params(8, 8, 1))
    r.upload(scene)
    with pytest.raises(rt.RtError):
        r.render(scene.camera, rt.make_params(0, 8, 1))
    with pytest.raises(rt.RtError):
        r.render(scene.camera, rt.make_params(8, 8, 1, shard_count=2, shard_id=2))
    # a scene with a dangling material index is refused at upload
    import ctypes as C
    fs = rt.RtFlatScene.from_buffer_copy(scene.flat)
    bad = (C.c_uint32 * 2)(0, 7)
    fs.sph_mat = C.cast(bad, C.POINTER(C.c_uint32))
    with pytest.raises(rt.RtError, match="material index"):
        r.upload(fs)
    r.close()


def test_a_wrapper_around_a_medium_must_lie_on_its_boundary_chains(rt):
    """RtFlatScene::med_xform names the innermost wrapper AROUND a medium; the chains of the boundary's primitives run through it.
    A wrapper index beyond the table, or one that no boundary chain passes, is RT_ERR_INVALID — and a NULL med_xform (a host written
    against ABI 10's meaning: no wrappers around media) still uploads."""
    import ctypes as C
    f = rt._ffi
    s = rt.Scene.new()
    fog = s.constant_tex((0.5, 0.5, 0.5))
    m = s.constant_medium(s.translate(s.sphere((0, 0, 0), 1.0, s.material(f.MAT_DIELECTRIC, p=(1.5,)), "b"), (1, 0, 0)), 0.5, fog)
    s.rotate_y(m, 30.0)                                     # chain of the boundary sphere: Translate (inside) -> RotateY (around)
    other = s.translate(s.sphere((5, 0, 0), 1.0, s.material(f.MAT_DIELECTRIC, p=(1.5,)), "o"), (0, 1, 0))
    s.set_camera((0, 0, 10), (0, 0, 0), (0, 1, 0), 30, 1.0)
    scene = s.finish()
    a = scene.arrays()
    assert scene.flat.n_xforms == 3 and a["med_xform"].tolist() == [0] and a["xf_parent"].tolist()[1] == 0  # RotateY was opened first
    r = rt.Renderer(0)
    r.upload(scene)
    for bad, what in ((7, "bad wrapper"), (int(a["sph_xform"][1]), "not on the chain")):
        fs = rt.RtFlatScene.from_buffer_copy(scene.flat)
        arr = (C.c_uint32 * 1)(bad)
        fs.med_xform = C.cast(arr, C.POINTER(C.c_uint32))
        with pytest.raises(rt.RtError, match=what):
            r.upload(fs)
    fs = rt.RtFlatScene.from_buffer_copy(scene.flat)
    fs.med_xform = C.POINTER(C.c_uint32)()
    r.upload(fs)
    r.close()


def test_non_finite_geometry_is_rejected(rt):
    """A NaN / infinite centre, radius, rectangle bound or transform parameter would reach the tree builder's sort
    comparators (host-side undefined behaviour): rt_scene_upload returns RT_ERR_INVALID instead."""
    import ctypes as C
    r = rt.Renderer(0)
    scene = rt.Scene.build("sphere_scene", 2.0)
    for field, bad in (("sph_cx", np.nan), ("sph_r", np.inf), ("sph_cz", -np.inf)):
        fs = rt.RtFlatScene.from_buffer_copy(scene.flat)
        arr = np.ctypeslib.as_array(getattr(fs, field), shape=(fs.n_spheres,)).astype(np.float32, copy=True)
        arr[7] = bad
        setattr(fs, field, arr.ctypes.data_as(C.POINTER(C.c_float)))
        with pytest.raises(rt.RtError, match="finite"):
            r.upload(fs)
    r.close()


def test_config4_and_5_full_resolution_low_spp(rt, orc, renderer):
    """BASELINE.json configs 4 (earthmap + newport_loft env sky) and 5 (pbr.rs sweep) at their full
    1920x1080 resolution and 2 spp against the oracle: exact ray and texel-fetch counts, RMSE in tolerance."""
    for name, depth in (("earth_env_scene", 50), ("pbr_sweep_scene", 50)):
        scene = rt.Scene.build(name, 16 / 9)
        renderer.upload(scene)
        p = rt.make_params(1920, 1080, 2, max_depth=depth)
        img, _, st = renderer.render(scene.camera, p)
        ref, _, so = _oracle(orc, scene, p)
        assert st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth), name
        assert st.n_texture_fetches == so.n_texture_fetches, name
        _compare_frames(orc, scene, p, img, ref, name, rt, renderer)


def test_config3_full_size_sharded_8_ways(rt, orc, renderer):
    """BASELINE.json config 3 as stated: sphere_scene 3840x2160, 1024 spp, depth 50 (8 493 465 600 paths, ~21.6 G rays).
    The unsharded frame and the 8 row-interleaved shards of the 8-GPU run (rendered one after the other on this
    GPU, reassembled like the gather does) are bit-identical, with equal ray counts per depth; and the same frame
    at a size the oracle covers (1 spp) matches it: exact rays per depth, RMSE in tolerance."""
    from ray_tracing_in_one_weekend_amd import shard
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    nx, ny, spp = 3840, 2160, 1024
    full, _, st = renderer.render(scene.camera, rt.make_params(nx, ny, spp, max_depth=50, seed=95))
    assert st.n_paths == 8493465600 and st.n_bad_dir == 0
    assert np.isfinite(full).all() and 2.4 < st.n_rays / st.n_paths < 2.7
    parts, rays, per_depth = [], 0, np.zeros(51, dtype=np.int64)
    for r in range(8):
        im, _, s = renderer.render(scene.camera, rt.make_params(nx, ny, spp, max_depth=50, seed=95, shard_band=8, shard_count=8, shard_id=r))
        parts.append(im)
        rays += s.n_rays
        per_depth += np.array(list(s.rays_per_depth)[:51], dtype=np.int64)
    out = shard.deinterleave(parts, ny, 8, 8)
    assert np.array_equal(out.view(np.uint32), full.view(np.uint32)) and rays == st.n_rays
    assert list(per_depth) == list(st.rays_per_depth)[:51]
    # the oracle on the same 4K frame, 1 spp (8.3 M paths)
    p1 = rt.make_params(nx, ny, 1, max_depth=50, seed=95)
    img, _, s1 = renderer.render(scene.camera, p1)
    ref, _, so = _oracle(orc, scene, p1)
    assert s1.n_rays == so.n_rays and list(s1.rays_per_depth) == list(so.rays_per_depth)
    _compare_frames(orc, scene, p1, img, ref, "config 3 at 1 spp", rt, renderer)


def test_in_library_multi_gpu_entry_points(rt, renderer):
    """rt_multi_* (one process, RCCL gather inside the library).  This box has one GPU, so: (1) the one-device
    RtMulti — ncclCommInitAll, an all_gather over one rank, the de-interleave and the RGB8 quantisation — must
    reproduce rt_render bit for bit; (2) the assembly of an 8-GPU run goes through the same entry point
    (rt_deinterleave_bands) on 8 shards rendered one after the other into the layout the all_gather delivers."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library itself is linked against (device buffers for part 2)
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]

    def dev_zeros(nbytes):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), nbytes) == 0 and hip.hipMemset(p, 0, nbytes) == 0
        return p

    def to_host(p, arr):
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemcpy(arr.ctypes.data_as(ctypes.c_void_p), p, arr.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return arr

    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    nx, ny, band = 320, 181, 8  # 181 rows: the shards differ in size and the band buffers are padded
    p = rt.make_params(nx, ny, 6, max_depth=20, seed=95)
    ref, ref8, st = renderer.render(scene.camera, p, want_rgb8=True)
    m = rt.MultiRenderer([0])
    assert m.n_devices == 1
    m.upload(scene)
    img, rgb8, sm = m.render(scene.camera, p, want_rgb8=True)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(rgb8, ref8)
    assert sm.n_rays == st.n_rays and sm.n_paths == st.n_paths and list(sm.rays_per_depth) == list(st.rays_per_depth)
    with pytest.raises(rt.RtError, match="twice"):
        rt.MultiRenderer([0, 0])
    m.close()
    for world in (8, 3):
        pad = max(renderer.shard_rows(rt.make_params(nx, ny, 6, shard_band=band, shard_count=world, shard_id=r)) for r in range(world))
        band_bytes = pad * nx * 3 * 4
        gathered = dev_zeros(world * band_bytes)
        rays = 0
        for r in range(world):
            ps = rt.make_params(nx, ny, 6, max_depth=20, seed=95, shard_band=band, shard_count=world, shard_id=r)
            rays += renderer.render_device(scene.camera, ps, gathered.value + r * band_bytes).n_rays
        out, out8 = dev_zeros(ny * nx * 3 * 4), dev_zeros(ny * nx * 3)
        renderer.deinterleave_bands(gathered.value, nx, ny, band, world, out.value, out8.value)
        full = to_host(out, np.zeros((ny, nx, 3), np.float32))
        full8 = to_host(out8, np.zeros((ny, nx, 3), np.uint8))
        assert np.array_equal(full.view(np.uint32), ref.view(np.uint32)) and rays == st.n_rays
        assert np.array_equal(full8, ref8)
        for b in (gathered, out, out8):
            assert hip.hipFree(b) == 0


def test_rt_multi_render_n_contexts_on_one_gpu(rt, renderer):
    """rt_multi_render's N > 1 code on a one-GPU box (rt_multi_create_ex, RT_MULTI_COPY_GATHER): n contexts share
    device 0, ONE HOST THREAD PER CONTEXT renders its row-interleaved bands at the same time (main.rs:72-108 is the
    fan-out this replaces), the all_gather is n*n device-to-device copies into the same gathered layout, then the padded
    band buffers (181 rows: shards differ in size) go through k_bands_to_image and the statistics are reduced.  Frames
    and ray counts are those of rt_render, bit for bit, for n = 2 and 3 — a general scene and the sphere-only one, so
    that both families of kernel instantiations (and their process-global LDS attribute) run from several threads."""
    for name, nx, ny, aspect in (("sphere_scene", 320, 181, 16 / 9), ("cornell_box", 181, 181, 1.0)):
        scene = rt.Scene.build(name, aspect)
        renderer.upload(scene)
        p = rt.make_params(nx, ny, 6, max_depth=20, seed=95)
        ref, ref8, st = renderer.render(scene.camera, p, want_rgb8=True)
        for n in (2, 3):
            m = rt.MultiRenderer([0] * n, copy_gather=True)
            assert m.n_devices == n
            m.upload(scene)
            for _ in range(2):  # the second call reuses every buffer
                img, rgb8, sm = m.render(scene.camera, p, want_rgb8=True)
                assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (name, n)
                assert np.array_equal(rgb8, ref8)
                assert sm.n_rays == st.n_rays and sm.n_paths == st.n_paths and list(sm.rays_per_depth) == list(st.rays_per_depth)
                assert sm.n_texture_fetches == st.n_texture_fetches and sm.n_bad_dir == 0
            m.close()
    import ctypes
    lib, handle = rt._ffi.load_gpu_library(), ctypes.c_void_p()
    assert lib.rt_multi_create_ex((ctypes.c_int * 1)(0), 1, 0x80, ctypes.byref(handle)) == -1 and not handle  # RT_ERR_INVALID
    assert b"unknown flag" in lib.rt_multi_last_error(None)


def test_two_host_threads_render_on_two_contexts_at_once(rt):
    """Two contexts on one device, each driven by its own Python thread (ctypes releases the GIL during the call), eight
    renders each, different scenes so that different kernel instantiations are launched side by side: every frame equals
    the one the same context renders alone.  Covers what a multi-threaded host relies on: per-context streams, events and
    error strings, and kernel attributes that are set once per process."""
    import threading
    jobs = []
    for name, nx, ny, aspect in (("sphere_scene", 256, 144, 16 / 9), ("final_scene", 160, 160, 1.0)):
        scene = rt.Scene.build(name, aspect)
        r = rt.Renderer(0)
        r.upload(scene)
        p = rt.make_params(nx, ny, 4, max_depth=12, seed=7)
        ref, _, st = r.render(scene.camera, p)
        jobs.append((r, scene, p, ref, st.n_rays))
    errors = []

    def work(job):
        r, scene, p, ref, n_rays = job
        try:
            for _ in range(8):
                img, _, st = r.render(scene.camera, p)
                assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and st.n_rays == n_rays
        except Exception as e:  # noqa: BLE001 (reported below, on the main thread)
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for r, *_ in jobs:
        r.close()


def test_scene_too_large_for_the_lds_bvh_uses_the_list_walk(rt, orc, renderer):
    """3000 spheres: the BVH (2999 nodes x 64 B) no longer fits LDS, so closest hit falls back to the
    tiled list walk (two LDS tiles); results still equal the oracle's."""
    rng = np.random.default_rng(11)
    s = rt.Scene.new()
    mats = [s.material(rt._ffi.MAT_DIFFUSE, tex0=s.constant_tex(rng.uniform(0.2, 0.9, 3))) for _ in range(8)]
    mats.append(s.material(rt._ffi.MAT_METAL, color=(0.8, 0.8, 0.9), p=(0.1,)))
    mats.append(s.material(rt._ffi.MAT_DIELECTRIC, p=(1.5,)))
    s.sphere((0, -1000, 0), 1000.0, mats[0], "ground")
    for i in range(2999):
        c = rng.uniform(-20, 20, 3)
        c[1] = rng.uniform(0.1, 3.0)
        s.sphere(c, float(rng.uniform(0.05, 0.3)), mats[int(rng.integers(0, len(mats)))], f"s{i}")
    s.set_camera((13, 4, 3), (0, 0.5, 0), (0, 1, 0), 40, 16 / 9)
    s.finish()
    renderer.upload(s)
    p = rt.make_params(160, 90, 2, max_depth=10)
    img, _, st = renderer.render(s.camera, p)
    ref, _, so = _oracle(orc, s, p)
    assert st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    _compare_frames(orc, s, p, img, ref, "3000 spheres, list walk", rt, renderer)
    o, d, keys = rays_on_scene(4000, 21)
    g = renderer.debug_bounce(o, d, keys)
    c = orc.debug_bounce(s.flat_ptr, o, d, keys, accel=orc.ACCEL_LIST)
    assert np.array_equal(g["hit"], c["hit"]) and np.array_equal(g["t"].view(np.uint32), c["t"].view(np.uint32))


def test_russian_roulette_opt_in(rt, orc, renderer):
    """RT_FLAG_RUSSIAN_ROULETTE = the estimator commented out at main.rs:49-53; GPU == oracle with the flag
    (exact ray counts), and the estimate stays unbiased with far fewer rays."""
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    p = rt.make_params(320, 180, 16, max_depth=50, flags=rt._ffi.FLAG_RUSSIAN_ROULETTE)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p)
    it, _, si = _oracle(orc, scene, p, estimator=orc.EST_ITERATIVE)
    assert st.n_rays == so.n_rays == si.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    # (sphere_scene holds an image-textured sphere: pixels with a lookup on a texel edge are shown to be just that and set
    # aside, _explain_outliers; the re-trace follows the path's own draws, which roulette leaves untouched — it takes
    # the counter after them, DESIGN.md "RNG")
    keep = ~_explain_outliers(rt, orc, renderer, scene, p, img, it, "sphere_scene + russian roulette")[0][:, :, None]
    assert rmse_display(np.where(keep, img, 0), np.where(keep, it, 0)) <= 2e-5 and rmse_display(img, ref) <= RMSE_TOL
    plain, _, sp = renderer.render(scene.camera, rt.make_params(320, 180, 16, max_depth=50))
    assert st.n_rays < 0.9 * sp.n_rays
    # unbiased: compare LINEAR means (the gamma/clamp of the display transform is not linear in the noise)
    assert abs(img.mean() - plain.mean()) / plain.mean() < 0.01


def _box_room(rt, env=None):
    """Cornell-box geometry from the reference constructors that are on the accelerated path: the five walls and
    the light of demo_scene.rs:131-136, two axis-aligned GBox (the reference rotates/translates them and fills
    them with smoke: instance transforms and ConstantMedium are not accelerated yet), a glass and a metal sphere."""
    s = rt.Scene.new()
    f = rt._ffi
    red = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.65, 0.05, 0.05)))
    white = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.73, 0.73, 0.73)))
    green = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.12, 0.45, 0.15)))
    light = s.material(f.MAT_EMISSION, tex0=s.constant_tex((7, 7, 7)))
    earth = s.material(f.MAT_LAMBERT, tex0=s.image_tex("res/earthmap.jpg"))
    s.rect(f.RECT_XZ, (113, 554, 127), (443, 554, 432), light)
    s.rect(f.RECT_XY, (0, 0, 555), (555, 555, 555), earth)   # back wall carries the image texture: rect uv
    s.rect(f.RECT_XZ, (0, 0, 0), (555, 0, 555), white)
    s.rect(f.RECT_XZ, (0, 555, 0), (555, 555, 555), white)
    s.rect(f.RECT_YZ, (0, 0, 0), (0, 555, 555), red)
    s.rect(f.RECT_YZ, (555, 0, 0), (555, 555, 555), green)
    s.gbox((265, 0, 295), (430, 330, 460), white)
    s.gbox((130, 0, 65), (295, 165, 230), s.material(f.MAT_OREN_NAYAR, tex0=s.checker_tex((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), p=(0.5,)))
    s.sphere((190, 230, 150), 60.0, s.material(f.MAT_DIELECTRIC, p=(1.5,)), "glass")
    s.sphere((400, 80, 120), 80.0, s.material(f.MAT_METAL, color=(0.8, 0.85, 0.9), p=(0.05,)), "metal")
    s.set_sky(f.SKY_BLACK)
    s.set_camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40, 1.0)
    return s.finish()


def test_rectangles_and_boxes(rt, orc, renderer):
    """hitable.rs:244-402 on the GPU: XY/XZ/YZ rectangles and GBox sides in the LDS BVH and in the list walk,
    rect uv for image textures, every result equal to the oracle's."""
    scene = _box_room(rt)
    assert scene.flat.n_rects == 18 and scene.flat.n_spheres == 2
    renderer.upload(scene)
    rng = np.random.default_rng(5)
    n = 30000
    o = rng.uniform(5, 550, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 10] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 10)] * rng.choice([-1, 1], (n // 10, 1)).astype(np.float32)
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d * (np.float32(1) / ln)[:, None]).astype(np.float32)
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    g = renderer.debug_bounce(o, d, keys)
    b = renderer.debug_bounce(o, d, keys, flags=rt._ffi.FLAG_BRUTE_FORCE)
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, accel=orc.ACCEL_LIST)
    for other in (b, c):
        assert np.array_equal(g["hit"], other["hit"]) and np.array_equal(g["t"].view(np.uint32), other["t"].view(np.uint32))
        assert np.array_equal(g["alive"], other["alive"])
        assert np.array_equal(g["o"].view(np.uint32), other["o"].view(np.uint32))
        assert np.array_equal(g["d"].view(np.uint32), other["d"].view(np.uint32))
    assert np.allclose(g["attenuation"], c["attenuation"], rtol=2e-5, atol=1e-6) and np.allclose(g["radiance"], c["radiance"], rtol=2e-5, atol=1e-6)
    assert (g["hit"] >= 2).mean() > 0.8  # inside the room nearly every ray ends on a rectangle
    p = rt.make_params(200, 200, 16, max_depth=50)
    img, _, st = renderer.render(scene.camera, p)
    # the GPU search returns HitableList::hit's result, so the list-walk oracle is the exact reference
    ref, _, so = _oracle(orc, scene, p, accel=orc.ACCEL_LIST)
    assert st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    assert st.n_texture_fetches == so.n_texture_fetches and st.n_texture_fetches > 0
    _compare_frames(orc, scene, p, img, ref, "box room", rt, renderer)
    # the reference's own BvhNode culls with UNPADDED boxes (`t_max <= t_min` rejects, math.rs:109): a hit within an
    # ulp of a rectangle's edge passes XYRect::hit but not the box around it, ~1e-6 of the rays; those paths differ
    bv, _, sb = _oracle(orc, scene, p, accel=orc.ACCEL_BVH)
    assert abs(int(sb.n_rays) - int(st.n_rays)) / st.n_rays < 1e-4
    assert (np.abs(display(img) - display(bv)).max(axis=2) > 1e-3).mean() < 1e-3
    p.flags = rt._ffi.FLAG_BRUTE_FORCE
    img2, _, st2 = renderer.render(scene.camera, p)
    assert np.array_equal(img.view(np.uint32), img2.view(np.uint32)) and st2.n_rays == st.n_rays
    # DisneyMetal on a rectangle would read the reference's stale HitRecord.tang: refused
    bad = rt.Scene.new()
    m = bad.material(rt._ffi.MAT_DISNEY_METAL, tex0=bad.constant_tex((1, 1, 1)), p=(0.3, 0.5, 0.0))
    bad.rect(rt._ffi.RECT_XY, (0, 0, 0), (1, 1, 0), m)
    bad.set_camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40, 1.0)
    bad.finish()
    with pytest.raises(rt.RtError, match="stale"):
        renderer.upload(bad)


def test_repeated_renders_reuploads_and_contexts(rt):
    """Buffers are reused across calls, scenes can be replaced, two contexts coexist, and nothing leaks."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library itself is linked against

    def free_bytes():
        f, tot = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(tot)) == 0
        return f.value

    a, b = rt.Renderer(0), rt.Renderer(0)
    free0 = free_bytes()
    s1, s2 = rt.Scene.build("sphere_scene", 2.0), rt.Scene.build("simple_light_scene", 2.0)
    a.upload(s1)
    b.upload(s2)
    ref1, _, _ = a.render(s1.camera, rt.make_params(96, 48, 4, max_depth=10))
    ref2, _, _ = b.render(s2.camera, rt.make_params(96, 48, 4, max_depth=10))
    free_mid = None
    for it in range(12):
        nx = 96 if it % 2 == 0 else 64
        i1, _, _ = a.render(s1.camera, rt.make_params(nx, nx // 2, 4, max_depth=10))
        i2, _, _ = b.render(s2.camera, rt.make_params(nx, nx // 2, 4, max_depth=10))
        if nx == 96:
            assert np.array_equal(i1, ref1) and np.array_equal(i2, ref2)
        a.upload(s2 if it % 3 == 2 else s1)  # replace the scene, then put it back
        a.upload(s1)
        if it == 3:
            free_mid = free_bytes()
    assert abs(free_bytes() - free_mid) < (8 << 20)  # steady state: no growth per call
    # two sphere-only scenes of different size in two contexts (same k_intersect instantiation, 51 KB and ~1 KB of
    # tree): the dynamic-LDS attribute is per function and process-wide, so the larger context must keep working after
    # the smaller one has uploaded
    s3 = rt.Scene.build("test_sphere", 2.0)
    b.upload(s3)
    i1, _, _ = a.render(s1.camera, rt.make_params(96, 48, 4, max_depth=10))
    i3, _, _ = b.render(s3.camera, rt.make_params(96, 48, 4, max_depth=10))
    i1b, _, _ = a.render(s1.camera, rt.make_params(96, 48, 4, max_depth=10))
    assert np.array_equal(i1, ref1) and np.array_equal(i1b, ref1) and np.isfinite(i3).all()
    a.close()
    b.close()
    assert free0 - free_bytes() < (64 << 20)  # everything the two contexts allocated is released


def _render_with_the_pool_grown(r, scene, p, limit_s=60.0):
    """Renders until a frame took ONE slice: a context's work-buffer pool is backed by a helper thread, and on a device that is
    busy taking freed memory back a chunk request can wait for seconds (rt_pool.h) — frames rendered meanwhile take more slices
    (same bits), which is the design, not a failure."""
    t0 = time.time()
    while True:
        img, _, st = r.render(scene.camera, p)
        if st.n_slices == 1 or time.time() - t0 > limit_s:
            return img, st
        time.sleep(0.2)


def _hip_memory():
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library itself is linked against
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]

    def free_bytes():
        f, tot = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(tot)) == 0
        return f.value

    def hold(n_bytes, piece=16 << 30):
        """hipMalloc n_bytes in pieces (smaller ones where the device refuses a large one: what hipMemGetInfo calls free is not
        always there in one piece); returns the pointers"""
        out = []
        while n_bytes >= (8 << 20) and piece >= (8 << 20):
            p = ctypes.c_void_p()
            sz = min(piece, n_bytes)
            if hip.hipMalloc(ctypes.byref(p), sz) != 0:
                piece = sz // 2
                continue
            out.append(p)
            n_bytes -= sz
        return out

    def leave_free(target, ptrs, piece=4 << 30):
        """Allocates (appending to ptrs) or gives back pieces until hipMemGetInfo reports about `target` bytes free — its figure
        trails the allocations by a moment, so this converges instead of trusting one reading."""
        for _ in range(40):
            time.sleep(0.05)
            f = free_bytes()
            if f > target + (48 << 20):
                got = hold(min(f - target, piece), piece=min(f - target, piece))
                if not got:
                    piece = max(piece // 2, 8 << 20)
                ptrs += got
            elif f < target - (20 << 20) and ptrs:
                assert hip.hipFree(ptrs.pop()) == 0
                piece = 32 << 20
            else:
                break
        return ptrs

    def release(ptrs):
        for p in ptrs:
            assert hip.hipFree(p) == 0
        ptrs.clear()

    return free_bytes, hold, release, leave_free


@pytest.mark.gpu
def test_render_under_memory_pressure(rt):
    """The work buffers take what the device can give (rt_pool.h): beside a tenant that leaves 20 GB free a 13 GB slice becomes
    several smaller ones and the frame is the same bits; with next to nothing free rt_render returns RT_ERR_NOMEM and says so,
    the same context renders a small frame in what there is and the large one once memory is back, and nothing leaks."""
    free_bytes, hold, release, leave_free = _hip_memory()
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    big = rt.make_params(1920, 1080, 64, max_depth=50, seed=95)
    small = rt.make_params(64, 36, 4, max_depth=50, seed=95)
    r0 = rt.Renderer(0)
    r0.upload(scene)
    ref, _, st0 = r0.render(scene.camera, big)
    ref_small, _, _ = r0.render(scene.camera, small)
    img, st0b = _render_with_the_pool_grown(r0, scene, big)
    assert st0b.n_slices == 1 and np.array_equal(img.view(np.uint32), ref.view(np.uint32))  # a device with memory to give: one slice once the pool has grown
    r0.close()
    free_start = free_bytes()  # (after a first context: what the HIP runtime keeps for itself is there already)
    tenant = []
    r1 = r2 = None
    try:
        # ---- 20 GB free: a context may take half of it
        leave_free(20 << 30, tenant, piece=16 << 30)
        assert abs(free_bytes() - (20 << 30)) < (1 << 30)
        r1 = rt.Renderer(0)
        r1.upload(scene)
        img, _, st1 = r1.render(scene.camera, big)
        assert st1.n_slices > 1 and st1.n_rays == st0.n_rays and list(st1.rays_per_depth) == list(st0.rays_per_depth)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        img, _, st1b = r1.render(scene.camera, big)
        assert st1b.n_slices > 1 and np.array_equal(img.view(np.uint32), ref.view(np.uint32))
        pool_mb = r1.render_parts()["pool_mapped_mb"]
        assert 2048 <= pool_mb <= 10 * 1024 + 512, pool_mb  # half of what was free, never more
        # ---- ~150 MB free: the large frame's small buffers (92 MB) and one sample per pixel (199 MB) do not fit
        squeeze = leave_free(160 << 20, [], piece=4 << 30)
        assert (128 << 20) <= free_bytes() < (330 << 20), free_bytes()
        r2 = rt.Renderer(0)
        r2.upload(scene)
        with pytest.raises(rt.RtError, match=r"\(-3\).*no device memory"):
            r2.render(scene.camera, big)
        img, _, _ = r2.render(scene.camera, small)  # ... but a small frame does (one 128 MB chunk), on the same context
        assert np.array_equal(img.view(np.uint32), ref_small.view(np.uint32))
        release(squeeze)
        img, _, st2 = r2.render(scene.camera, big)  # memory is back: the pool grows again
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and st2.n_rays == st0.n_rays
    finally:
        for r in (r1, r2):
            if r is not None:
                r.close()
        release(tenant)
    assert free_start - free_bytes() < (64 << 20)  # everything is released


@pytest.mark.gpu
def test_first_frame_starts_in_what_is_mapped(rt):
    """A fresh context renders its first frame while the helper thread is still backing the pool: however many slices that
    frame took, it is the frame of the grown pool bit for bit; with rt_prepare before the scene is built the pool is there
    when the frame starts; a slice size the caller asked for is honoured whatever the pool's state."""
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    p = rt.make_params(1920, 1080, 96, max_depth=50, seed=95)
    r = rt.Renderer(0)
    r.upload(scene)
    first, _, st1 = r.render(scene.camera, p)
    parts = r.render_parts()
    second, st2 = _render_with_the_pool_grown(r, scene, p)
    assert st2.n_slices == 1 and 1 <= st1.n_slices <= 32
    assert np.array_equal(first.view(np.uint32), second.view(np.uint32)) and st1.n_rays == st2.n_rays
    assert {"small_buffers_and_pool_request", "queue_probe", "enqueue_all_launches", "wait_for_the_device", "pool_mapped_mb"} <= set(parts)
    r.close()
    # a device that hands out memory slowly (test hook: 0.75 ms per 128 MB chunk, 120 ms for this frame's 20 GB): the frame starts in
    # what has arrived and is done long before the pool is — several slices, the same bits
    r = rt.Renderer(0)
    r.set_option("pool_chunk_delay_us", 750)
    r.upload(scene)
    t0 = time.perf_counter()
    slow, _, st = r.render(scene.camera, p)
    wall_ms = (time.perf_counter() - t0) * 1e3
    assert 2 <= st.n_slices <= 32 and np.array_equal(slow.view(np.uint32), second.view(np.uint32)) and st.n_rays == st2.n_rays
    parts = r.render_parts()
    # (the clock holds unless the driver itself made a chunk request wait meanwhile — it does that for seconds now and then)
    assert (parts["of_which_waiting_for_the_pool"] < 40.0 and wall_ms < 150.0) or parts["pool_slowest_chunk_ms"] > 20.0, (parts, wall_ms)
    r.set_option("pool_chunk_delay_us", 0)
    again, st = _render_with_the_pool_grown(r, scene, p)
    assert st.n_slices == 1 and np.array_equal(again.view(np.uint32), second.view(np.uint32))
    r.close()
    r = rt.Renderer(0)
    r.prepare(p)
    scene2 = rt.Scene.build("sphere_scene", 16 / 9)  # (the host's scene build: the pool grows meanwhile)
    r.upload(scene2)
    img, _, st = r.render(scene.camera, p)
    assert np.array_equal(img.view(np.uint32), second.view(np.uint32))
    r.close()
    r = rt.Renderer(0)
    r.upload(scene)
    img, _, st = r.render(scene.camera, rt.make_params(1920, 1080, 96, max_depth=50, seed=95, spp_slice=40))
    assert st.n_slices == 3 and np.array_equal(img.view(np.uint32), second.view(np.uint32))
    r.close()


def _cornell_with_instances(rt):
    """demo_scene.rs:112-148 without the smoke: the two boxes are RotateY + Translate instances of GBox
    (hitable.rs:404-520) used as solid white boxes, plus a translated/rotated sphere."""
    s = rt.Scene.new()
    f = rt._ffi
    red = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.65, 0.05, 0.05)))
    white = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.73, 0.73, 0.73)))
    green = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.12, 0.45, 0.15)))
    light = s.material(f.MAT_EMISSION, tex0=s.constant_tex((7, 7, 7)))
    s.rect(f.RECT_XZ, (113, 554, 127), (443, 554, 432), light)
    s.rect(f.RECT_XY, (0, 0, 555), (555, 555, 555), white)
    s.rect(f.RECT_XZ, (0, 0, 0), (555, 0, 555), white)
    s.rect(f.RECT_XZ, (0, 555, 0), (555, 555, 555), white)
    s.rect(f.RECT_YZ, (0, 0, 0), (0, 555, 555), red)
    s.rect(f.RECT_YZ, (555, 0, 0), (555, 555, 555), green)
    b1 = s.gbox((0, 0, 0), (165, 330, 165), white)
    s.rotate_y(b1, 15.0)
    s.translate(b1, (265, 0, 295))
    b2 = s.gbox((0, 0, 0), (165, 165, 165), s.material(f.MAT_LAMBERT, tex0=s.image_tex("res/earthmap.jpg")))
    s.rotate_y(b2, -18.0)
    s.translate(b2, (130, 0, 65))
    sp = s.sphere((0, 0, 0), 50.0, s.material(f.MAT_DIELECTRIC, p=(1.5,)), "glass")
    s.translate(sp, (30, 215, 40))
    s.rotate_y(sp, 30.0)       # nesting the other way round: RotateY(Translate(sphere))
    s.translate(sp, (200, 0, 60))
    s.set_sky(f.SKY_BLACK)
    s.set_camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40, 1.0)
    return s.finish()


def test_translate_and_rotate_y_instances(rt, orc, renderer):
    scene = _cornell_with_instances(rt)
    a = scene.arrays()
    assert scene.flat.n_xforms == 7 and a["xf_type"].tolist() == [0, 1, 0, 1, 0, 1, 0]
    assert a["xf_parent"].tolist() == [rt._ffi.NO_XFORM, 0, rt._ffi.NO_XFORM, 2, rt._ffi.NO_XFORM, 4, 5]
    renderer.upload(scene)
    rng = np.random.default_rng(9)
    n = 30000
    o = rng.uniform(5, 550, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d * (np.float32(1) / ln)[:, None]).astype(np.float32)
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    g = renderer.debug_bounce(o, d, keys)
    b = renderer.debug_bounce(o, d, keys, flags=rt._ffi.FLAG_BRUTE_FORCE)
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, accel=orc.ACCEL_LIST)
    for other in (b, c):
        assert np.array_equal(g["hit"], other["hit"]) and np.array_equal(g["t"].view(np.uint32), other["t"].view(np.uint32))
        assert np.array_equal(g["alive"], other["alive"])
        assert np.array_equal(g["o"].view(np.uint32), other["o"].view(np.uint32))
        assert np.array_equal(g["d"].view(np.uint32), other["d"].view(np.uint32))
    assert np.allclose(g["attenuation"], c["attenuation"], rtol=2e-5, atol=1e-6)
    inst = (g["hit"] == 0) | (g["hit"] >= 1 + 6)  # the wrapped sphere (prim 0) or a side of a wrapped box
    assert inst.mean() > 0.15
    p = rt.make_params(160, 160, 16, max_depth=50)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p, accel=orc.ACCEL_LIST)
    assert st.n_rays == so.n_rays and list(st.rays_per_depth) == list(so.rays_per_depth)
    assert st.n_texture_fetches == so.n_texture_fetches
    _compare_frames(orc, scene, p, img, ref, "cornell box with instances", rt, renderer)


def test_constant_medium_and_cornell_box(rt, orc, renderer):
    """hitable.rs:523-588 on the GPU and the reference's cornell_box (demo_scene.rs:112-148).  A medium's scatter
    distance goes through ln(), where device and host libm differ in the last ulp, so medium hits agree to 1e-6
    relative instead of bit for bit, and downstream ray counts to 1e-4."""
    scene = rt.Scene.build("cornell_box", 1.0)
    a = scene.arrays()
    assert scene.flat.n_media == 2 and scene.flat.n_rects == 18 and a["med_neg_inv_density"].tolist() == [-100.0, -100.0]
    assert a["rect_medium"].tolist() == [rt._ffi.NO_XFORM] * 6 + [0] * 6 + [1] * 6
    renderer.upload(scene)
    rng = np.random.default_rng(13)
    n = 40000
    o = rng.uniform(5, 550, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d * (np.float32(1) / ln)[:, None]).astype(np.float32)
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    g = renderer.debug_bounce(o, d, keys, depth=3)
    b = renderer.debug_bounce(o, d, keys, depth=3, flags=rt._ffi.FLAG_BRUTE_FORCE)
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=3, accel=orc.ACCEL_LIST)
    for k in g:  # BVH search == list walk on the device, bit for bit, media included
        assert np.array_equal(g[k].view(np.uint8), b[k].view(np.uint8)), k
    assert np.array_equal(g["hit"], c["hit"]) and np.array_equal(g["alive"], c["alive"])
    med = g["hit"] >= 18
    assert 0.05 < med.mean() < 0.6                       # a good share of the rays scatters inside the smoke
    assert np.array_equal(g["t"][~med].view(np.uint32), c["t"][~med].view(np.uint32))
    assert np.allclose(g["t"][med], c["t"][med], rtol=2e-6)
    assert np.allclose(g["o"], c["o"], rtol=1e-5, atol=1e-4) and np.array_equal(g["d"].view(np.uint32), c["d"].view(np.uint32))
    assert np.allclose(g["attenuation"], c["attenuation"], rtol=2e-5, atol=1e-6)
    p = rt.make_params(200, 200, 16, max_depth=50)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p, accel=orc.ACCEL_LIST)
    _rays_agree(st, so, scene, p)
    _compare_frames(orc, scene, p, img, ref, "cornell_box", rt, renderer)  # every pixel to 1e-4 unless re-traced and explained
    # sharding and slicing stay bit-invariant with media (the medium draw is keyed like every other draw)
    from ray_tracing_in_one_weekend_amd import shard
    parts = [renderer.render(scene.camera, rt.make_params(200, 200, 16, max_depth=50, shard_band=8, shard_count=2, shard_id=r, spp_slice=5))[0]
             for r in range(2)]
    assert np.array_equal(shard.deinterleave(parts, 200, 8, 2).view(np.uint32), img.view(np.uint32))


@pytest.mark.gpu
def test_shared_reciprocal_division_is_ieee_division(rt, renderer):
    """csrc/rt_device.h div_shared: the roots of a ray are divided by |d|^2 (hitable.rs:85-89) and the components of a sphere's
    normal by its radius (hitable.rs:95) with the compiler's own fp32 division sequence, the refined reciprocal of the divisor
    computed once and the operand scaling left out.  Bit for bit IEEE division (numpy float32) wherever the hardware's
    v_div_scale would not have scaled — a normal divisor below 2^126, a numerator of at least 2^-103 (or zero, infinite, NaN),
    a quotient between 2^-126 and 2^96 — which covers every quotient a kernel can take a result from: unit directions, radii of
    ordinary size, roots that have to pass `t >= 1e-3`.  Outside that range (mapped in the last block; no kernel divides there) it is wrong."""
    f = np.float32
    rng = np.random.default_rng(7)
    n = 3_000_000

    def bits_equal(q, ref):
        both_nan = np.isnan(q) & np.isnan(ref)
        return (q.view(np.uint32) == ref.view(np.uint32)) | both_nan

    def pow2(lo, hi, size):  # random sign, exponent in [lo, hi), mantissa uniform
        return (rng.choice([-1.0, 1.0], size) * rng.uniform(1.0, 2.0, size) * 2.0 ** rng.integers(lo, hi, size)).astype(f)

    with np.errstate(all="ignore"):
        # (1) the divisor of a ray's roots, |d|^2 of a unit direction; numerators of any size the discriminant can produce
        a = (1.0 + rng.uniform(-3e-6, 3e-6, n)).astype(f)
        x = pow2(-100, 100, n)
        assert bits_equal(renderer.debug_shared_division(x, a), x / a).all()
        # (2) the divisor of a normal: a radius (either sign), numerators p - c down to an ulp of a coordinate and exactly zero
        a = pow2(-20, 20, n)
        x = pow2(-60, 30, n)
        x[: n // 100] = 0.0
        x[n // 100: n // 50] = -0.0
        assert bits_equal(renderer.debug_shared_division(x, a), x / a).all()
        # (3) the whole range the sequence is exact on, exponents at random
        ea = rng.integers(-120, 126, n)
        ex = np.clip(ea + rng.integers(-124, 95, n), -102, 126)
        keep = (ex - ea > -125) & (ex - ea < 95)
        a = (rng.choice([-1.0, 1.0], n) * rng.uniform(1.0, 2.0, n) * 2.0 ** ea).astype(f)[keep]
        x = (rng.choice([-1.0, 1.0], n) * rng.uniform(1.0, 2.0, n) * 2.0 ** ex).astype(f)[keep]
        q, ref = renderer.debug_shared_division(x, a), x / a
        ok = bits_equal(q, ref) | ~np.isfinite(ref) | (np.abs(ref) < f(2.0 ** -126))  # (a quotient that rounds into the edges)
        assert ok.all(), (int((~ok).sum()), x[~ok][:4], a[~ok][:4], q[~ok][:4], ref[~ok][:4])
        # (4) the special values v_div_fixup takes care of, against every kind of divisor
        sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -3.5, 1e-30, 3e38], dtype=f)
        x, a = (v.ravel() for v in np.meshgrid(sp, np.concatenate([sp[:-1], np.array([1e30], dtype=f)])))  # (a divisor of 3e38 is case 5)
        q, ref = renderer.debug_shared_division(x, a), x / a
        assert bits_equal(q, ref).all(), (x[~bits_equal(q, ref)], a[~bits_equal(q, ref)], q[~bits_equal(q, ref)], ref[~bits_equal(q, ref)])
        # (5) every pair of exponents: it differs from IEEE ONLY where the hardware would have scaled its operands (no kernel divides there)
        a = pow2(-149, 128, n)
        x = pow2(-149, 128, n)
        q, ref = renderer.debug_shared_division(x, a), x / a
        ea, ex = np.frexp(a)[1] - 1, np.frexp(x)[1] - 1  # |v| = m 2^e, 1 <= m < 2
        scaled = (ea < -126) | (ea >= 126) | (ex < -103) | (ex - ea >= 95) | (ex - ea <= -125)
        differ = ~bits_equal(q, ref)
        print(f"all exponents: {int(differ.sum())} of {n} random pairs differ from IEEE, all of them among the {int(scaled.sum())} with a denormal "
              f"or >= 2^126 divisor, a numerator below 2^-103 or a quotient beyond 2^95 / below 2^-125")
        assert not (differ & ~scaled).any(), (x[differ & ~scaled][:4], a[differ & ~scaled][:4])


@pytest.mark.gpu
def test_square_root_without_argument_scaling_is_ieee(rt, renderer):
    """csrc/rt_device.h sqrt_ns: v_sqrt_f32 and the compiler's own correction (both neighbours tried with exact residuals) without
    the 2^32 scaling it puts around arguments below 2^-96.  Bit for bit numpy's correctly rounded sqrt for every argument from 2^-96
    up — all 2^23 mantissas of an even and an odd exponent among them — and for zeros, infinities, NaN and negative numbers."""
    f = np.float32
    rng = np.random.default_rng(11)
    with np.errstate(all="ignore"):
        for e in (0, 1, -95, -96, 126, 127):  # every mantissa of these exponents (2^-96 = the first argument the hardware would not scale)
            x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32((e + 127) << 23)).view(f)
            q, ref = renderer.debug_sqrt(x), np.sqrt(x)
            assert np.array_equal(q.view(np.uint32), ref.view(np.uint32)), e
        x = (rng.uniform(1.0, 2.0, 4_000_000) * 2.0 ** rng.integers(-96, 128, 4_000_000)).astype(f)
        assert np.array_equal(renderer.debug_sqrt(x).view(np.uint32), np.sqrt(x).view(np.uint32))
        sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, -1e-30], dtype=f)
        q, ref = renderer.debug_sqrt(sp), np.sqrt(sp)
        assert np.array_equal(np.isnan(q), np.isnan(ref)) and np.array_equal(q[~np.isnan(q)].view(np.uint32), ref[~np.isnan(ref)].view(np.uint32))
        # below 2^-96 (no kernel takes such a root): mapped, not asserted
        x = (rng.uniform(1.0, 2.0, 1_000_000) * 2.0 ** rng.integers(-149, -96, 1_000_000)).astype(f)
        q, ref = renderer.debug_sqrt(x), np.sqrt(x)
        print(f"arguments below 2^-96: {int((q.view(np.uint32) != ref.view(np.uint32)).sum())} of {len(x)} differ from IEEE")


@pytest.mark.gpu
def test_float_to_integer_conversions_follow_the_rust_rule(rt, renderer):
    """`f as i32` / `f as u32` in Rust: toward zero, saturating at the ends of the range, NaN -> 0 (math.rs:137-152 converts
    n * 256 for the origin offset, texture.rs:183-193 converts u * W for the texel index).  The kernels use v_cvt_i32_f32 /
    v_cvt_u32_f32, which do exactly that by themselves; held here over every kind of argument."""
    f = np.float32
    rng = np.random.default_rng(3)
    x = np.concatenate([
        (rng.normal(size=2_000_000) * 10.0 ** rng.uniform(-3, 12, 2_000_000)).astype(f),
        np.array([0.0, -0.0, 0.5, -0.5, 0.99999994, -0.99999994, 1.0, -1.0, 255.99, 256.0, -256.0, 2147483520.0, 2147483648.0, -2147483648.0,
                  -2147483904.0, 4294967040.0, 4294967296.0, 1e20, -1e20, np.inf, -np.inf, np.nan, -np.nan, 1e-40, -1e-40], dtype=f)])
    with np.errstate(all="ignore"):
        t = np.trunc(np.where(np.isnan(x), 0.0, x).astype(np.float64))
        want_i = np.clip(t, -2147483648.0, 2147483647.0).astype(np.int64).astype(np.int32)
        want_u = np.clip(t, 0.0, 4294967295.0).astype(np.int64).astype(np.uint32)
    assert np.array_equal(renderer.debug_to_i32(x), want_i)
    assert np.array_equal(renderer.debug_to_u32(x), want_u)


@pytest.mark.gpu
def test_hbm_resident_bvh_matches_lds(rt, renderer):
    """Scenes whose tree exceeds LDS traverse it out of HBM/L2; forcing that path on sphere_scene must give the
    LDS path's frame bit for bit (same tree, same traversal order, same tie rule)."""
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    p = rt.make_params(240, 135, 8, max_depth=50)
    renderer.upload(scene)
    a, _, sa = renderer.render(scene.camera, p)
    renderer.set_option("tree_placement", 1)
    renderer.upload(scene)
    assert not renderer.scene_info()["tree_in_lds"] and not renderer.scene_info()["grid"]
    b, _, sb = renderer.render(scene.camera, p)
    renderer.set_option("tree_placement", 0)
    renderer.upload(scene)
    assert renderer.scene_info()["tree_in_lds"]
    assert int(sa.n_rays) == int(sb.n_rays) and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell_box", "final_scene", "random_1003", "random_1015", "random_1023", "random_1031", "random_1041", "random_1044"])
def test_medium_boundary_one_evaluation_matches_two_searches(rt, renderer, name):
    """ConstantMedium::hit (hitable.rs:541-552) searches its boundary twice; the kernels answer both searches from one
    evaluation of the boundary where it is a box or a sphere (medium_root, rt_kernels.h).  Forcing the two searches the
    reference makes must give the same bounce records and the same frame bit for bit."""
    scene = _random_scene(rt, int(name[7:])) if name.startswith("random_") else rt.Scene.build(name, 1.0)
    rng = np.random.default_rng(5)
    n = 40000
    lo, hi = (-12, 12) if name.startswith("random_") else (-100, 650)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 40, 1] = 0.0                                        # rays inside an axis plane: infinite / NaN plane distances
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    p = rt.make_params(160, 160, 8, max_depth=50)
    out = []
    for two_searches in (0, 1):
        renderer.set_option("medium_search", two_searches)
        try:
            renderer.upload(scene)
            g = renderer.debug_bounce(o, d, keys, depth=1)
            img, _, st = renderer.render(scene.camera, p)
        finally:
            renderer.set_option("medium_search", 0)
        out.append((g, img, int(st.n_rays)))
    (g0, i0, n0), (g1, i1, n1) = out
    n_prims = scene.flat.n_spheres + scene.flat.n_rects
    assert scene.flat.n_media > 0 and (g0["hit"] >= n_prims).sum() > 20, "the scene's media are not reached"
    for k in g0:
        assert np.array_equal(g0[k].view(np.uint8), g1[k].view(np.uint8)), k
    assert n0 == n1 and np.array_equal(i0.view(np.uint32), i1.view(np.uint32))


@pytest.mark.gpu
def test_final_scene(rt, orc, renderer):
    """demo_scene.rs:150-221 — every hitable kind at once (instanced 1000-sphere cloud, 400 boxes, a sphere-bounded
    medium, image + Perlin textures, glass, fuzzy metal, BurleyDiffuse).  3.4 k primitives do not fit LDS, so this
    runs on the HBM-resident tree.  Checked against the list-walk oracle like cornell_box."""
    scene = rt.Scene.build("final_scene", 1.0)
    renderer.upload(scene)
    rng = np.random.default_rng(17)
    n = 30000
    o = rng.uniform(-200, 600, size=(n, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(120, 540, size=n).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d * (np.float32(1) / ln)[:, None]).astype(np.float32)
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    g = renderer.debug_bounce(o, d, keys, depth=2)
    b = renderer.debug_bounce(o, d, keys, depth=2, flags=rt._ffi.FLAG_BRUTE_FORCE)
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=2, accel=orc.ACCEL_LIST)
    for k in g:  # tree search == list walk on the device, bit for bit
        assert np.array_equal(g[k].view(np.uint8), b[k].view(np.uint8)), k
    assert np.array_equal(g["hit"], c["hit"]) and np.array_equal(g["alive"], c["alive"])
    n_prims = scene.flat.n_spheres + scene.flat.n_rects
    med = g["hit"] >= n_prims       # a scatter inside medium m reports n_prims + m
    hit = g["hit"] >= 0
    assert hit.mean() > 0.3 and len(np.unique(g["hit"][hit])) > 500
    solid = hit & ~med
    assert np.array_equal(g["t"][solid].view(np.uint32), c["t"][solid].view(np.uint32))
    assert np.allclose(g["t"][hit], c["t"][hit], rtol=2e-6)
    assert np.allclose(g["attenuation"], c["attenuation"], rtol=2e-5, atol=1e-6)
    p = rt.make_params(160, 160, 8, max_depth=50)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p, accel=orc.ACCEL_LIST)
    _rays_agree(st, so, scene, p)
    _compare_frames(orc, scene, p, img, ref, "final_scene", rt, renderer)  # medium scatters an ulp apart AND texel-edge lookups, each re-traced
    assert abs(img.mean() - ref.mean()) / ref.mean() < 1e-3


@pytest.mark.gpu
def test_progressive_preview_and_driver(rt, renderer, tmp_path):
    """rt_set_progress (main.rs:114-123): previews are the running mean, quantised and flipped like the final image;
    the final image does not depend on whether a callback is registered.  Then the main()-equivalent CLI."""
    scene = rt.Scene.build("test_sphere", 2.0)
    renderer.upload(scene)
    base = dict(max_depth=50, spp_slice=4)
    ref, ref8, _ = renderer.render(scene.camera, rt.make_params(160, 80, 12, **base), want_rgb8=True)
    seen = []
    renderer.set_progress(lambda done, total, rgb8: seen.append((done, total, rgb8)))
    img, img8, _ = renderer.render(scene.camera, rt.make_params(160, 80, 12, **base), want_rgb8=True)
    renderer.set_progress(None)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(img8, ref8)
    assert [(d, t) for d, t, _ in seen] == [(4, 12), (8, 12)]       # every slice but the last
    # a preview after 4 samples is exactly the 4-spp image (samples are keyed by index, not by slice)
    _, four8, _ = renderer.render(scene.camera, rt.make_params(160, 80, 4, **base), want_rgb8=True)
    assert seen[0][2].shape == (80, 160, 3) and np.array_equal(seen[0][2], four8)
    again = []
    renderer.render(scene.camera, rt.make_params(160, 80, 12, **base))
    assert again == [] and len(seen) == 2                              # removed callbacks stay removed
    from PIL import Image
    from ray_tracing_in_one_weekend_amd import render as driver
    out = tmp_path / "frame.png"
    assert driver.main(["--scene", "test_sphere", "--nx", "160", "--ny", "80", "--spp", "12", "--preview-every", "4",
                        "--out", str(out)]) == 0
    assert np.array_equal(np.asarray(Image.open(out)), ref8)


@pytest.mark.gpu
def test_primary_candidate_lists_do_not_change_images(rt, renderer):
    """k_primary_lists: depth 0 tests the entries listed for the pixel instead of walking the tree.  The lists are
    conservative and the closest-hit rule is order-independent, so frames are bit-identical with and without them —
    on every mirrored scene, for a sharded frame, and with the eye inside a primitive's bounding sphere."""
    cases = [("sphere_scene", 16 / 9, 320, 180), ("simple_light_scene", 16 / 9, 320, 180), ("cornell_box", 1.0, 200, 200),
             ("final_scene", 1.0, 200, 200), ("earth_env_scene", 16 / 9, 320, 180), ("pbr_sweep_scene", 16 / 9, 320, 180)]
    for name, aspect, nx, ny in cases:
        scene = rt.Scene.build(name, aspect)
        renderer.upload(scene)
        p = rt.make_params(nx, ny, 8, max_depth=12)
        a, _, sa = renderer.render(scene.camera, p)
        renderer.set_option("primary_lists", 1)
        b, _, sb = renderer.render(scene.camera, p)
        renderer.set_option("primary_lists", 0)
        assert int(sa.n_rays) == int(sb.n_rays), name
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
    # sharded rows use local pixel indices for the lists
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    full, _, _ = renderer.render(scene.camera, rt.make_params(320, 180, 8, max_depth=12))
    from ray_tracing_in_one_weekend_amd import shard
    parts = [renderer.render(scene.camera, rt.make_params(320, 180, 8, max_depth=12, shard_band=8, shard_count=3, shard_id=r))[0]
             for r in range(3)]
    assert np.array_equal(shard.deinterleave(parts, 180, 8, 3).view(np.uint32), full.view(np.uint32))
    # the eye inside a glass ball, a wall of small spheres behind it (many candidates per pixel -> overflow -> tree)
    s = rt.Scene.new()
    glass = s.material(rt._ffi.MAT_DIELECTRIC, p=(1.5, 0, 0, 0))
    red = s.material(rt._ffi.MAT_DIFFUSE, tex0=s.constant_tex((0.8, 0.2, 0.2)))
    s.sphere((0, 0, 0), 2.0, glass, "around the eye")
    rng = np.random.default_rng(5)
    for c in rng.uniform(-3, 3, size=(300, 2)):
        s.sphere((float(c[0]), float(c[1]), -8.0 - float(rng.uniform(0, 4))), 0.5, red, "wall")
    s.set_camera((0, 0, 0.5), (0, 0, -8), (0, 1, 0), 40, 1.5)
    s.finish()
    renderer.upload(s)
    p = rt.make_params(150, 100, 8, max_depth=8)
    a, _, _ = renderer.render(s.camera, p)
    renderer.set_option("primary_lists", 1)
    b, _, _ = renderer.render(s.camera, p)
    renderer.set_option("primary_lists", 0)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.std() > 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [494, 110, 205, 28, 206])
def test_candidate_lists_of_a_camera_far_from_the_origin(rt, renderer, seed):
    """camera.rs:43-46 in fp32 for a camera 1e5 units from the origin with the reference's unit focal length: `llc + u H + v V -
    origin` is quantised to steps the size of a pixel, so a sample's ray can point outside the cone through the footprint's
    corners.  k_primary_lists allows for the rounding (the cone opens, the list overflows, the pixel's rays use the tree):
    lists == tree == list walk bit for bit.  The scenes are those of scripts/gpu_grid_fuzz.py with 500 spheres of scale 250
    centred ~1e5 units out; seed 494 is the one that found it (a listed pixel had lost a grazing sphere 2 300 units away)."""
    from helpers import grid_fuzz_scene
    s, _, n, scale, _, _, centre, _, _, _ = grid_fuzz_scene(rt, seed)
    assert n == 500 and scale == 250.0 and np.abs(centre).max() > 100.0
    renderer.upload(s)
    p = rt.make_params(96, 64, 4, max_depth=12, seed=seed)
    a, _, sa = renderer.render(s.camera, p)
    renderer.set_option("primary_lists", 1)
    b, _, sb = renderer.render(s.camera, p)
    renderer.set_option("primary_lists", 0)
    c, _, sc = renderer.render(s.camera, rt.make_params(96, 64, 4, max_depth=12, seed=seed, flags=rt._ffi.FLAG_BRUTE_FORCE))
    assert a.std() > 0.01
    assert list(sa.rays_per_depth) == list(sb.rays_per_depth) == list(sc.rays_per_depth)
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32)) and np.array_equal(b.view(np.uint32), c.view(np.uint32))


@pytest.mark.gpu
def test_depth0_closest_hit_launch_is_decided_inside_every_frame(rt):
    """Depth 0 of a sphere-only scene launches its closest-hit kernel only when some pixel's candidate list overflowed, and
    rt_render decides that inside the frame (the count k_primary_lists leaves is read back before anything else of the frame
    is enqueued) — so the first frame of a view is the same work as any later one, and nothing carries over from frame to frame:
    view A twice, view B, A again, then another scene under the same camera.  Every frame equals the frame of a fresh context bit
    for bit, equals the frame without candidate lists, has the same rays per depth, and its launch count says whether the
    closest-hit launch was skipped (RtStats.n_trace_launches: two per depth, less that one).  Once with overflowing lists, once
    without."""
    def wall(n_behind, seed):
        s = rt.Scene.new()
        red = s.material(rt._ffi.MAT_DIFFUSE, tex0=s.constant_tex((0.8, 0.2, 0.2)))
        mir = s.material(rt._ffi.MAT_METAL, color=(0.8, 0.8, 0.9), p=(0.05,))
        s.sphere((0, -1000, 0), 1000.0, red, "ground")
        rng = np.random.default_rng(seed)
        for k in range(40):
            s.sphere((float(rng.uniform(-4, 4)), 0.3, float(rng.uniform(-4, 4))), 0.3, mir if k % 3 else red, "small")
        for k in range(n_behind):  # a column of spheres behind each other along the view axis: more than 7 candidates per pixel
            s.sphere((0.0, 1.0, -2.0 - 1.5 * k), 0.6, red, "column")
        s.set_camera((0, 1.0, 6), (0, 1.0, 0), (0, 1, 0), 35, 1.5)
        s.finish()
        return s

    max_depth = 10
    full = 2 * (max_depth + 1)
    for n_behind, overflow in ((0, False), (12, True)):
        sa, sb = wall(n_behind, 3), wall(n_behind, 4)
        cam_a = sa.camera
        other = rt.Scene.new()
        other.set_camera((3, 2.0, 5), (0, 0.5, 0), (0, 1, 0), 40, 1.5)
        other.finish()
        cam_b = other.camera
        p = rt.make_params(150, 100, 8, max_depth=max_depth)

        def fresh(scene, cam, lists=True):
            r = rt.Renderer(0)
            if not lists:
                r.set_option("primary_lists", 1)
            r.upload(scene)
            img, _, st = r.render(cam, p)
            r.close()
            return img, st

        r = rt.Renderer(0)
        r.upload(sa)
        seq = [(sa, cam_a), (sa, cam_a), (sa, cam_b), (sa, cam_a), (sb, cam_a), (sb, cam_a)]
        current = sa
        for k, (scene, cam) in enumerate(seq):
            if scene is not current:
                r.upload(scene)
                current = scene
            img, _, st = r.render(cam, p)
            ref, st_ref = fresh(scene, cam)
            plain, st_plain = fresh(scene, cam, lists=False)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (n_behind, k)
            assert np.array_equal(img.view(np.uint32), plain.view(np.uint32)), (n_behind, k)
            assert list(st.rays_per_depth) == list(st_ref.rays_per_depth) == list(st_plain.rays_per_depth), (n_behind, k)
            assert st.n_trace_launches == st_ref.n_trace_launches, (n_behind, k)
            assert st_plain.n_trace_launches == full
            if cam is cam_a:  # (view B looks along the column from the side: whether it overflows is not the point)
                assert st.n_trace_launches == (full if overflow else full - 1), (n_behind, k, st.n_trace_launches)
        r.close()


@pytest.mark.gpu
def test_config2_full_size_three_searches_agree(rt, renderer):
    """BASELINE config 2 at full size (1920x1080, 256 spp, depth 50): candidate lists, tree and list walk give the same
    frame bit for bit.  Ray counts may differ by the handful of grazing rays for which fp32 Sphere::hit reports a hit
    outside the padded box (2 of 1.35e9 on this frame) — lists == list walk exactly."""
    scene = rt.Scene.build("sphere_scene", 16 / 9)
    renderer.upload(scene)
    p = rt.make_params(1920, 1080, 256, max_depth=50)
    a, _, sa = renderer.render(scene.camera, p)
    renderer.set_option("primary_lists", 1)
    b, _, sb = renderer.render(scene.camera, p)
    renderer.set_option("primary_lists", 0)
    c, _, sc = renderer.render(scene.camera, rt.make_params(1920, 1080, 256, max_depth=50, flags=rt._ffi.FLAG_BRUTE_FORCE))
    assert sa.n_paths == 1920 * 1080 * 256 and 1 <= sa.n_slices <= 32  # one slice when HBM has 62 GB to give and the pool has grown
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32)) and np.array_equal(b.view(np.uint32), c.view(np.uint32))
    assert sa.rays_per_depth[1] == sc.rays_per_depth[1]          # primary rays: lists == list walk
    assert abs(int(sb.n_rays) - int(sc.n_rays)) <= 8 and abs(int(sa.n_rays) - int(sc.n_rays)) <= 8
    assert 2.5 < sa.n_rays / sa.n_paths < 2.6


def _random_scene(rt, seed, offset=None, nesting=False):
    """A seeded random scene through the piecewise API: spheres, rectangles and boxes, some below Translate / RotateY
    wrappers, some bounding a ConstantMedium, with every material and texture kind.  `offset`: the same scene moved there as a
    whole — one more Translate around every object, and the camera (scripts/gpu_frame_fuzz.py: fp32 far from the origin).
    `nesting`: what the reference's object model allows beyond its five demo scenes (hitable.rs:404-588 hold `Arc<dyn Hitable>`):
    3-9 wrappers around every object, 33-47 media, half of them with 3-9 more wrappers AROUND the ConstantMedium itself."""
    rng = np.random.default_rng(seed)
    f = rt._ffi
    s = rt.Scene.new()
    texs = [s.constant_tex(tuple(rng.uniform(0.1, 0.9, 3))), s.checker_tex((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)),
            s.perlin_tex(float(rng.uniform(0.5, 5.0))), s.image_tex("res/earthmap.jpg")]
    any_mat = [f.MAT_EMISSION, f.MAT_DIFFUSE, f.MAT_LAMBERT, f.MAT_METAL, f.MAT_DIELECTRIC, f.MAT_OREN_NAYAR, f.MAT_BURLEY_DIFFUSE,
               f.MAT_ROUGH_PLASTIC, f.MAT_DISNEY_DIFFUSE, f.MAT_DISNEY_SHEEN, f.MAT_DISNEY_CLEARCOAT]

    def material(on_sphere):
        kinds = any_mat + ([f.MAT_DISNEY_METAL] if on_sphere else [])   # DisneyMetal needs the sphere's tangent
        k = kinds[int(rng.integers(len(kinds)))]
        p = {f.MAT_METAL: (float(rng.uniform(0, 1)),), f.MAT_DIELECTRIC: (1.5,), f.MAT_ROUGH_PLASTIC: (float(rng.uniform(0.05, 0.9)), 1.5),
             f.MAT_DISNEY_METAL: (float(rng.uniform(0.1, 0.9)), float(rng.uniform(0, 0.9)), float(rng.uniform(0, 1)))}.get(
                 k, (float(rng.uniform(0.05, 0.95)), float(rng.uniform(0, 1)), 0.25, 0.0))
        p = tuple(p) + (0.0,) * (4 - len(p))
        return s.material(k, tex0=texs[int(rng.integers(4))], tex1=texs[int(rng.integers(4))], color=tuple(rng.uniform(0.2, 0.9, 3)), p=p)

    def wrap(h):
        for _ in range(int(rng.integers(3, 10)) if nesting else int(rng.integers(0, 3))):
            if rng.random() < 0.5:
                h = s.translate(h, tuple(rng.uniform(-3, 3, 3)))
            else:
                h = s.rotate_y(h, float(rng.uniform(-60, 60)))
        return h if offset is None else s.translate(h, tuple(float(x) for x in offset))

    for _ in range(int(rng.integers(3, 25))):
        wrap(s.sphere(tuple(rng.uniform(-8, 8, 3)), float(rng.uniform(0.3, 2.5)), material(True), "s"))
    for _ in range(int(rng.integers(0, 8))):
        axis = int(rng.integers(3))
        mn = rng.uniform(-8, 4, 3)
        mx = mn + rng.uniform(0.5, 6, 3)
        wrap(s.rect(axis, tuple(mn), tuple(mx), material(False)))
    for _ in range(int(rng.integers(0, 5))):
        mn = rng.uniform(-8, 5, 3)
        wrap(s.gbox(tuple(mn), tuple(mn + rng.uniform(0.5, 4, 3)), material(False)))
    for _ in range(int(rng.integers(33, 48)) if nesting else int(rng.integers(0, 4))):
        if rng.random() < 0.5:
            b = s.sphere(tuple(rng.uniform(-6, 6, 3)), float(rng.uniform(1, 3)), material(True), "boundary")
        else:
            mn = rng.uniform(-7, 3, 3)
            b = s.gbox(tuple(mn), tuple(mn + rng.uniform(1, 5, 3)), material(False))
        m = s.constant_medium(wrap(b), float(rng.uniform(0.05, 1.5)), texs[int(rng.integers(3))])
        if nesting and rng.random() < 0.5:
            wrap(m)  # Translate { ptr: ConstantMedium } and RotateY around it: the medium's hit() sees the moved ray
    sky = int(rng.integers(3))
    s.set_sky(sky, "res/newport_loft.jpg" if sky == f.SKY_ENV else None)
    off = np.zeros(3) if offset is None else np.asarray(offset, dtype=np.float64)
    s.set_camera(tuple(float(x) for x in off + np.array([13.0, 2.0, 3.0])), tuple(float(x) for x in off), (0, 1, 0), 40, 1.5)
    s.finish(use_bvh=bool(rng.integers(2)))
    return s


def _one_ulp_of_the_direction_moves_it_as_far(orc, renderer, scene, o, d, key, depth, what, dev, ref):
    """Device and host evaluate the same fp32 expressions, but not with the same last bits everywhere (libm; the order a compiler
    sums a dot product in registers is fixed, sin / cos / acos / atan2 / ln are not).  Where the reference's own formula cancels, one
    such bit is amplified: smith_geo_ggx_aniso (pbr.rs:98-100) is 1 / (v.z + sqrt(.. + v.z^2)), and with v.z < 0 — a DisneyMetal
    sphere below a RotateY wrapper whose face-normal quirk leaves n . i negative — the sum loses three digits (random scene 2546 of
    scripts/gpu_random_scene_sweep.py, ray 20888: device 1.595165, host 1.595254, float64 1.595219; plain numpy float32 gives either,
    depending on which way the direction's last bit is rounded).  No fixed tolerance covers that and still means something for the
    well-conditioned materials, so a colour that differs by more than 5e-5 is accepted only if one of the two implementations' OWN
    colour for this very ray moves at least a quarter as far when one component of the ray direction moves by one ulp (same draws):
    then the difference is what a single rounding upstream does to this lobe."""
    o6 = np.repeat(o[None], 6, axis=0)
    k6 = np.repeat(key[None], 6, axis=0)
    d6 = np.repeat(d[None], 6, axis=0)
    for axis in range(3):
        d6[2 * axis, axis] = np.nextafter(d[axis], np.float32(np.inf))
        d6[2 * axis + 1, axis] = np.nextafter(d[axis], np.float32(-np.inf))
    worst = 0.0
    for res, base in ((orc.debug_bounce(scene.flat_ptr, o6, d6, k6, depth=depth, accel=orc.ACCEL_LIST), ref),
                      (renderer.debug_bounce(o6, d6, k6, depth=depth), dev)):
        q = res[what].astype(np.float64)
        q = q[np.isfinite(q).all(axis=1)]
        if len(q):
            worst = max(worst, float(np.abs(q - base).max()))
    ok = float(np.abs(dev - ref).max()) <= 4.0 * worst
    if ok:
        print(f"ill-conditioned {what}: device {dev.tolist()} host {ref.tolist()}, moves by {worst:.3g} under a one-ulp nudge of the direction (hit record at depth {depth})")
    return ok


@pytest.mark.parametrize("seed", list(range(48)))
def test_random_scenes_bounce_parity(rt, orc, renderer, seed, offset=None, nesting=False):
    """Seeded random scenes over every hitable / material / texture kind: tree search == list walk on the device bit for
    bit, and both == the list-walk oracle (hit, alive, directions exact; t exact except inside media, where ln() differs
    in the last ulp; colours to 2e-5).  `offset` (scripts/gpu_random_scene_sweep.py): scene, camera and rays moved there."""
    scene = _random_scene(rt, 1000 + seed, offset=offset, nesting=nesting)
    renderer.upload(scene)
    rng = np.random.default_rng(seed)
    n = 30000
    o = (rng.uniform(-12, 12, size=(n, 3)) + (0.0 if offset is None else np.asarray(offset, np.float64))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 50, int(seed % 3)] *= np.float32(1e-6)          # a few rays almost parallel to an axis plane (exact-slab path)
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d * (np.float32(1) / ln)[:, None]).astype(np.float32)
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    depth = int(rng.integers(0, 50))
    g = renderer.debug_bounce(o, d, keys, depth=depth)
    b = renderer.debug_bounce(o, d, keys, depth=depth, flags=rt._ffi.FLAG_BRUTE_FORCE)
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=depth, accel=orc.ACCEL_LIST)
    for k in g:
        assert np.array_equal(g[k].view(np.uint8), b[k].view(np.uint8)), k
    assert np.array_equal(g["hit"], c["hit"]) and np.array_equal(g["alive"], c["alive"])
    n_prims = scene.flat.n_spheres + scene.flat.n_rects
    med = g["hit"] >= n_prims
    assert np.array_equal(g["t"][~med].view(np.uint32), c["t"][~med].view(np.uint32))
    assert np.allclose(g["t"][med], c["t"][med], rtol=4e-6)
    assert np.array_equal(g["d"].view(np.uint32), c["d"].view(np.uint32))
    assert np.array_equal(g["o"][~med].view(np.uint32), c["o"][~med].view(np.uint32))  # origins exact, like t, except a medium's scatter point (ln)
    assert np.allclose(g["o"][med], c["o"][med], rtol=1e-5, atol=1e-4)
    for k in ("radiance", "attenuation"):
        a_, c_ = g[k].astype(np.float64), c[k].astype(np.float64)
        fin = np.isfinite(c_)
        assert np.array_equal(np.isfinite(a_), fin), k
        # an ImageTex / environment lookup is nearest-neighbour (texture.rs:183-193): where acos/atan2 of get_uv differ
        # in the last ulp between device and host libm, a (u, v) on a texel edge picks the neighbouring texel.  The
        # near-axis rays above sit exactly on such edges of the 1600x800 environment map (v = 0.5, u = 0.25 / 0.75), so
        # their colours are only compared loosely; for ordinary rays a flip is a 1e-5 event.
        bad = (~np.isclose(a_, c_, rtol=5e-5, atol=2e-6) & fin).any(axis=1)
        assert bad[n // 50:].mean() < 1e-3, (k, int(bad[n // 50:].sum()))  # (ordinary rays: a flip is a 1e-5 event; the near-axis ones all sit on edges)
        for r in np.nonzero(bad)[0]:  # every colour that differs is such a lookup: (u, v) within 2 x 2^-23 of a texel edge, or the test fails
            if med[r] and g["t"][r] != c["t"][r]:
                continue  # a medium scatter whose t differs in the last bits (ln): the phase texture is evaluated an ulp away (hitable.rs:560-570)
            dist = _texel_edge_distance(scene, int(g["hit"][r]), o[r], d[r], g["t"][r])
            if dist is None and _one_ulp_of_the_direction_moves_it_as_far(orc, renderer, scene, o[r], d[r], keys[r], depth, k, a_[r], c_[r]):
                continue  # a formula that cancels (see there): one of the two moves that far itself when its input moves by one ulp
            assert dist is not None and dist <= 2.0, (k, "ray", int(r), "hit", int(g["hit"][r]), "colours differ away from a texel edge", dist, a_[r], c_[r])
    # and a small frame through the whole pipeline (queues, lists, media phase) against the oracle
    cam = scene.camera
    if not np.isfinite(np.array([list(cam.origin), list(cam.horizontal), list(cam.vertical), list(cam.lower_left_corner)])).all():
        return  # a scene moved so far out that lookfrom == lookat in fp32: Camera::new normalises a zero vector, main.rs:39 would panic
    p = rt.make_params(96, 64, 4, max_depth=6)
    img, _, st = renderer.render(scene.camera, p)
    ref, _, so = _oracle(orc, scene, p, accel=orc.ACCEL_LIST)
    if so.n_rays == 0 and st.n_bad_dir == st.n_paths:
        return  # so far out that every primary direction is the zero vector in fp32: main.rs:39 would panic on each; both sides drop them
    _rays_agree(st, so, scene, p)  # exact per depth unless the scene holds a medium
    _compare_frames(orc, scene, p, img, ref, f"random scene {seed}", rt, renderer)  # no medium, no image: no pixel may be off


@pytest.mark.parametrize("seed", list(range(8)))
def test_wrappers_and_media_nest_like_the_trait_objects(rt, orc, renderer, seed):
    """`Translate`, `RotateY`, `ConstantMedium` and `BvhNode` hold an `Arc<dyn Hitable>` each (hitable.rs:404-588): the reference nests
    them to any depth and puts any number of media into a world.  Chains of 3-9 wrappers (more than the four the kernels keep in
    registers: rt_device.h walks those through the list rt_scene_upload builds), 33-47 media (more than the per-lane mask of the
    closest-hit kernel has bits for, rt_kernels.h media_step; their free-path draws use the counter block above 2^30) and wrappers
    around media (RtFlatScene::med_xform), per ray through tree and list walk and as frames, against the oracle."""
    scene = _random_scene(rt, 3000 + seed, nesting=True)
    a = scene.arrays()
    depth = np.zeros(scene.flat.n_xforms, np.int64)
    for x in range(scene.flat.n_xforms):  # a parent precedes its child
        depth[x] = 1 if a["xf_parent"][x] == rt._ffi.NO_XFORM else depth[a["xf_parent"][x]] + 1
    assert depth.max() > 4 and scene.flat.n_media > 32 and (a["med_xform"] != rt._ffi.NO_XFORM).any()
    test_random_scenes_bounce_parity(rt, orc, renderer, 2000 + seed, nesting=True)


def _nested_cloud(rt, n_spheres, shared_chain, own_chains, n_media):
    """An instanced cloud like final_scene's (demo_scene.rs:176-182) taken further: `n_spheres` small spheres below a SHARED chain of
    `shared_chain` wrappers each, equal from sphere to sphere (0: a bare cloud), `own_chains` spheres below 5-7 wrappers of their own,
    plus `n_media` fog boxes, every other one with three wrappers around the medium."""
    rng = np.random.default_rng(123)
    f = rt._ffi
    s = rt.Scene.new()
    white = s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.73, 0.73, 0.73)))
    metal = s.material(f.MAT_METAL, color=(0.8, 0.8, 0.9), p=(0.3,))
    s.sphere((0.0, -1000.0, 0.0), 1000.0, s.material(f.MAT_LAMBERT, tex0=s.checker_tex((0.2, 0.3, 0.1), (0.9, 0.9, 0.9))), "ground")
    s.rect(f.RECT_XZ, (-40.0, 120.0, -40.0), (40.0, 120.0, 40.0), s.material(f.MAT_EMISSION, tex0=s.constant_tex((7, 7, 7))))

    def chain(h, n):
        for _ in range(n):
            h = s.translate(h, tuple(rng.uniform(-4, 4, 3))) if rng.random() < 0.5 else s.rotate_y(h, float(rng.uniform(-40, 40)))
        return h

    first = None
    for k in range(n_spheres - own_chains):  # the cloud (rth_* wrap one handle at a time: the chain is built per sphere from the same draws)
        h = s.sphere(tuple(rng.uniform(0, 100, 3) + (0, 5, 0)), 2.0, white if k % 3 else metal, "cloud")
        first = h if first is None else first
    state = rng.bit_generator.state
    for h in range(first, first + n_spheres - own_chains):
        rng.bit_generator.state = state
        chain(h, shared_chain)
    for _ in range(own_chains):
        chain(s.sphere(tuple(rng.uniform(-60, 60, 3) * (1, 0.3, 1) + (0, 25, 0)), 3.0, metal, "own"), int(rng.integers(5, 8)))
    fog = s.constant_tex((0.6, 0.7, 0.9))
    for m in range(n_media):
        mn = rng.uniform(-70, 50, 3) * (1, 0, 1) + (0, 1, 0)
        h = s.constant_medium(chain(s.gbox(tuple(mn), tuple(mn + rng.uniform(8, 20, 3)), white), 2), 0.05, fog)
        if m % 2:
            chain(h, 3)
    s.set_sky(f.SKY_BLACK)
    s.set_camera((160, 60, -160), (20, 30, 20), (0, 1, 0), 40, 1.5)
    return s.finish()


@pytest.mark.parametrize("n_spheres,shared_chain,own_chains,lds_tree,lds_tables", [(2400, 6, 0, False, False), (2400, 0, 40, False, True),
                                                                                    (150, 0, 30, True, True), (150, 7, 20, True, False)])
def test_nested_scenes_through_every_placement_of_tree_and_tables(rt, orc, renderer, n_spheres, shared_chain, own_chains, lds_tree, lds_tables):
    """The NEST instantiations of the closest-hit kernel (rt_device.h) with the tree in LDS and read through L2, with the wrapper /
    medium tables staged in LDS and read from memory: final_scene-sized instanced clouds below chains of 6-7 wrappers, 36 media, half
    of them below wrappers of their own.  Per ray (the single-kernel hook AND the production kernels through the queue) and as a
    frame against the oracle; tree == list walk on the device."""
    scene = _nested_cloud(rt, n_spheres, shared_chain, own_chains, 36)
    renderer.upload(scene)
    info = renderer.scene_info()
    assert info["nest"] == 1 and info["general_kernels"] == 1
    assert (info["tree_in_lds"], info["general_tables_in_lds"]) == (int(lds_tree), int(lds_tables)), info
    rng = np.random.default_rng(n_spheres + own_chains)
    n = 20000
    o = (rng.uniform(-90, 130, size=(n, 3)) * (1, 0.5, 1) + (0, 2, 0)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
    d = (d * (np.float32(1) / ln)[:, None]).astype(np.float32)
    keys = path_keys(0, np.arange(n), np.zeros(n, dtype=np.uint64))  # what the production kernels derive from the slot
    g = renderer.debug_bounce(o, d, keys, depth=2)
    b = renderer.debug_bounce(o, d, keys, depth=2, flags=rt._ffi.FLAG_BRUTE_FORCE)
    p = renderer.debug_bounce(o, d, keys, depth=2, flags=rt._ffi.FLAG_PRODUCTION_KERNELS)
    c = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=2, accel=orc.ACCEL_LIST)
    for k in g:
        assert np.array_equal(g[k].view(np.uint8), b[k].view(np.uint8)), k
    n_prims = scene.flat.n_spheres + scene.flat.n_rects
    med = c["hit"] >= n_prims
    assert (c["hit"] >= 0).mean() > 0.5 and med.mean() > 0.01
    for dev in (g, p):
        assert np.array_equal(dev["hit"], c["hit"]) and np.array_equal(dev["alive"], c["alive"])
        assert np.array_equal(dev["t"][~med].view(np.uint32), c["t"][~med].view(np.uint32)) and np.allclose(dev["t"][med], c["t"][med], rtol=4e-6)
        live = c["alive"].astype(bool)
        assert np.array_equal(dev["d"][live].view(np.uint32), c["d"][live].view(np.uint32))
        assert np.array_equal(dev["o"][live & ~med].view(np.uint32), c["o"][live & ~med].view(np.uint32))
        assert np.allclose(dev["o"][live & med], c["o"][live & med], rtol=1e-5, atol=1e-3)
        assert np.allclose(dev["attenuation"][live], c["attenuation"][live], rtol=2e-5, atol=1e-6)
    prm = rt.make_params(96, 64, 4, max_depth=8)
    img, _, st = renderer.render(scene.camera, prm)
    ref, _, so = _oracle(orc, scene, prm, accel=orc.ACCEL_LIST)
    _rays_agree(st, so, scene, prm)
    _compare_frames(orc, scene, prm, img, ref, f"nested cloud {n_spheres} / {own_chains}", rt, renderer)


def test_bench_two_ranks_render_and_gather(tmp_path):
    """`python bench.py --gpus 2` end to end on this one GPU: the ranks start themselves, both render their interleaved bands of
    the frame on device 0 (RTOW_DIST_BACKEND=gloo: the gather goes through host memory — the rehearsal of the RCCL run that no
    box with two GPUs has been available for), rank 0 prints the line: config 3's workload label with the requested frame, whole-job
    rays over the slowest rank's time, the gather time, every rank's own trace-step HBM fraction, and the config-2 leg."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RTOW_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--nx", "960", "--ny", "540",
                        "--spp", "16"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["workload"].startswith("config 3:") and "960x540" in out["config"]["workload"]
    assert out["config"]["paths_per_step"] == 960 * 540 * 16 and out["value"] > 0 and out["gather_ms"] > 0 and out["rccl_ranks"] == 0
    assert len(out["roofline"]["per_rank"]) == 2 and all(0 < q["frac"] < 1 for q in out["roofline"]["per_rank"])
    also = out["also"]
    assert also["workload"].startswith("config 2:") and also["scaling"] == "weak" and also["value"] > 0 and len(also["roofline"]["per_rank"]) == 2
    assert out["in_library"]["devices"] == 2  # (rt_multi_create over devices 0 and 1: fails on a one-GPU box, reported, never fatal)
