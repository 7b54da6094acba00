"""Nesting beyond the demo scenes, on the CPU: the flattening of the host mirror and the oracle's evaluation of it.

The reference's wrappers and media hold an `Arc<dyn Hitable>` each (hitable.rs:404-588) and therefore nest to any depth.  The oracle is
pinned here without the GPU by a property the reference's arithmetic has: `Translate` by zero and `RotateY` by 0 degrees (sin = 0,
cos = 1 exactly) change no bit of a ray or of a record, so a scene whose objects sit below a dozen such wrappers — inside the boundary
of a medium and around the medium itself — must give, ray for ray, what the bare scene gives."""
import numpy as np
import pytest

from helpers import rays_on_scene


def _scene(rt, wrappers):
    f = rt._ffi
    s = rt.Scene.new()
    rng = np.random.default_rng(7)

    def wrap(h, n, rotate=True):
        for k in range(n if wrappers else 0):
            h = s.translate(h, (0.0, 0.0, 0.0)) if (k + h) % 2 or not rotate else s.rotate_y(h, 0.0)
        return h

    grey = s.material(f.MAT_DIFFUSE, tex0=s.checker_tex((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)))
    glass = s.material(f.MAT_DIELECTRIC, p=(1.5,))
    metal = s.material(f.MAT_METAL, color=(0.8, 0.6, 0.2), p=(0.3,))
    lamb = s.material(f.MAT_LAMBERT, tex0=s.perlin_tex(4.0))
    wrap(s.sphere((1.0, -1000.0, -1.0), 1000.0, grey, "ground"), 5)
    for k in range(12):
        c = tuple(float(x) for x in rng.uniform(-5, 5, 3) * (1, 0.2, 1) + (0, 0.8, 0))
        # (no RotateY around glass: RotateY::hit calls set_face_normal on a normal that already faces the ray, hitable.rs:505, so a
        # hit from inside comes out as front_face — the one thing even a rotation by zero changes, and only Dielectric reads it)
        wrap(s.sphere(c, float(rng.uniform(0.3, 0.9)), (glass, metal, lamb)[k % 3], "s"), 3 + k, rotate=k % 3 != 0)
    wrap(s.rect(f.RECT_XY, (-3.0, 0.0, -4.0), (3.0, 3.0, -4.0), metal), 9)
    wrap(s.gbox((2.0, 0.0, 1.0), (3.5, 1.5, 2.5), lamb), 6)
    fog = s.constant_tex((0.8, 0.8, 0.9))
    wrap(s.constant_medium(wrap(s.sphere((-2.0, 1.0, 2.0), 1.5, glass, "boundary"), 7), 0.8, fog), 8)   # wrappers inside AND around
    wrap(s.constant_medium(s.gbox((-1.0, 0.0, -3.0), (1.0, 2.0, -1.0), grey), 0.5, fog), 5)             # only around
    s.constant_medium(wrap(s.gbox((3.0, 0.0, -2.0), (5.0, 2.0, 0.0), grey), 12), 0.5, fog)              # only inside
    s.set_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20, 1.5)
    return s.finish()


def test_identity_wrappers_change_nothing(rt, orc):
    bare, deep = _scene(rt, False), _scene(rt, True)
    a = deep.arrays()
    assert bare.flat.n_xforms == 0 and deep.flat.n_xforms > 100 and deep.flat.n_media == 3
    depth = np.zeros(deep.flat.n_xforms, np.int64)
    for x in range(deep.flat.n_xforms):
        depth[x] = 1 if a["xf_parent"][x] == rt._ffi.NO_XFORM else depth[a["xf_parent"][x]] + 1
    assert depth.max() == 15  # 7 inside the first medium's boundary + 8 around the medium
    # the wrapper AROUND each medium: the innermost of the 8 / 5 around the first two, none around the third
    mx = a["med_xform"]
    assert mx[2] == rt._ffi.NO_XFORM and depth[mx[0]] == 8 and depth[mx[1]] == 5
    assert depth[a["sph_xform"][a["sph_medium"] == 0][0]] == 15 and (depth[a["rect_xform"][a["rect_medium"] == 2]] == 12).all()
    o, d, keys = rays_on_scene(40000, 3, radius=9.0)
    for accel in (orc.ACCEL_LIST, orc.ACCEL_BVH):
        for depth_block in (0, 7):
            b = orc.debug_bounce(bare.flat_ptr, o, d, keys, depth=depth_block, accel=accel)
            w = orc.debug_bounce(deep.flat_ptr, o, d, keys, depth=depth_block, accel=accel)
            assert (w["hit"] >= 0).mean() > 0.5 and (w["hit"] >= bare.flat.n_spheres + bare.flat.n_rects).mean() > 0.02
            for k in b:  # == on values: a Translate by zero turns a -0.0 coordinate of a hit point into +0.0
                assert np.array_equal(b[k], w[k], equal_nan=True), (k, accel, depth_block)
    # and whole frames: the same draws, the same paths
    p = rt.make_params(96, 64, 8, max_depth=12)
    opt = orc.options(rng_mode=orc.RNG_COUNTER, accel=orc.ACCEL_LIST)
    img_b, _, st_b = orc.render(bare.flat_ptr, bare.camera, p, opt)
    img_w, _, st_w = orc.render(deep.flat_ptr, deep.camera, p, opt)
    assert list(st_b.rays_per_depth) == list(st_w.rays_per_depth) and np.array_equal(img_b, img_w)


def test_many_media_draw_from_their_own_counters(rt, orc):
    """Media beyond the 32nd use the counter block above 2^30 (oracle.cpp rng_medium_draw, rt_kernels.h medium_counter): 40 concentric
    shells of fog, every ray through the middle meets all of them, and the 40 free-path draws of one ray are 40 different numbers —
    the scatter distances of the media that win follow their own densities."""
    f = rt._ffi
    s = rt.Scene.new()
    fog = s.constant_tex((1.0, 1.0, 1.0))
    glass = s.material(f.MAT_DIELECTRIC, p=(1.5,))
    for k in range(40):
        s.constant_medium(s.sphere((0.0, 0.0, 0.0), 1.0 + 0.1 * k, glass, "shell"), 0.002, fog)
    s.set_camera((0, 0, 20), (0, 0, 0), (0, 1, 0), 20, 1.0)
    scene = s.finish()
    assert scene.flat.n_media == 40
    n = 200000
    rng = np.random.default_rng(5)
    o = np.tile(np.array([[0.0, 0.0, 20.0]], np.float32), (n, 1))
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (n, 1))
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    r = orc.debug_bounce(scene.flat_ptr, o, d, keys, depth=0, accel=orc.ACCEL_LIST)
    won = r["hit"][r["hit"] >= 0] - scene.flat.n_spheres
    share = np.bincount(won, minlength=40) / n
    # what the shares must be: medium k scatters at rate s_k inside [a_k, b_k] of the ray (a shell's chord through the middle), the
    # nearest scatter point wins: P(k wins) = integral of f_k(t) prod_{j != k} P(j has not scattered before t) dt.  The rates come
    # from the flat scene (a medium alone in a BvhNode counts twice, hitable.rs:188: rtow.hpp folds that into the rate).
    rate = -1.0 / scene.arrays()["med_neg_inv_density"].astype(np.float64)
    rad = 1.0 + 0.1 * np.arange(40)
    a, b = 20.0 - rad, 20.0 + rad
    t = np.linspace(a.min(), b.max(), 400001)
    inside = (t[None, :] >= a[:, None]) & (t[None, :] <= b[:, None])
    log_surv = -rate[:, None] * (np.clip(t[None, :], a[:, None], b[:, None]) - a[:, None])  # ln P(no scatter of j before t)
    total = log_surv.sum(axis=0)
    want = np.array([np.trapezoid(np.where(inside[k], rate[k] * np.exp(total), 0.0), t) for k in range(40)])
    assert want.min() > 0.002 and 0.3 < want.sum() < 0.6
    # every one of the 40 wins its share, to the counting noise of 200 000 rays (were the draws of the media from 32 on shared with
    # each other or with other draws of the path, their shares would collapse or double)
    assert np.all(np.abs(share - want) < 5.0 * np.sqrt(want / n) + 1e-4), (share / want)
    assert abs(share[32:].sum() - want[32:].sum()) < 5.0 * np.sqrt(want[32:].sum() / n)
