"""Pins the CPU oracle against the known-answer vectors (tests/golden/kat.json): the public
xoshiro256++ vector and the analytic values derived from the reference formulas.
The reference itself has no tests; these are the strongest pins available (parity unpinned)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))
F = C.POINTER(C.c_float)


def fa(v):
    return np.asarray(v, dtype=np.float32).copy()


def fp(a):
    return a.ctypes.data_as(F)


def test_xoshiro256pp_reference_vector(orc):
    lib = orc.load()
    st = (C.c_uint64 * 4)(1, 2, 3, 4)
    out = (C.c_uint64 * 10)()
    lib.orc_xoshiro_from_state(st, out, 10)
    assert [str(v) for v in out] == KAT["xoshiro256pp_state_1_2_3_4"]


def test_smallrng_seed_candidates_match_survey_scratch(orc):
    lib = orc.load()
    got = np.zeros(4, np.float32)
    lib.orc_smallrng_f32(95, 0, fp(got), 4)
    assert np.allclose(got, KAT["smallrng_seed95_first4_f32"]["pcg32_expansion"], atol=5e-8)
    lib.orc_smallrng_f32(95, 1, fp(got), 4)
    assert np.allclose(got, KAT["smallrng_seed95_first4_f32"]["splitmix64"], atol=5e-8)


def test_f32_samples_are_24_bit_and_below_one(orc):
    lib = orc.load()
    got = np.zeros(4096, np.float32)
    lib.orc_smallrng_f32(1995, 0, fp(got), 4096)
    assert got.min() >= 0.0 and got.max() < 1.0
    assert np.all(got * 16777216.0 == np.round(got * 16777216.0))
    assert 0.45 < got.mean() < 0.55


def test_gen_range_and_shuffle(orc):
    lib = orc.load()
    vals = [lib.orc_gen_range_usize(1995, 0, 3, k) for k in range(64)]
    assert set(vals) <= {0, 1, 2} and len(set(vals)) == 3
    data = np.arange(256, dtype=np.uint16)
    lib.orc_shuffle_u16(7, data.ctypes.data_as(C.POINTER(C.c_uint16)), 256)
    assert sorted(data.tolist()) == list(range(256)) and data.tolist() != list(range(256))


@pytest.mark.parametrize("case", KAT["get_uv"])
def test_sphere_get_uv(orc, case):
    lib = orc.load()
    uv = np.zeros(2, np.float32)
    lib.orc_sphere_get_uv(fp(fa(case["n"])), fp(uv))
    assert np.allclose(uv, case["uv"], atol=1e-7), (case, uv)


def test_camera_new_and_get_ray(orc, rt):
    lib = orc.load()
    k = KAT["camera_test_sphere"]
    a = k["args"]
    cam = rt.RtCamera()
    lib.orc_camera_new(fp(fa(a["lookfrom"])), fp(fa(a["lookat"])), fp(fa(a["vup"])), a["vfov"], a["aspect"], C.byref(cam))
    assert np.allclose(list(cam.horizontal), k["horizontal"], atol=1e-6)
    assert np.allclose(list(cam.vertical), k["vertical"], atol=1e-6)
    assert np.allclose(list(cam.lower_left_corner), k["lower_left_corner"], atol=1e-6)
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib.orc_camera_get_ray(C.byref(cam), 0.5, 0.5, fp(o), fp(d))
    assert np.allclose(d, k["ray_center_d"], atol=1e-6) and np.all(o == 0)


def test_sphere_hit(orc):
    lib = orc.load()
    k = KAT["sphere_hit"]
    out = np.zeros(14, np.float32)
    hit = lib.orc_sphere_hit(fp(fa(k["c"])), k["r"], fp(fa(k["o"])), fp(fa(k["d"])), 1e-3, 3.4028235e38, fp(out))
    assert hit == 1
    assert out[0] == np.float32(k["t"])
    assert np.allclose(out[1:4], k["p"]) and np.allclose(out[4:7], k["n"])
    assert bool(out[7]) == k["front_face"]
    assert np.allclose(out[8:11], k["tang"]) and np.allclose(out[11:13], k["uv"], atol=1e-7)
    # a ray pointing away misses; a root equal to t_max is accepted (hitable.rs:86: `t_max < root`)
    assert lib.orc_sphere_hit(fp(fa(k["c"])), k["r"], fp(fa(k["o"])), fp(fa([0, 0, 1])), 1e-3, 3.4028235e38, fp(out)) == 0
    assert lib.orc_sphere_hit(fp(fa(k["c"])), k["r"], fp(fa(k["o"])), fp(fa(k["d"])), 1e-3, 0.5, fp(out)) == 1
    assert lib.orc_sphere_hit(fp(fa(k["c"])), k["r"], fp(fa(k["o"])), fp(fa(k["d"])), 1e-3, 0.49, fp(out)) == 0
    # from inside: the near root is negative, the far root is taken and the normal flips
    assert lib.orc_sphere_hit(fp(fa(k["c"])), k["r"], fp(fa([0, 0, -1])), fp(fa(k["d"])), 1e-3, 3.4028235e38, fp(out)) == 1
    assert out[0] == np.float32(0.5) and bool(out[7]) is False and np.allclose(out[4:7], [0, 0, 1])


def test_offset_hit_point(orc):
    lib = orc.load()
    k = KAT["offset_hit_point"]
    out = np.zeros(3, np.float32)
    lib.orc_offset_hit_point(fp(fa(k["p"])), fp(fa(k["n"])), fp(out))
    assert out.view(np.uint32)[0] == int(k["out_bits_x"], 16)
    assert out.tolist() == k["out"]
    # near the origin the float offset branch is used (math.rs:149-151)
    lib.orc_offset_hit_point(fp(fa([0.01, -0.01, 0])), fp(fa([0, 1, 0])), fp(out))
    assert out[1] == np.float32(np.float32(-0.01) + np.float32(1.0 / 65536.0)) and out[0] == np.float32(0.01)
    # negative coordinate: the integer offset is subtracted so the point still moves along +n
    lib.orc_offset_hit_point(fp(fa([-1, 0, 0])), fp(fa([1, 0, 0])), fp(out))
    assert out.view(np.uint32)[0] == 0xBF800000 - 256


def test_reflectance_and_sky(orc):
    lib = orc.load()
    for c in KAT["reflectance"]:
        assert abs(lib.orc_reflectance(c["cos"], c["ior"]) - c["r"]) < 1e-7
    out = np.zeros(3, np.float32)
    for c in KAT["sky_color"]:
        lib.orc_sky_gradient(fp(fa(c["d"])), fp(out))
        assert np.allclose(out, c["c"], atol=1e-7)


def test_reflect_refract(orc):
    lib = orc.load()
    out = np.zeros(3, np.float32)
    lib.orc_reflect(fp(fa([1, -1, 0])), fp(fa([0, 1, 0])), fp(out))
    assert out.tolist() == [1, 1, 0]
    # straight-through refraction at normal incidence
    lib.orc_refract(fp(fa([0, -1, 0])), fp(fa([0, 1, 0])), 1.0 / 1.5, fp(out))
    assert np.allclose(out, [0, -1, 0], atol=1e-7)


def test_aabb_hit(orc):
    lib = orc.load()
    mn, mx = fa([-1, -1, -1]), fa([1, 1, 1])
    assert lib.orc_aabb_hit(fp(mn), fp(mx), fp(fa([0, 0, -5])), fp(fa([0, 0, 1])), 1e-3, 3.4e38) == 1
    assert lib.orc_aabb_hit(fp(mn), fp(mx), fp(fa([0, 0, -5])), fp(fa([0, 0, -1])), 1e-3, 3.4e38) == 0
    assert lib.orc_aabb_hit(fp(mn), fp(mx), fp(fa([0, 3, -5])), fp(fa([0, 0, 1])), 1e-3, 3.4e38) == 0
    # t_max <= t_min rejects (math.rs:109): the box starts exactly at t = 4
    assert lib.orc_aabb_hit(fp(mn), fp(mx), fp(fa([0, 0, -5])), fp(fa([0, 0, 1])), 1e-3, 4.0) == 0


def test_counter_rng_is_a_function_of_key_and_counter(orc):
    lib = orc.load()
    k = (C.c_uint32 * 2)()
    keys = set()
    for pix in range(64):
        for s in range(64):
            lib.orc_ctr_path_key(95, pix, s, k)
            keys.add((k[0], k[1]))
    assert len(keys) == 64 * 64
    draws = np.array([lib.orc_ctr_draw(123, 456, c) for c in range(4096)], dtype=np.uint64)
    assert len(set(draws.tolist())) == 4096
    u = (draws >> 8).astype(np.float64) / 16777216.0
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01


def test_counter_draw_statistics(orc):
    """The per-draw hash of the counter RNG (DESIGN.md "RNG": one lowbias32 round over a keyed Weyl sequence) on the key
    set a frame actually uses — 2^20 consecutive pixels, the counters of one depth block: uniform in 1-D (4096 bins) and in
    the 3-D cells the ball sampler's (x, y, z) triples fall into (math.rs:28-37), no serial correlation between a path's
    consecutive draws nor between neighbouring pixels, acceptance rate of the rejection loop = pi/6, and every counter bit
    flips every one of the 24 output bits used (main.rs:89-90 keeps the top 24) with probability 1/2."""
    from helpers import ctr_draw, path_keys
    from scipy import stats
    lib = orc.load()
    n = 1 << 20
    keys = path_keys(95, np.arange(n), np.zeros(n, dtype=np.uint64)).astype(np.uint64)
    k0, k1 = keys[:, 0], keys[:, 1]
    for c in (0, 1, 256, 12345):  # the numpy restatement used below is the oracle's function
        assert int(ctr_draw(k0[7], k1[7], c)) == lib.orc_ctr_draw(int(k0[7]), int(k1[7]), c)
    u = [(ctr_draw(k0, k1, 256 + i) >> 8).astype(np.float64) / 16777216.0 for i in range(4)]
    cnt = np.bincount((u[0] * 4096).astype(np.int64), minlength=4096)
    assert stats.chi2.sf(((cnt - n / 4096) ** 2 / (n / 4096)).sum(), 4095) > 1e-4
    cells = (np.floor(u[0] * 16) * 256 + np.floor(u[1] * 16) * 16 + np.floor(u[2] * 16)).astype(np.int64)
    cnt = np.bincount(cells, minlength=4096)
    assert stats.chi2.sf(((cnt - n / 4096) ** 2 / (n / 4096)).sum(), 4095) > 1e-4
    assert abs(np.corrcoef(u[0], u[1])[0, 1]) < 5e-3 and abs(np.corrcoef(u[0][:-1], u[0][1:])[0, 1]) < 5e-3
    v = [2.0 * x - 1.0 for x in u[:3]]
    assert abs(((v[0] ** 2 + v[1] ** 2 + v[2] ** 2) < 1.0).mean() - np.pi / 6) < 2e-3
    rng = np.random.default_rng(3)
    a0, a1 = rng.integers(0, 2 ** 32, 1 << 16, dtype=np.uint64), rng.integers(0, 2 ** 32, 1 << 16, dtype=np.uint64)
    ctr = rng.integers(0, 1 << 14, 1 << 16, dtype=np.uint64)
    for bit in range(14):
        x = (ctr_draw(a0, a1, ctr) ^ ctr_draw(a0, a1, ctr ^ np.uint64(1 << bit))) >> 8
        flips = np.array([((x >> k) & 1).mean() for k in range(24)])
        assert np.abs(flips - 0.5).max() < 0.012, (bit, flips)  # 6 sigma at 2^16 samples


def test_rect_hit_known_answers(orc):
    """hitable.rs:244-362: t = (k - o)/d on the constant axis, bounds inclusive, uv = (p - min)/(max - min)."""
    lib = orc.load()
    out = np.zeros(10, np.float32)
    mn, mx = fa([3, 1, -2]), fa([5, 3, -2])  # the XYRect of simple_light_scene (demo_scene.rs:100)
    assert lib.orc_rect_hit(2, fp(mn), fp(mx), fp(fa([4, 2, 0])), fp(fa([0, 0, -1])), 1e-3, 3.4e38, fp(out)) == 1
    assert out[0] == 2.0 and out[1:4].tolist() == [4, 2, -2] and out[4:7].tolist() == [0, 0, 1] and out[7] == 1.0
    assert out[8:10].tolist() == [0.5, 0.5]
    # from behind: same plane, flipped normal, front_face false
    assert lib.orc_rect_hit(2, fp(mn), fp(mx), fp(fa([4, 2, -4])), fp(fa([0, 0, 1])), 1e-3, 3.4e38, fp(out)) == 1
    assert out[4:7].tolist() == [0, 0, -1] and out[7] == 0.0
    # the edge is inside (`p.x < min.x || p.x > max.x`), just outside is not; t_max is inclusive (`t > t_max`)
    assert lib.orc_rect_hit(2, fp(mn), fp(mx), fp(fa([5, 3, 0])), fp(fa([0, 0, -1])), 1e-3, 3.4e38, fp(out)) == 1
    assert lib.orc_rect_hit(2, fp(mn), fp(mx), fp(fa([5.001, 3, 0])), fp(fa([0, 0, -1])), 1e-3, 3.4e38, fp(out)) == 0
    assert lib.orc_rect_hit(2, fp(mn), fp(mx), fp(fa([4, 2, 0])), fp(fa([0, 0, -1])), 1e-3, 2.0, fp(out)) == 1
    assert lib.orc_rect_hit(2, fp(mn), fp(mx), fp(fa([4, 2, 0])), fp(fa([0, 0, -1])), 1e-3, 1.999, fp(out)) == 0
    # XZ (axis 1): uv = (x, z); YZ (axis 0): uv = (y, z)
    assert lib.orc_rect_hit(1, fp(fa([0, 5, 0])), fp(fa([4, 5, 2])), fp(fa([1, 9, 0.5])), fp(fa([0, -1, 0])), 1e-3, 3.4e38, fp(out)) == 1
    assert out[0] == 4.0 and out[8:10].tolist() == [0.25, 0.25] and out[4:7].tolist() == [0, 1, 0]
    assert lib.orc_rect_hit(0, fp(fa([7, 0, 0])), fp(fa([7, 4, 2])), fp(fa([0, 1, 1])), fp(fa([1, 0, 0])), 1e-3, 3.4e38, fp(out)) == 1
    assert out[0] == 7.0 and out[8:10].tolist() == [0.25, 0.5] and out[4:7].tolist() == [-1, 0, 0]
