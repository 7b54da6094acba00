"""Re-derives the analytic known-answer vectors in kat.json with plain Python/numpy (no oracle,
no product code) and checks that the committed JSON agrees; and (re)generates bounce_vectors.json, the per-ray
known answers of tests/golden/np_ref.py — the numpy float32 restatement of one bounce (every Material::scatter,
Sphere::hit, RotateY::hit, Perlin turbulence, ImageTex lookup, the counter RNG and its rejection loop).
Run: python tests/golden/make_golden.py [--write]   (without --write the committed vectors are only re-checked)"""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def xoshiro256pp(s, n):
    M = (1 << 64) - 1
    rotl = lambda x, k: ((x << k) | (x >> (64 - k))) & M
    out = []
    s = list(s)
    for _ in range(n):
        out.append((rotl((s[0] + s[3]) & M, 23) + s[0]) & M)
        t = (s[1] << 17) & M
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t
        s[3] = rotl(s[3], 45)
    return out


def get_uv(n):  # hitable.rs:65-71 in float32
    pi = f32(math.pi)
    theta = f32(np.arccos(f32(-f32(n[1]))))
    phi = f32(np.arctan2(f32(-f32(n[2])), f32(n[0]))) + pi
    return [float(f32(phi / (f32(2) * pi))), float(f32(theta / pi))]


def main():
    kat = json.load(open(os.path.join(HERE, "kat.json")))
    assert [str(v) for v in xoshiro256pp([1, 2, 3, 4], 10)] == kat["xoshiro256pp_state_1_2_3_4"]
    for case in kat["get_uv"]:
        got = get_uv(case["n"])
        assert np.allclose(got, case["uv"], atol=1e-7), (case, got)
    # offset_hit_point((1,0,0),(1,0,0)): bits(1.0) + 256
    bits = np.array([1.0], dtype=np.float32).view(np.uint32)[0] + 256
    assert hex(int(bits)) == kat["offset_hit_point"]["out_bits_x"]
    assert np.array([bits], dtype=np.uint32).view(np.float32)[0] == kat["offset_hit_point"]["out"][0]
    # reflectance(1, 1.5) = r0 = ((1-1.5)/(1+1.5))^2 = 0.04 ; reflectance(0, 1.5) = r0 + (1-r0) = 1
    r0 = f32(f32(1 - 1.5) / f32(1 + 1.5)) ** 2
    assert abs(float(r0) - 0.04) < 1e-7
    print("kat.json agrees with the analytic derivations")




# ---------------------------------------------------------------------------------------------------------------
# bounce_vectors.json
# ---------------------------------------------------------------------------------------------------------------
def bits(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


def bits3(v):
    return [bits(x) for x in v]


def make_rays(n, seed, c, r):
    """n rays towards (mostly) a sphere; float32, directions normalised the glam way."""
    import np_ref as R
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        o = R.v3(*(np.asarray(c) + rng.normal(size=3) * 0.3 + np.array([0.0, 0.4, 0.0]) + 4.0 * np.array(
            [math.cos(0.7 * i + seed), 0.35 * math.sin(1.3 * i), math.sin(0.7 * i + seed)])))
        tgt = R.v3(*(np.asarray(c) + rng.uniform(-1.25, 1.25, size=3) * r))
        d = R.normalize(R.sub(tgt, o))
        out.append((o, d))
    if n > 2:  # an origin inside the sphere (back face) and one on its surface
        out[1] = (R.v3(c[0] + 0.1 * r, c[1] - 0.2 * r, c[2] + 0.05 * r), out[1][1])
    return out


MATS = {  # RtMatType -> parameters (include/rtow_mi355x.h)
    0: {}, 1: {}, 2: {}, 3: {"p0": 0.3}, 4: {"p0": 1.5}, 5: {}, 6: {"p0": 0.5}, 7: {"p0": 0.5}, 8: {"p0": 0.3, "p1": 1.5},
    9: {"p0": 0.5, "p1": 0.5}, 10: {"p0": 0.4, "p1": 0.6, "p2": 0.125}, 11: {"p0": 0.5}, 12: {"p0": 0.7}}
TEX, TEX1, COLOR = (0.7, 0.5, 0.3), (0.2, 0.6, 0.9), (0.8, 0.6, 0.2)
SPHERE_C, SPHERE_R = (0.25, 0.7, -0.5), 0.7


def run_case(case, rays, seed, depth):
    import np_ref as R
    rows = []
    for i, (o, d) in enumerate(rays):
        key = R.path_key(0, i, 0)  # the key the renderer derives for ray i of rt_debug_bounce's production path (seed unused)
        b = R.bounce(case, o, d, key, depth)
        rows.append({"o": bits3(o), "d": bits3(d), "key": [key[0], key[1]], "hit": b["hit"], "t": bits(b["t"]),
                     "alive": int(b["alive"]), "att": bits3(b["att"]), "so": bits3(b["o"]), "sd": bits3(b["d"]),
                     "rad": bits3(b["rad"])})
    return rows


def build_vectors():
    import np_ref as R
    rng = np.random.default_rng(20260404)
    out = {"_about": "Per-ray known answers of tests/golden/np_ref.py (numpy float32, written separately from oracle/ and from "
                     "the HIP kernels); floats are IEEE binary32 bit patterns.  Regenerate: python tests/golden/make_golden.py --write",
           "sphere": {"c": list(SPHERE_C), "r": SPHERE_R}, "tex": list(TEX), "tex1": list(TEX1), "color": list(COLOR), "materials": []}
    for ty, prm in MATS.items():
        mat = dict(type=ty, tex=TEX, tex1=TEX1, color=COLOR, **prm)
        case = {"c": SPHERE_C, "r": SPHERE_R, "mat": mat}
        depth = 1 + ty % 5
        out["materials"].append({"type": ty, "p": [prm.get("p0", 0.0), prm.get("p1", 0.0), prm.get("p2", 0.0), 0.0], "depth": depth,
                                 "rays": run_case(case, make_rays(24, 100 + ty, SPHERE_C, SPHERE_R), 7, depth)})
    # RotateY over a Diffuse sphere: the second set_face_normal of hitable.rs:505
    ang = 30.0
    rad_ = R.f32(ang) * (R.PI / R.f32(180.0))  # f32::to_radians
    sin_t, cos_t = R.f32(np.sin(rad_)), R.f32(np.cos(rad_))
    rot_c = (0.6, 0.7, 0.1)  # the sphere's centre in the wrapper's object space
    world_c = (float(cos_t) * rot_c[0] + float(sin_t) * rot_c[2], rot_c[1], -float(sin_t) * rot_c[0] + float(cos_t) * rot_c[2])
    case = {"c": rot_c, "r": SPHERE_R, "mat": dict(type=1, tex=TEX), "rot": (sin_t, cos_t)}
    out["rotate_y"] = {"angle": ang, "sin": bits(sin_t), "cos": bits(cos_t), "c": list(rot_c), "depth": 2,
                       "rays": run_case(case, make_rays(24, 300, world_c, SPHERE_R), 7, 2)}
    # XZRect below a Translate, Diffuse (hitable.rs:284-322, 404-418): hits from above (front) and from below (back face)
    rect = {"axis": 1, "min": (-1.0, 0.5, -1.25), "max": (1.5, 0.5, 1.0), "offset": (0.3, 0.2, -0.1)}
    case = {"rect": rect, "mat": dict(type=1, tex=TEX)}
    rays = []
    for i in range(32):  # from above (front face) and, every fourth, from below (back face); a fifth aim beside the rectangle
        up = -1.0 if i % 4 == 3 else 1.0
        o = R.v3(0.55 + rng.uniform(-2, 2), 0.7 + up * rng.uniform(1.0, 3.0), -0.225 + rng.uniform(-2, 2))
        tgt = R.v3(0.3 + rng.uniform(-1.3, 1.8), 0.7, -0.1 + rng.uniform(-1.55, 1.3))
        rays.append((o, R.normalize(R.sub(tgt, o))))
    out["rect_translate"] = {"rect": {k: list(v) if isinstance(v, tuple) else v for k, v in rect.items()}, "depth": 1,
                             "rays": run_case(case, rays, 7, 1)}
    # ConstantMedium over a sphere, Isotropic phase function (hitable.rs:523-579, material.rs:99-113)
    density = 1.5
    nid = R.f32(-1.0) / R.f32(density)  # ConstantMedium::new: -1. / density
    case = {"c": SPHERE_C, "r": SPHERE_R, "mat": dict(type=5, tex=TEX), "medium": {"neg_inv_density": nid}}
    out["medium"] = {"density": density, "depth": 3, "rays": run_case(case, make_rays(32, 700, SPHERE_C, SPHERE_R), 7, 3)}
    # Diffuse over a PerlinTex with a fixed table (texture.rs:93-146, 164-168)
    vec = (rng.uniform(-1, 1, size=(256, 3))).astype(np.float32)
    perm = np.stack([rng.permutation(256) for _ in range(3)]).astype(np.uint16)
    scale_ = 4.0
    case = {"c": SPHERE_C, "r": SPHERE_R, "mat": dict(type=1),
            "tex_eval": lambda h: R.perlin_value(vec, perm, R.f32(scale_), h["p"])}
    out["perlin"] = {"scale": scale_, "vec": [bits(x) for x in vec.reshape(-1)], "perm": [int(x) for x in perm.reshape(-1)],
                     "depth": 0, "rays": run_case(case, make_rays(24, 400, SPHERE_C, SPHERE_R), 7, 0)}
    # Emission over an ImageTex (texture.rs:183-193): nearest texel through Sphere::get_uv; rays whose uv lies within
    # 1e-3 of a texel edge are dropped (acos / atan2 differ in the last ulp between libms)
    img = (rng.integers(0, 256, size=(4, 8, 3)).astype(np.float32) / np.float32(255.0)).astype(np.float32)

    def img_eval(h):
        return R.image_value(img, R.get_uv(h["on"]))
    rays = []
    for o, d in make_rays(64, 500, SPHERE_C, SPHERE_R):
        h = R.sphere_hit(R.v3(*SPHERE_C), R.f32(SPHERE_R), o, d, R.f32(1e-3), R.f32(np.finfo(np.float32).max))
        if h is None:
            continue
        u, v = R.get_uv(h["on"])
        fu, fv = float(u) * 8.0, (1.0 - float(v)) * 4.0
        if min(fu % 1.0, 1.0 - fu % 1.0, fv % 1.0, 1.0 - fv % 1.0) > 1e-3 * 8:
            rays.append((o, d))
    case = {"c": SPHERE_C, "r": SPHERE_R, "mat": dict(type=0), "tex_eval": img_eval}
    out["image"] = {"w": 8, "h": 4, "texels": [bits(x) for x in img.reshape(-1)], "depth": 0, "rays": run_case(case, rays[:24], 7, 0)}
    # ImageTex::value on its own, including the edges and a NaN uv (`as u32` gives 0, texture.rs:186-187)
    uvs = [(0.0, 0.0), (1.0, 1.0), (0.999999, 0.000001), (-0.25, 1.5), (0.5, 0.5), (float("nan"), 0.5), (0.3, float("nan")), (0.124999, 0.75),
           (0.125, 0.75), (0.875, 0.25)]
    out["image_lookup"] = [{"uv": [bits(R.f32(u)), bits(R.f32(v))], "rgb": bits3(R.image_value(img, (R.f32(u), R.f32(v))))} for u, v in uvs]
    # the rejection loop of random_in_unit_sphere (math.rs:28-37) on the counter generator: accepted vector and draws used
    rej = []
    for i in range(16):
        key = R.path_key(11, i, 3)
        g = R.Rng(key[0], key[1], 4)
        c0 = g.ctr
        v = R.random_in_unit_sphere(g)
        rej.append({"key": [key[0], key[1]], "depth": 4, "v": bits3(v), "draws": g.ctr - c0})
    out["rejection"] = rej
    return out


if __name__ == "__main__":
    import sys
    sys.path.insert(0, HERE)
    main()
    vec_path = os.path.join(HERE, "bounce_vectors.json")
    with np.errstate(all="ignore"):
        new = build_vectors()
    if "--write" in sys.argv:
        with open(vec_path, "w") as f:
            json.dump(new, f, separators=(",", ":"))
        print("wrote", vec_path, os.path.getsize(vec_path), "bytes")
    else:
        old = json.load(open(vec_path))
        assert json.loads(json.dumps(new)) == old, "bounce_vectors.json is stale: rerun with --write"
        print("bounce_vectors.json agrees with np_ref.py")
