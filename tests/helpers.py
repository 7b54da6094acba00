import numpy as np


def display(img):
    """gamma-2, [0,1]-clamped display values (main.rs:99 before quantisation); NaN -> 0."""
    x = np.nan_to_num(np.asarray(img, dtype=np.float64), nan=0.0, posinf=1.0, neginf=0.0)
    return np.sqrt(np.clip(x, 0.0, 1.0))


def rmse_display(a, b):
    d = display(a) - display(b)
    return float(np.sqrt(np.mean(d * d)))


def rays_on_scene(n, seed, center=(0.0, 0.5, 0.0), radius=12.0):
    """Unit-direction rays starting on a sphere around the scene pointing roughly inwards, plus
    rays starting just above the ground; keys are arbitrary."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v[:, 1] = np.abs(v[:, 1]) * 0.6 + 0.02
    o = (np.asarray(center) + radius * v).astype(np.float32)
    tgt = rng.uniform(-6, 6, size=(n, 3))
    tgt[:, 1] = rng.uniform(-0.2, 1.5, size=n)
    d = tgt - o
    d = d.astype(np.float32)
    # normalise in float32 the glam way: v * (1/len)
    ln = np.sqrt(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
    d = (d * (np.float32(1.0) / ln)[:, None]).astype(np.float32)
    keys = rng.integers(0, 2**32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
    return o, d, keys


def fmix32(h):
    """murmur3 finaliser on uint32 arrays (DESIGN.md "RNG"), written here a third time, in numpy."""
    h = np.asarray(h, dtype=np.uint64) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def mix32(h):
    """the per-draw finaliser of the counter RNG (lowbias32) on uint32 arrays"""
    h = np.asarray(h, dtype=np.uint64) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x7FEB352D) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x846CA68B) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def ctr_draw(k0, k1, ctr):
    """draw(k0, k1, ctr) = mix32((k0 ^ ctr * 0x9E3779B9) + k1) on uint32 arrays (DESIGN.md "RNG")"""
    k0, k1 = np.asarray(k0, dtype=np.uint64), np.asarray(k1, dtype=np.uint64)
    return mix32(((k0 ^ ((np.asarray(ctr, dtype=np.uint64) * 0x9E3779B9) & 0xFFFFFFFF)) + k1) & 0xFFFFFFFF)


def path_keys(seed, pix, samp):
    """(k0, k1) of path (pixel, sample) as uint32 [n, 2] — the key the renderer derives for a slot."""
    pix = np.asarray(pix, dtype=np.uint64)
    samp = np.asarray(samp, dtype=np.uint64)
    s_lo, s_hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    a = fmix32(pix ^ s_lo)
    k0 = fmix32((a + samp * 0x9E3779B9 + s_hi) & 0xFFFFFFFF)
    k1 = fmix32(((a ^ 0xA511E9B3) + samp * 0xC2B2AE3D) & 0xFFFFFFFF)
    return np.stack([k0, k1], axis=1).astype(np.uint32)


def grid_fuzz_scene(rt, seed):
    """The random sphere-only scene `seed` of scripts/gpu_grid_fuzz.py (16 to 3 000 spheres, equal or log-normal radii, a flat layer /
    a cube / a thin slab, with and without huge spheres, at the origin or ~1e5 units from it, unit / tiny / large scale).
    Returns the finished scene, the generator in the state the script continues from, and the parameters drawn."""
    f = rt._ffi
    rng = np.random.default_rng(seed)
    n = int(rng.choice([16, 40, 150, 500, 1200, 3000]))
    scale = float(rng.choice([1.0, 1.0, 1e-3, 250.0]))
    shape = rng.choice(["layer", "cube", "slab"])
    ext = {"layer": (10, 0.3, 10), "cube": (6, 6, 6), "slab": (12, 12, 0.5)}[shape]
    centre = np.array([0.0, 0.0, 0.0]) if rng.random() < 0.7 else rng.normal(size=3) * 300.0
    c = (rng.uniform(-1, 1, (n, 3)) * np.array(ext) + centre) * scale
    rad = (np.full(n, 0.2) if rng.random() < 0.5 else np.exp(rng.normal(np.log(0.2), 0.5, n))) * scale
    if rng.random() < 0.3:
        rad[rng.integers(0, n, n // 10 + 1)] *= -1.0
    s = rt.Scene.new()
    mats = [s.material(f.MAT_DIFFUSE, tex0=s.constant_tex((0.7, 0.6, 0.5))), s.material(f.MAT_METAL, color=(0.8, 0.8, 0.8), p=(0.1,)),
            s.material(f.MAT_DIELECTRIC, p=(1.5,))]
    n_huge = int(rng.choice([0, 1, 1, 2]))
    for k in range(n_huge):  # a ground (and a second big sphere beside the cloud)
        cc = (centre + (np.array([0.0, -ext[1] - 1000.0 - 0.3, 0.0]) if k == 0 else np.array([ext[0] + 1004.0, 0.0, 0.0]))) * scale
        s.sphere(tuple(float(x) for x in cc), 1000.0 * scale, mats[0], "huge")
    for ci, ri in zip(c, rad):
        s.sphere(tuple(float(x) for x in ci), float(ri), mats[int(rng.integers(0, 3))], "s")
    eye = (centre + np.array([0.3, 0.5, 2.2]) * max(ext)) * scale
    s.set_camera(tuple(float(x) for x in eye), tuple(float(x) for x in centre * scale), (0, 1, 0), 50, 1.5)
    s.finish()
    return s, rng, n, scale, shape, ext, centre, c, rad, n_huge
