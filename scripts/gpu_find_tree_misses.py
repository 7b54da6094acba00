"""Finds primary rays of config 2 whose closest hit differs between the tree search and the list walk
(diagnostic for the culling slack): bisects by row band with cheap depth-1 renders, then regenerates the band's
rays on the host (numpy restatement of gen_primary) and compares rt_debug_bounce with and without BRUTE_FORCE."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_in_one_weekend_amd as rt

rt.register_default_images()
scene = rt.Scene.build("sphere_scene", 16 / 9)
r = rt.Renderer(0)
r.set_option("primary_lists", 1)  # depth 0 through the tree
r.set_option("grid", 1)           # ... and every other depth too
r.upload(scene)
nx, ny, spp, seed = 1920, 1080, 256, 95
nb = (ny + 7) // 8
bad = []
for b in range(nb):
    pa = rt.make_params(nx, ny, spp, max_depth=1, shard_band=8, shard_count=nb, shard_id=b)
    pb = rt.make_params(nx, ny, spp, max_depth=1, shard_band=8, shard_count=nb, shard_id=b, flags=rt._ffi.FLAG_BRUTE_FORCE)
    _, _, sa = r.render(scene.camera, pa)
    _, _, sb = r.render(scene.camera, pb)
    if int(sa.n_rays) != int(sb.n_rays):
        bad.append(b)
        print("band", b, "tree", sa.n_rays, "brute", sb.n_rays, flush=True)
print("bands with differences:", bad)

M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def mix32(h):
    h = h & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & M32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & M32
    h ^= h >> np.uint64(16)
    return h


def draw(k0, k1, ctr):
    return mix32(((k0 ^ ((np.uint64(ctr) * np.uint64(0x9E3779B9)) & M32)) + k1) & M32)


cam = scene.camera
f32 = np.float32
O = np.array(cam.origin, f32); H = np.array(cam.horizontal, f32); V = np.array(cam.vertical, f32); L = np.array(cam.lower_left_corner, f32)
for b in bad:
    rows = np.arange(8 * b, min(8 * b + 8, ny), dtype=np.uint64)
    jj, ii, ss = np.meshgrid(rows, np.arange(nx, dtype=np.uint64), np.arange(spp, dtype=np.uint64), indexing="ij")
    jj, ii, ss = jj.ravel(), ii.ravel(), ss.ravel()
    pix = (jj * np.uint64(nx) + ii) & M32
    a = fmix32(pix ^ np.uint64(seed & 0xFFFFFFFF))
    k0 = fmix32((a + ss * np.uint64(0x9E3779B9) + np.uint64(seed >> 32)) & M32)
    k1 = fmix32(((a ^ np.uint64(0xA511E9B3)) + ss * np.uint64(0xC2B2AE3D)) & M32)
    r0 = (draw(k0, k1, 0) >> np.uint64(8)).astype(f32) * f32(2.0 ** -24)
    r1 = (draw(k0, k1, 1) >> np.uint64(8)).astype(f32) * f32(2.0 ** -24)
    u = ((ii.astype(f32) + r0) / f32(nx)).astype(f32)
    v = ((jj.astype(f32) + r1) / f32(ny)).astype(f32)
    D = ((L[None, :] + (u[:, None] * H[None, :]).astype(f32)).astype(f32) + (v[:, None] * V[None, :]).astype(f32)).astype(f32) - O[None, :]
    D = D.astype(f32)
    ln = np.sqrt(((D[:, 0] * D[:, 0]).astype(f32) + (D[:, 1] * D[:, 1]).astype(f32)).astype(f32) + (D[:, 2] * D[:, 2]).astype(f32)).astype(f32)
    D = (D * (f32(1.0) / ln)[:, None]).astype(f32)
    Oa = np.broadcast_to(O, D.shape).copy()
    keys = np.stack([k0.astype(np.uint32), k1.astype(np.uint32)], axis=1)
    for c0 in range(0, len(D), 1 << 20):
        sl = slice(c0, c0 + (1 << 20))
        g = r.debug_bounce(Oa[sl], D[sl], keys[sl], depth=0)
        h = r.debug_bounce(Oa[sl], D[sl], keys[sl], depth=0, flags=rt._ffi.FLAG_BRUTE_FORCE)
        diff = np.nonzero(g["hit"] != h["hit"])[0]
        for x in diff:
            gi = c0 + x
            s = int(h["hit"][x])
            arr = scene.arrays()
            print("ray", gi, "pixel", int(ii[gi]), int(jj[gi]), "samp", int(ss[gi]), "o", Oa[gi].tolist(), "d", [float.hex(float(t)) for t in D[gi]],
                  "tree hit", int(g["hit"][x]), "brute hit", s, "t", float(h["t"][x]),
                  "sphere", [float(arr[k][s]) for k in ("sph_cx", "sph_cy", "sph_cz", "sph_r")], flush=True)
