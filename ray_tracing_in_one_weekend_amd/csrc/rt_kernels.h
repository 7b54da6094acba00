// rt_kernels.h — the wavefront (ray-queue) kernels for gfx950.
//
//   gen_primary()    : main.rs:86-94 + camera.rs:40-46; at depth 0 k_intersect and k_shade
//                      regenerate the primary ray from its queue position (no queue traffic);
//                      k_gen_primary materialises the queue only for the list-walk fallback
//   k_intersect      : main.rs:44 world.hit(..) for every queued ray of one depth: closest hit by
//                      traversing the LDS-resident sphere BVH (results == HitableList::hit,
//                      hitable.rs:117-132); persistent lanes refill from the queue as they finish
//   k_intersect_list : the same by the plain list walk over LDS tiles of the sphere list
//                      (RT_FLAG_BRUTE_FORCE, and scenes whose BVH does not fit LDS)
//   k_shade          : main.rs:45-58: emitted + scatter (material.rs, pbr.rs, texture.rs), sky on
//                      a miss, radiance retirement, wave64 ballot compaction of the survivors
//   k_resolve        : main.rs:95-98 sample sum in sample order
//   k_finalize       : main.rs:98-105,127 /spp, gamma 2, *255.99 as u8, vertical flip
//
// Ray queue layout in HBM (SoA, 40 B per ray: two 16 B arrays that k_intersect reads and an 8 B one that only
// k_shade needs), plus an 8 B hit record per ray between the kernels:
//   qa[i] = (o.x, o.y, o.z, slot)   slot = path slot of the slice = s_local * npix + pixel_local
//   qb[i] = (d.x, d.y, d.z, T.x)    T = path throughput          (qa and qb interleaved: one 32 B record, struct Queue)
//   qc[i] = (T.y, T.z)
//   qh[i] = (t, entry index)        written by k_intersect, read by k_shade
// The per-path RNG key (k0, k1) is a pure function of (seed, pixel, sample), i.e. of the slot: it is recomputed
// where it is needed (path_key_of_slot, ~60 instructions) instead of travelling with the ray — past depth 0 k_shade is
// bound by HBM bytes, 8 B less per ray read and per survivor written.
// The queue is cut into `nq` shards of capacity `cap` rays.  Shard q is read and appended to by
// exactly ONE workgroup per kernel (k_shade: workgroup q; k_intersect: workgroup q % gridDim), so
// queue positions come from an LDS counter: the bounce loop issues no global atomics at all.
// (One device-scope atomic per wave on 32 shared counters cost 2x in k_shade: the waves sat in
// s_waitcnt for the returned slot index, profiles/round1.)
#pragma once
#include "rt_device.h"

namespace rt {

// a / b interleaved (RT_QSTRIDE 2: b = a + 1, ray i at a[2 i] and b[2 i]): one 32 B record (o, slot, d, T.x) per ray.
// k_intersect streams through it exactly as through two arrays.  k_shade gathers rays in class order: a 128 B line
// now holds 4 rays instead of 8, so fewer class segments of the sort window come back to it after it has left the
// L2, and one fetched line serves both halves of the ray.
#ifndef RT_QSTRIDE
#define RT_QSTRIDE 2u
#endif
struct Queue {
    float4* a;
    float4* b;
    float2* c;
};
// Radiance slots: RT_RAD_FLOATS floats per path of the slice, written once (one global_store_dwordx3 when a path ends) and
// read once by k_resolve.  3: no padding — 4 B less written and 4 B less read per path than the float4 of round 1.
#ifndef RT_RAD_FLOATS
#define RT_RAD_FLOATS 3
#endif
__device__ __forceinline__ void rad_store(float* __restrict__ rad, size_t slot, float x, float y, float z) {
    if (RT_RAD_FLOATS == 4) *reinterpret_cast<float4*>(rad + 4u * slot) = make_float4(x, y, z, 0.0f);
    else *reinterpret_cast<float3*>(rad + 3u * slot) = make_float3(x, y, z);
}

struct GenParams {
    float cam_origin[3], cam_horizontal[3], cam_vertical[3], cam_llc[3];
    uint32_t nx, ny;
    uint32_t npix;        // pixels of this shard (rows_local * nx)
    uint32_t n_rays;      // npix * s_count
    uint32_t s0;          // first sample index of the slice
    uint32_t shard_band, shard_count, shard_id;
    uint32_t nq, cap;
    uint32_t seed_lo, seed_hi;
    const uint4* lists;   // per-pixel candidate lists of the primary rays (k_primary_lists) or NULL
    const uint32_t* n_overflow; // with `lists`: how many pixels have more candidates than a list holds (k_primary_lists counts them).
                          // None, as on config 2: depth 0 of a sphere-only scene needs no k_intersect at all, and its workgroups
                          // return before they stage anything (1.6 % of the frame's dispatch time went into skipping 530 M rays)
    float inv_npix, inv_nx, inv_band; // reciprocals rounded towards zero by 2^-22 (udiv_inv)
    // Local pixel order.  tiles_per_row = 0: row-major (pl = lj * nx + i).  Else (nx a multiple of 8): the first
    // tile_pixels = 64 * tiles_per_row * (rows / 8) pixels are enumerated in 8 x 8 tiles,
    // pl = (tile_y * tiles_per_row + tile_x) * 64 + (lj & 7) * 8 + (i & 7) — 64 consecutive path slots, i.e. one wave of depth 0,
    // are one 8 x 8 block of pixels instead of a 64 x 1 strip, which crosses fewer silhouettes — and the rows % 8 rows that are
    // left (a shard of 135 rows) follow row-major.
    uint32_t tiles_per_row, tile_pixels;
    float inv_tpr;
};

// x / d for the slot arithmetic of gen_primary / path_key_of_slot (quotients below 2^21): a float product that never exceeds the true quotient
// (inv = (1/d)(1 - 2^-22) absorbs the roundings of the conversion and of the product) and two correction steps.
__device__ __forceinline__ uint32_t udiv_inv(uint32_t x, uint32_t d, float inv, uint32_t& rem) {
    uint32_t qt = (uint32_t)((float)x * inv);
    uint32_t r = x - qt * d;
    if (r >= d) ++qt, r -= d;
    if (r >= d) ++qt, r -= d;
    rem = r;
    return qt;
}
// local pixel index -> (column i, local row lj), and back (k_finalize)
__device__ __forceinline__ void pixel_of_local(const GenParams& gp, uint32_t pl, uint32_t& i, uint32_t& lj) {
    if (gp.tiles_per_row == 0u) {
        lj = udiv_inv(pl, gp.nx, gp.inv_nx, i);
        return;
    }
    if (pl >= gp.tile_pixels) { // the last rows % 8 rows
        lj = udiv_inv(pl, gp.nx, gp.inv_nx, i); // (tile_pixels is a multiple of nx: the same quotient as for a row-major frame)
        return;
    }
    uint32_t tx;
    const uint32_t ty = udiv_inv(pl >> 6, gp.tiles_per_row, gp.inv_tpr, tx);
    i = tx * 8u + (pl & 7u);
    lj = ty * 8u + ((pl >> 3) & 7u);
}
__host__ __device__ inline uint32_t local_of_pixel(uint32_t nx, uint32_t tiles_per_row, uint32_t tile_pixels, uint32_t i, uint32_t lj) {
    if (tiles_per_row == 0u || lj * nx >= tile_pixels) return lj * nx + i;
    return ((lj >> 3) * tiles_per_row + (i >> 3)) * 64u + (lj & 7u) * 8u + (i & 7u);
}
__device__ __forceinline__ uint32_t local_row_to_image_row(uint32_t lj, uint32_t band, uint32_t count, uint32_t id) {
    if (count <= 1) return lj;
    return ((lj / band) * count + id) * band + (lj % band);
}

// RNG key of the path in slot `slot` of the slice: the inverse of slot = s_local * npix + pixel_local, then path_key
// exactly as gen_primary computes it.
__device__ __forceinline__ void path_key_of_slot(const GenParams& gp, uint32_t slot, uint32_t& k0, uint32_t& k1) {
    uint32_t pl, i;
    const uint32_t s_local = udiv_inv(slot, gp.npix, gp.inv_npix, pl);
    uint32_t lj;
    pixel_of_local(gp, pl, i, lj);
    uint32_t j = lj;
    if (gp.shard_count > 1u) {
        uint32_t rb;
        const uint32_t bq = udiv_inv(lj, gp.shard_band, gp.inv_band, rb);
        j = (bq * gp.shard_count + gp.shard_id) * gp.shard_band + rb;
    }
    path_key(((uint64_t)gp.seed_hi << 32) | gp.seed_lo, j * gp.nx + i, gp.s0 + s_local, k0, k1);
}

// Primary ray of path `idx` of the slice (main.rs:86-94 + camera.rs:40-46).
// idx = s_local * npix + pixel_local, so consecutive idx are consecutive pixels of a row.
__device__ __forceinline__ void gen_primary(const GenParams& gp, uint32_t idx, V3& o, V3& d, uint32_t& k0, uint32_t& k1,
                                            uint32_t& pl) {
    uint32_t i;
    const uint32_t s_local = udiv_inv(idx, gp.npix, gp.inv_npix, pl); // pl = local pixel
    uint32_t lj;
    pixel_of_local(gp, pl, i, lj);
    const uint32_t j = local_row_to_image_row(lj, gp.shard_band, gp.shard_count, gp.shard_id);
    const uint32_t samp = gp.s0 + s_local;
    Rng rng;
    path_key(((uint64_t)gp.seed_hi << 32) | gp.seed_lo, j * gp.nx + i, samp, rng.k0, rng.k1);
    rng.w = 0u; // counter 0
    // main.rs:89-90
    const float u = ((float)i + rng.next()) / (float)gp.nx;
    const float v = ((float)j + rng.next()) / (float)gp.ny;
    // camera.rs:40-46
    const V3 origin = v3(gp.cam_origin[0], gp.cam_origin[1], gp.cam_origin[2]);
    const V3 H = v3(gp.cam_horizontal[0], gp.cam_horizontal[1], gp.cam_horizontal[2]);
    const V3 Vv = v3(gp.cam_vertical[0], gp.cam_vertical[1], gp.cam_vertical[2]);
    const V3 llc = v3(gp.cam_llc[0], gp.cam_llc[1], gp.cam_llc[2]);
    o = origin;
    d = normalize(llc + u * H + v * Vv - origin);
    k0 = rng.k0, k1 = rng.k1;
}
__device__ __forceinline__ void gen_primary(const GenParams& gp, uint32_t idx, V3& o, V3& d, uint32_t& k0, uint32_t& k1) {
    uint32_t pl;
    gen_primary(gp, idx, o, d, k0, k1, pl);
}

// Depth-0 queue geometry: chunks of 256 consecutive paths are dealt round-robin to the queue
// shards (every shard holds a lattice of image locations, so shards stay balanced as paths die).
//   path idx -> shard (idx/256) % nq, position ((idx/256)/nq)*256 + idx%256     and back:
__device__ __forceinline__ uint32_t primary_idx_of(uint32_t nq, uint32_t shard, uint32_t pos) {
    return ((pos >> 8) * nq + shard) * 256u + (pos & 255u);
}
// Number of primary rays of each shard (closed form of the mapping above).
// Also parks the slice's GenParams in HBM for the depth-0 kernels (they read them through a
// pointer: 24 dwords of kernel arguments would push k_intersect past 80 SGPRs and cost a wave).
__global__ __launch_bounds__(256) void k_init_counts(GenParams gp, uint32_t* __restrict__ counts,
                                                     GenParams* __restrict__ gp_dev) {
    const uint32_t sq = blockIdx.x * 256u + threadIdx.x;
    if (sq == 0) *gp_dev = gp;
    if (sq >= gp.nq) return;
    const uint32_t nchunks = (gp.n_rays + 255u) / 256u;
    const uint32_t nc = sq < nchunks ? (nchunks - sq + gp.nq - 1u) / gp.nq : 0u;
    uint32_t cnt = nc * 256u;
    if (nc && ((nchunks - 1u) % gp.nq) == sq) cnt -= nchunks * 256u - gp.n_rays;
    counts[sq] = cnt;
}

// Materialises the primary rays in the queue.  Only the list-walk fallback uses it: on the BVH
// path k_intersect and k_shade regenerate the ray of depth 0 from its queue position instead
// (80 instructions twice per path against 48 B written + 80 B read back from HBM).
__global__ __launch_bounds__(256) void k_gen_primary(GenParams gp, Queue q) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= gp.n_rays) return;
    V3 o, d;
    uint32_t k0, k1;
    gen_primary(gp, idx, o, d, k0, k1);
    const uint32_t chunk = idx >> 8;
    const uint32_t sq = chunk % gp.nq;
    const size_t pos = (size_t)sq * gp.cap + (size_t)(chunk / gp.nq) * 256u + (idx & 255u);
    q.a[RT_QSTRIDE * pos] = make_float4(o.x, o.y, o.z, __uint_as_float(idx));
    q.b[RT_QSTRIDE * pos] = make_float4(d.x, d.y, d.z, 1.0f);
    q.c[pos] = make_float2(1.0f, 1.0f);
}

// Candidate lists of the primary rays.  All samples of a pixel leave the camera through the pixel's footprint
// (main.rs:89-90: u in [i/nx, (i+1)/nx], v likewise), i.e. inside a narrow cone around the ray through its
// centre.  One lane per pixel tests that cone against the bounding sphere of every world entry and keeps the
// entries it may touch: with 256 samples per pixel, depth 0 of every slice then tests ~2 listed primitives per
// ray with the exact leaf code instead of walking the tree (k_intersect, GEN).  The test is conservative by
// construction — cone half-angle from the four footprint corners widened by 2 % + 1e-4 rad, radii by 0.1 % —
// so a list contains every entry any sample can hit, and the closest-hit rule does not depend on the order or
// on extra candidates: same bits as the traversal.  More than 7 candidates: the pixel's rays use the tree.
#define RT_LIST_MAX 7u
#define RT_LIST_OVERFLOW 0xFFFFu
#define RT_LIST_WAVE_CAP 512u // survivors of the wave-level cull kept per wave (more: the lanes scan all entries)
// cone (axis, half-angle alpha given as sin/cos) against a bounding sphere seen from `origin`
__device__ __forceinline__ bool cone_touches_sphere(V3 origin, V3 axis, float sa, float ca, float4 bs) {
    const V3 c = v3(bs.x, bs.y, bs.z) - origin;
    // (the last term: `c` is a difference of fp32 coordinates, good to half an ulp of the larger one per component — nothing next to
    // the 0.1 % for a scene around the origin, the whole margin for a small sphere 1e5 units away from it)
    const float big = fmaxf(fmaxf(fmaxf(fabsf(bs.x), fabsf(bs.y)), fabsf(bs.z)), fmaxf(fmaxf(fabsf(origin.x), fabsf(origin.y)), fabsf(origin.z)));
    const float r = bs.w * 1.001f + 1e-6f + big * (1.0f / 2097152.0f);
    const float dist2 = length_squared(c);
    if (!(dist2 > r * r * 1.0001f)) return true; // the eye is inside or on the bounding sphere (or NaN)
    const float dist = sqrtf(dist2);
    const float sb = fminf(r / dist, 1.0f), cb = sqrtf(fmaxf(1.0f - sb * sb, 0.0f));
    // theta = angle(axis, c) <= alpha + beta, beta = angular radius.  Both below 90 degrees (the usual case):
    // compared through sines, |c x axis| = dist*sin(theta), which stays well-conditioned for the sub-milliradian
    // angles of a pixel (cosines would need a slack of several pixels).  Otherwise through cosines, which are
    // well-conditioned there.
    const float c_theta = dot(c, axis);
    const float c_sum = ca * cb - sa * sb, s_sum = sa * cb + ca * sb;
    if (c_theta > 0.0f && c_sum > 0.0f) return length(cross(c, axis)) <= s_sum * dist * 1.0001f + 1e-6f * dist;
    return c_theta >= c_sum * dist - 1e-5f * dist;
}
__global__ __launch_bounds__(256) void k_primary_lists(DevScene sc, GenParams gp, uint4* __restrict__ lists, uint32_t* __restrict__ n_overflow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_bs = reinterpret_cast<float4*>(smem);
    unsigned short* s_keep = reinterpret_cast<unsigned short*>(s_bs + sc.n_entries) + (threadIdx.x >> 6) * RT_LIST_WAVE_CAP;
    for (uint32_t e = threadIdx.x; e < sc.n_entries; e += 256u) s_bs[e] = sc.ent_bs[e];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pl = blockIdx.x * 256u + threadIdx.x;
    const bool active = pl < gp.npix;
    const uint32_t plc = active ? pl : gp.npix - 1u;
    uint32_t i, lj;
    pixel_of_local(gp, plc, i, lj);
    const uint32_t j = local_row_to_image_row(lj, gp.shard_band, gp.shard_count, gp.shard_id);
    const V3 origin = v3(gp.cam_origin[0], gp.cam_origin[1], gp.cam_origin[2]);
    const V3 H = v3(gp.cam_horizontal[0], gp.cam_horizontal[1], gp.cam_horizontal[2]);
    const V3 Vv = v3(gp.cam_vertical[0], gp.cam_vertical[1], gp.cam_vertical[2]);
    const V3 llc = v3(gp.cam_llc[0], gp.cam_llc[1], gp.cam_llc[2]);
    const float u0 = (float)i / (float)gp.nx, u1 = ((float)i + 1.0f) / (float)gp.nx;
    const float v0 = (float)j / (float)gp.ny, v1 = ((float)j + 1.0f) / (float)gp.ny;
    const V3 axis = normalize(llc + (0.5f * (u0 + u1)) * H + (0.5f * (v0 + v1)) * Vv - origin);
    // half-angle of the footprint cone through sines (|axis x corner|): a pixel spans ~1e-4 rad, far below the
    // resolution of a cosine near 1
    float smax = 0.0f;
    smax = fmaxf(smax, length(cross(axis, normalize(llc + u0 * H + v0 * Vv - origin))));
    smax = fmaxf(smax, length(cross(axis, normalize(llc + u1 * H + v0 * Vv - origin))));
    smax = fmaxf(smax, length(cross(axis, normalize(llc + u0 * H + v1 * Vv - origin))));
    smax = fmaxf(smax, length(cross(axis, normalize(llc + u1 * H + v1 * Vv - origin))));
    // The rays are what fp32 makes of camera.rs:43-46, and so are the four corners above: every component of
    // llc + u H + v V - origin goes through four roundings of operands up to S = |llc| + |H| + |V| + |origin| (per component), so a
    // sample's direction and each corner lie within e = 4 * 2^-23 * S of where exact arithmetic puts them, an angle of sqrt(3) e / |dir|
    // each.  Nothing for a camera near the origin (1e-5 rad); for one 1e5 units away from it with the reference's unit focal length the
    // directions are quantised to whole pixels (scripts/gpu_grid_fuzz.py, seed 494: a listed pixel lost a grazing sphere) — the
    // cone then opens until the list overflows and the pixel's rays use the tree.
    const V3 dc = llc + (0.5f * (u0 + u1)) * H + (0.5f * (v0 + v1)) * Vv - origin;
    const float S = fmaxf(fmaxf(fabsf(llc.x) + fabsf(H.x) + fabsf(Vv.x) + fabsf(origin.x), fabsf(llc.y) + fabsf(H.y) + fabsf(Vv.y) + fabsf(origin.y)),
                          fabsf(llc.z) + fabsf(H.z) + fabsf(Vv.z) + fabsf(origin.z));
    const float rounding = 2.0f * 1.7321f * (S * (4.0f / 8388608.0f)) / (length(dc) * (1.0f - smax));
    const float alpha = asinf(fminf(smax, 1.0f)) * 1.02f + 1e-4f + rounding;
    const bool unusable = !(alpha < 1.5f); // (or NaN) no cone to speak of: the pixel's rays use the tree
    const float ca = cosf(alpha), sa = sinf(alpha);

    // ---- wave-level cull: one cone around the 64 pixels of the wave (any 64: a row strip, a row wrap, rows of
    // different shard bands), every lane tests 1/64 of the entries against it ------------------------------
    V3 wsum = axis; // inactive lanes duplicate the last pixel: harmless for a bounding cone
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        wsum.x += __shfl_xor(wsum.x, off);
        wsum.y += __shfl_xor(wsum.y, off);
        wsum.z += __shfl_xor(wsum.z, off);
    }
    const V3 waxis = normalize(wsum);
    // angle between the wave axis and this pixel's axis (through the sine when below 90 degrees), plus alpha
    const float wc = dot(waxis, axis);
    float wang = wc > 0.0f ? asinf(fminf(length(cross(waxis, axis)), 1.0f)) : 3.2f;
    wang = wang * 1.001f + alpha;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wang = fmaxf(wang, __shfl_xor(wang, off));
    const bool wide = !(wang < 1.5f); // a cone wider than ~86 degrees (or NaN): no cull
    const float wca = cosf(wang), wsa = sinf(wang);
    uint32_t n_keep = 0; // wave-uniform
    bool scan_all = wide;
    if (!wide) {
        for (uint32_t e0 = 0; e0 < sc.n_entries; e0 += 64u) {
            const uint32_t e = e0 + lane;
            const bool keep = e < sc.n_entries && cone_touches_sphere(origin, waxis, wsa, wca, s_bs[e]);
            const unsigned long long m = __ballot(keep);
            const uint32_t cnt = (uint32_t)__popcll(m);
            if (n_keep + cnt > RT_LIST_WAVE_CAP) {
                scan_all = true;
                break;
            }
            if (keep) s_keep[n_keep + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (unsigned short)e;
            n_keep += cnt;
        }
    }
    // ---- per pixel: the survivors (or everything) against the pixel's own cone, in entry order ---------------
    uint32_t n = 0;
    uint32_t ids[RT_LIST_MAX];
#pragma unroll
    for (uint32_t t = 0; t < RT_LIST_MAX; ++t) ids[t] = 0u;
    const uint32_t n_scan = scan_all ? sc.n_entries : n_keep;
    for (uint32_t t0 = 0; t0 < n_scan; ++t0) {
        const uint32_t e = scan_all ? t0 : (uint32_t)s_keep[t0];
        if (cone_touches_sphere(origin, axis, sa, ca, s_bs[e])) {
#pragma unroll
            for (uint32_t t = 0; t < RT_LIST_MAX; ++t)
                if (n == t) ids[t] = sc.ent_leaf[e];
            ++n;
        }
    }
    if (!active) return;
    const uint32_t head = (n > RT_LIST_MAX || unusable) ? RT_LIST_OVERFLOW : n;
    if (head == RT_LIST_OVERFLOW) atomicAdd(n_overflow, 1u);
    lists[pl] = make_uint4(head | (ids[0] << 16), ids[1] | (ids[2] << 16), ids[3] | (ids[4] << 16), ids[5] | (ids[6] << 16));
}

// LDS tile of the sphere list: (cx, cy, cz, r) as float4, read by every lane at the same
// address (broadcast, conflict-free).
#define RT_SPHERE_TILE 2048u

// Closest hit over a tile of the list, HitableList::hit order and acceptance rule
// (hitable.rs:117-132: t_max shrinks to the closest so far; a root equal to t_max is accepted).
__device__ __forceinline__ void closest_hit_tile(const float4* s_geo, uint32_t n, uint32_t base, V3 o, V3 d, float a,
                                                 float& tbest, int& hit) {
    for (uint32_t s = 0; s < n; ++s) {
        float th;
        if (sphere_root(s_geo[s], o, d, a, 1e-3f, tbest, th)) {
            tbest = th;
            hit = (int)(base + s);
        }
    }
}

// Per-ray data a ConstantMedium needs inside hit(): the path's RNG key and the depth's counter block.
struct MediumCtx {
    uint32_t k0, k1, base;
};
// Counter of medium m's free-path draw at the depth whose block starts at `base` = (depth + 1) * 256 (DESIGN.md "RNG"): slots 224-255
// of the block for the first 32 media of a scene (every scene of rounds 1-5: the same draws as before), and for the media beyond them
// a block of 65 536 counters per depth above 2^30, where no other draw of a path lies (base < 2^21: max_depth <= 4096).
__device__ __forceinline__ uint32_t medium_counter(uint32_t base, uint32_t m) {
    return m < 32u ? base + 224u + m : 0x40000000u + (base << 8) + m;
}
// Candidate root of one geometric primitive through its wrapper chain (general scenes), geometry from `geo`
// (spheres, then 2 float4 per rectangle; LDS in k_intersect, HBM in the list walk).
template <bool NEST, class Tables>
__device__ __forceinline__ bool prim_root(const Tables& sc, const float4* geo, uint32_t s, V3 o, V3 d, float t_min,
                                          float t_max, float& th) {
    const uint32_t xf = sc.prim_xform[s];
    if (xf != RT_NO_XFORM_DEV) ray_to_object<NEST>(sc, xf, o, d);
    if (s < sc.n_spheres) return sphere_root(geo[s], o, d, length_squared(d), t_min, t_max, th);
    const uint32_t gi = sc.n_spheres + 2u * (s - sc.n_spheres);
    return rect_root(geo[gi], geo[gi + 1u], o, d, t_min, t_max, th);
}
// The object-space ray of the wrapper chain used last: the leaves of one node usually share their chain (the
// instanced spheres of a cloud), so the leaf loop of a node step moves the ray once instead of once per leaf.
struct ChainCache {
    uint32_t xf; // RT_NO_XFORM_DEV = empty
    V3 o, d;
};
template <bool NEST, class Tables>
__device__ __forceinline__ bool prim_root_cached(const Tables& sc, const float4* geo, uint32_t s, V3 o, V3 d, float t_min,
                                                 float t_max, ChainCache& cc, float& th) {
    const uint32_t xf = sc.prim_xform[s];
    if (xf != RT_NO_XFORM_DEV) {
        if (xf != cc.xf) {
            cc.o = o, cc.d = d;
            ray_to_object<NEST>(sc, xf, cc.o, cc.d);
            cc.xf = xf;
        }
        o = cc.o, d = cc.d;
    }
    if (s < sc.n_spheres) return sphere_root(geo[s], o, d, length_squared(d), t_min, t_max, th);
    const uint32_t gi = sc.n_spheres + 2u * (s - sc.n_spheres);
    return rect_root(geo[gi], geo[gi + 1u], o, d, t_min, t_max, th);
}
// med_range[m] = (first entry in med_prims, count | kind << 24): see "ConstantMedium::hit searches its boundary twice" below
#define RT_MED_KIND_RECTS 1u
#define RT_MED_KIND_SPHERE 2u
#define RT_MED_COUNT_MASK 0x00FFFFFFu
// Root of one boundary primitive for a ray that is already in the object space of the medium's common chain
// (RAW) or still in world space (the primitive applies its own chain).
template <bool RAW, bool NEST, class Tables>
__device__ __forceinline__ bool boundary_prim_root(const Tables& sc, const float4* geo, uint32_t s, V3 o, V3 d,
                                                   float t_min, float t_max, float& th) {
    if (!RAW) return prim_root<NEST>(sc, geo, s, o, d, t_min, t_max, th);
    if (s < sc.n_spheres) return sphere_root(geo[s], o, d, length_squared(d), t_min, t_max, th);
    const uint32_t gi = sc.n_spheres + 2u * (s - sc.n_spheres);
    return rect_root(geo[gi], geo[gi + 1u], o, d, t_min, t_max, th);
}
// boundary.hit(r, t_min, t_max): closest accepted root over the medium's boundary primitives
template <bool RAW, bool NEST, class Tables>
__device__ __forceinline__ bool boundary_root(const Tables& sc, const float4* geo, uint32_t m, V3 o, V3 d, float t_min,
                                              float t_max, float& t_out) {
    const uint2 rg = sc.med_range[m];
    bool any = false;
    for (uint32_t k = 0; k < (rg.y & RT_MED_COUNT_MASK); ++k) {
        float th;
        if (boundary_prim_root<RAW, NEST>(sc, geo, sc.med_prims[rg.x + k], o, d, t_min, t_max, th)) {
            t_max = th;
            any = true;
        }
    }
    t_out = t_max;
    return any;
}
#define RT_MED_XF_MIXED 0xFFFFFFFEu // the boundary's primitives do not share one wrapper chain
// ConstantMedium::hit searches its boundary twice (hitable.rs:541-552: `boundary.hit(r, -inf, inf)`, then `boundary.hit(r, t1 + 0.0001,
// inf)`), and a search of a GBox is six rectangle tests.  Where a rectangle's plane distance and its bounds test, or a sphere's two
// roots, do not depend on the search window, both searches can be answered from ONE evaluation of the primitives: the candidates are
// computed once and the two windows — with HitableList::hit's shrinking t_max (hitable.rs:117-132), comparison for comparison — are
// applied to the stored values.  Same operations on the same operands as rect_root / sphere_root, so the same bits; half the
// arithmetic of a medium test (cornell_box spends a third of its closest-hit time in its two smoke boxes).  rt_scene_upload marks
// the boundaries this applies to in the top byte of med_range[m].y: up to six rectangles, or one sphere, below one wrapper chain.
__device__ __forceinline__ bool in_window(float t, float t_min, float t_max) { return !(t < t_min || t_max < t); }
// both searches over <= 6 rectangles in object space; false when either finds nothing or the medium is entered behind t_cull
template <class Tables>
__device__ __forceinline__ bool boundary_both_rects(const Tables& sc, const float4* geo, uint2 rg, V3 o, V3 d, float t_cull, float& t1, float& t2) {
    const uint32_t n = rg.y & RT_MED_COUNT_MASK;
    float tk[6]; // plane distance of rectangle k where it passes XYRect::hit's tests that do not involve the window, else NaN
#pragma unroll
    for (uint32_t k = 0; k < 6u; ++k) {
        tk[k] = __int_as_float(0x7FC00000);
        if (k < n) {
            const uint32_t s = sc.med_prims[rg.x + k];
            const uint32_t gi = sc.n_spheres + 2u * (s - sc.n_spheres);
            const float4 g0 = geo[gi], g1 = geo[gi + 1u];
            const uint32_t axis = __float_as_uint(g1.y);
            const float oa = axis == 0u ? o.x : (axis == 1u ? o.y : o.z);
            const float da = axis == 0u ? d.x : (axis == 1u ? d.y : d.z);
            const float t = (g0.x - oa) / da;       // rect_root, hitable.rs:252
            const V3 p = o + d * t;                 // Ray::at
            const float pu = axis == 0u ? p.y : p.x;
            const float pv = axis == 2u ? p.y : p.z;
            if (t == t && !(pu < g0.y || pu > g0.z || pv < g0.w || pv > g1.x)) tk[k] = t;
        }
    }
    // boundary.hit(r, -inf, inf): rect_root's window test `t < t_min || t > t_max` with t_max shrinking to every accepted root
    float t_max = INFINITY;
    bool any = false;
#pragma unroll
    for (uint32_t k = 0; k < 6u; ++k)
        if (tk[k] == tk[k] && !(tk[k] < -INFINITY || tk[k] > t_max)) t_max = tk[k], any = true;
    if (!any) return false;
    t1 = t_max;
    if (t1 > t_cull) return false;
    const float t_min2 = t1 + 0.0001f; // boundary.hit(r, rec1.t + 0.0001, inf)
    t_max = INFINITY, any = false;
#pragma unroll
    for (uint32_t k = 0; k < 6u; ++k)
        if (tk[k] == tk[k] && !(tk[k] < t_min2 || tk[k] > t_max)) t_max = tk[k], any = true;
    t2 = t_max;
    return any;
}
// both searches of a boundary that is one sphere: Sphere::hit's two roots (hitable.rs:75-91) once, its window logic twice
template <class Tables>
__device__ __forceinline__ bool boundary_both_sphere(const Tables& sc, const float4* geo, uint2 rg, V3 o, V3 d, float t_cull, float& t1, float& t2) {
    const float4 g = geo[sc.med_prims[rg.x]];
    const float a = length_squared(d);
    const V3 oc = o - v3(g.x, g.y, g.z);
    const float half_b = dot(oc, d);
    const float c = length_squared(oc) - g.w * g.w;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return false;
    const float sqrtd = sqrtf(discriminant);
    const float r0 = (-half_b - sqrtd) / a, r1 = (-half_b + sqrtd) / a;
    // (sphere_root's early-out for a sphere behind the origin only skips arithmetic whose roots fail the window below: rt_device.h)
    if (in_window(r0, -INFINITY, INFINITY)) t1 = r0;
    else if (in_window(r1, -INFINITY, INFINITY)) t1 = r1;
    else return false;
    if (t1 > t_cull) return false;
    const float t_min2 = t1 + 0.0001f;
    if (in_window(r0, t_min2, INFINITY)) t2 = r0;
    else if (in_window(r1, t_min2, INFINITY)) t2 = r1;
    else return false;
    return true;
}
// ConstantMedium::hit, hitable.rs:536-579, up to the accepted t.  The random draw is counter slot
// 224 + m of the depth block (DESIGN.md "RNG"): independent of the order in which media are visited.
// t_max clamps like the reference's (hitable.rs:553-555); callers pass FLT_MAX and apply the
// order-independent winner rule, which differs from the clamp only by rounding at exact ties.
// `t_cull`: a scatter point cannot lie before the entry t1, so a medium entered beyond the best hit so far
// cannot win and the second boundary search is skipped (pure culling, no effect on the result).
// When all boundary primitives sit below the same wrapper chain (a GBox under RotateY/Translate) the ray is
// moved to object space once instead of once per primitive and search — the same bits either way.
template <bool NEST, class Tables>
__device__ __forceinline__ bool medium_root(const Tables& sc, const float4* geo, uint32_t m, V3 o, V3 d, float t_min,
                                            float t_max, float t_cull, const MediumCtx& mc, float& t_hit) {
    float t1, t2;
    // r.d.length() of the ray ConstantMedium::hit receives (hitable.rs:560): the world ray, or (NEST) what the wrappers AROUND the
    // medium make of it (a rotation changes the last bits of a length)
    float ray_len = length(d);
    const uint32_t fx = NEST ? sc.med_xform[m].y : RT_NO_XFORM_DEV;
    if (NEST && fx != RT_NO_XFORM_DEV) {
        V3 fo = o, fd = d;
        ray_to_object<NEST>(sc, fx, fo, fd);
        ray_len = length(fd);
    }
    const uint32_t cx = sc.med_xform[m].x;
    if (cx != RT_MED_XF_MIXED) {
        if (cx != RT_NO_XFORM_DEV) ray_to_object<NEST>(sc, cx, o, d);
        const uint2 rg = sc.med_range[m];
        const uint32_t kind = rg.y >> 24;
        if (kind == RT_MED_KIND_RECTS) {
            if (!boundary_both_rects(sc, geo, rg, o, d, t_cull, t1, t2)) return false;
        } else if (kind == RT_MED_KIND_SPHERE) {
            if (!boundary_both_sphere(sc, geo, rg, o, d, t_cull, t1, t2)) return false;
        } else {
            if (!boundary_root<true, NEST>(sc, geo, m, o, d, -INFINITY, INFINITY, t1)) return false;
            if (t1 > t_cull) return false;
            if (!boundary_root<true, NEST>(sc, geo, m, o, d, t1 + 0.0001f, INFINITY, t2)) return false;
        }
    } else {
        if (!boundary_root<false, NEST>(sc, geo, m, o, d, -INFINITY, INFINITY, t1)) return false;
        if (t1 > t_cull) return false;
        if (!boundary_root<false, NEST>(sc, geo, m, o, d, t1 + 0.0001f, INFINITY, t2)) return false;
    }
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return false;
    if (t1 < 0.0f) t1 = 0.0f;
    const float dist_inside_boundary = (t2 - t1) * ray_len;
    const uint32_t r = mix32((mc.k0 ^ ((NEST ? medium_counter(mc.base, m) : mc.base + 224u + m) * 0x9E3779B9u)) + mc.k1);
    const float xi = (float)(r >> 8) * (1.0f / 16777216.0f);
    const float hit_dist = sc.med_neg_inv_density[m] * logf(xi);
    if (hit_dist > dist_inside_boundary) return false;
    t_hit = t1 + hit_dist / ray_len;
    return true;
}

// The rectangles and media of the list walk (they follow the spheres in the tie order), read from HBM.
// Primitives that only bound a medium are skipped (hitable.rs:523-533: the boundary lives inside the medium).
__device__ __forceinline__ void closest_hit_rects(const DevScene& sc, V3 o, V3 d, const MediumCtx& mc, float& tbest,
                                                  int& hit) {
    for (uint32_t r = 0; r < sc.n_rects; ++r) {
        float th;
        if (sc.prim_medium[sc.n_spheres + r] != RT_NO_XFORM_DEV) continue;
        V3 po = o, pd = d;
        const uint32_t xf = sc.prim_xform[sc.n_spheres + r];
        if (xf != RT_NO_XFORM_DEV) ray_to_object<true>(sc, xf, po, pd);
        if (rect_root(sc.rect_geo[2u * r], sc.rect_geo[2u * r + 1u], po, pd, 1e-3f, tbest, th)) {
            tbest = th;
            hit = (int)(sc.n_spheres + r);
        }
    }
    for (uint32_t m = 0; m < sc.n_media; ++m) {
        float th;
        if (medium_root<true>(sc, sc.prim_geo, m, o, d, 1e-3f, tbest, tbest, mc, th)) {
            tbest = th;
            hit = (int)(sc.n_prims + m);
        }
    }
}
// List walk over spheres of a scene with wrappers (read from HBM; the LDS tile loop assumes none).
__device__ __forceinline__ void closest_hit_spheres_general(const DevScene& sc, V3 o, V3 d, float& tbest, int& hit) {
    for (uint32_t s = 0; s < sc.n_spheres; ++s) {
        float th;
        if (sc.prim_medium[s] != RT_NO_XFORM_DEV) continue;
        V3 po = o, pd = d;
        const uint32_t xf = sc.prim_xform[s];
        if (xf != RT_NO_XFORM_DEV) ray_to_object<true>(sc, xf, po, pd);
        if (sphere_root(sc.sph_geo[s], po, pd, length_squared(pd), 1e-3f, tbest, th)) {
            tbest = th;
            hit = (int)s;
        }
    }
}

#define RT_BVH_BLOCK 1024 // threads per workgroup of k_intersect (one LDS copy of the tree)
#define RT_BVH_MAX_DEPTH 64u
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 48 // a wave refills from the queue when at least this many lanes are idle
                         // (measured: 8 -> 16.9 ms, 16 -> 15.6, 48 -> 14.7, 64 -> 16.0 per 337 M rays)
#endif
#define RT_ISECT_MAX_SHARDS 4u  // queue shards per k_intersect workgroup

// LDS carve of k_intersect, 4-wide nodes: 6 plane arrays + child ids (16 B per node each = 112 B per
// node), sphere list (16 B each), the per-lane traversal stack [level][thread] (u16; a node pushes up
// to 3 children, so 3 levels per tree level), 16 B of counters.
__host__ __device__ inline uint32_t bvh_stack_levels(const DevScene& sc) { return 3u * (sc.bvh4_depth ? sc.bvh4_depth : 1u) + 1u; }
__host__ __device__ inline size_t bvh_lds_bytes(const DevScene& sc, uint32_t block, bool lds_nodes = true) {
    return (lds_nodes ? (size_t)sc.n_bvh4_nodes * 112u + ((size_t)sc.n_spheres + 2u * sc.n_rects) * 16u : 0u) +
           (size_t)block * bvh_stack_levels(sc) * 2u + 16u;
}

// The small tables a leaf test of a general scene walks through: prim_xform -> xf_meta parent links -> xf_param is
// up to five DEPENDENT loads before the geometry is touched, and a medium adds its range / primitive list.  Read
// from HBM/L2 those round trips were most of a leaf's cost on final_scene's cloud of 1 000 instanced spheres;
// k_intersect<.., GLDS> copies them into LDS and traverses with a view that points at the copies.
struct GenTables {
    const float4* xf_param;
    const uint2* xf_meta;
    const uint32_t* prim_xform;
    const uint2* med_range;
    const uint32_t* med_prims;
    const uint2* med_xform;
    const float* med_neg_inv_density;
    uint32_t n_spheres, n_prims;
};
__device__ __forceinline__ GenTables tables_of(const DevScene& sc) {
    return GenTables{sc.xf_param, sc.xf_meta, sc.prim_xform, sc.med_range, sc.med_prims, sc.med_xform, sc.med_neg_inv_density,
                     sc.n_spheres, sc.n_prims};
}
__host__ __device__ inline size_t general_lds_bytes(const DevScene& sc) {
    return ((size_t)sc.n_xforms * 24u + (size_t)sc.n_xf_listed * 8u + (size_t)sc.n_prims * 4u + (size_t)sc.n_media * 20u + (size_t)sc.n_med_prims * 4u + 31u) & ~(size_t)15u;
}
template <int BLOCK>
__device__ __forceinline__ GenTables stage_general(const DevScene& sc, char* smem) {
    float4* xp = reinterpret_cast<float4*>(smem);
    uint2* xm = reinterpret_cast<uint2*>(xp + sc.n_xforms); // (with the lists of the chains deeper than RT_MAX_CHAIN behind it, rt_device.h)
    uint2* mr = xm + sc.n_xforms + sc.n_xf_listed;
    uint2* mx = mr + sc.n_media;
    uint32_t* px = reinterpret_cast<uint32_t*>(mx + sc.n_media);
    uint32_t* mp = px + sc.n_prims;
    float* md = reinterpret_cast<float*>(mp + sc.n_med_prims);
    for (uint32_t i = threadIdx.x; i < sc.n_xforms; i += BLOCK) xp[i] = sc.xf_param[i];
    for (uint32_t i = threadIdx.x; i < sc.n_xforms + sc.n_xf_listed; i += BLOCK) xm[i] = sc.xf_meta[i];
    for (uint32_t i = threadIdx.x; i < sc.n_media; i += BLOCK) mr[i] = sc.med_range[i], mx[i] = sc.med_xform[i], md[i] = sc.med_neg_inv_density[i];
    for (uint32_t i = threadIdx.x; i < sc.n_prims; i += BLOCK) px[i] = sc.prim_xform[i];
    for (uint32_t i = threadIdx.x; i < sc.n_med_prims; i += BLOCK) mp[i] = sc.med_prims[i];
    return GenTables{xp, xm, px, mr, mp, mx, md, sc.n_spheres, sc.n_prims};
}

struct BvhLds {
    const float4* pl[6]; // min_x, min_y, min_z, max_x, max_y, max_z of the 4 children
    const int4* id;
    const float4* geo;     // spheres (1 float4 each), then rectangles (2 float4 each)
    uint32_t n_spheres;
    GenTables gt;          // wrapper / medium tables of general scenes (HBM, or LDS with GLDS)
    unsigned short* stack; // this lane's column: stack[level * BLOCK]
};

// LDS_NODES = false: the tree and the primitive geometry stay in HBM (scenes whose tree does not fit the
// 160 KB of LDS, e.g. final_scene's 3.4 k primitives); only the traversal stacks live in LDS.  Keeping the top
// levels of the tree (breadth-first prefix, up to 270 of final_scene's 1 150 nodes) in the LDS that is left
// was measured: no gain (node fetches from L2 are hidden by 8 waves/SIMD; leaf work dominates).
template <int BLOCK, bool LDS_NODES>
__device__ __forceinline__ BvhLds stage_bvh(const DevScene& sc, char* smem) {
    const uint32_t n_nodes = sc.n_bvh4_nodes, n_sph = sc.n_spheres;
    float4* base = reinterpret_cast<float4*>(smem);
    BvhLds L;
    if (!LDS_NODES) {
#pragma unroll
        for (int a = 0; a < 6; ++a) L.pl[a] = sc.bvh4_p[a];
        L.id = sc.bvh4_id;
        L.geo = sc.prim_geo;
        L.n_spheres = n_sph;
        L.gt = tables_of(sc);
        L.stack = reinterpret_cast<unsigned short*>(smem) + threadIdx.x;
        return L;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        float4* dst = base + (size_t)a * n_nodes;
        for (uint32_t i = threadIdx.x; i < n_nodes; i += BLOCK) dst[i] = sc.bvh4_p[a][i];
        L.pl[a] = dst;
    }
    int4* ids = reinterpret_cast<int4*>(base + 6u * (size_t)n_nodes);
    for (uint32_t i = threadIdx.x; i < n_nodes; i += BLOCK) ids[i] = sc.bvh4_id[i];
    float4* geo = reinterpret_cast<float4*>(ids + n_nodes);
    for (uint32_t i = threadIdx.x; i < n_sph; i += BLOCK) geo[i] = sc.sph_geo[i];
    for (uint32_t i = threadIdx.x; i < 2u * sc.n_rects; i += BLOCK) geo[n_sph + i] = sc.rect_geo[i];
    L.id = ids;
    L.geo = geo;
    L.n_spheres = n_sph;
    L.gt = tables_of(sc);
    L.stack = reinterpret_cast<unsigned short*>(geo + n_sph + 2u * sc.n_rects) + threadIdx.x;
    return L;
}

// Exact test of world entry `s` (a BVH leaf or an entry of a primary-ray candidate list) and the order-independent
// accept.  Sphere-only scenes: Sphere::hit roots (hitable.rs:75-91).
template <bool RECTS, bool NEST = RECTS>
__device__ __forceinline__ void leaf_test(const BvhLds& L, int s, V3 o, V3 d, float a, uint32_t& pend, float& tbest,
                                          int& hit, ChainCache* cc = nullptr) {
    float th;
    // candidate root of this primitive (independent of tbest), then the order-independent accept
    bool ok;
    if (!RECTS) {
        ok = sphere_root(L.geo[s], o, d, a, 1e-3f, RT_FLT_MAX, th);
    } else {
        // general scene: a primitive that may sit below Translate / RotateY wrappers (t is unchanged by
        // them), or a medium.  A medium costs two searches over its boundary, ~12x a rectangle; tested
        // here it would stall the lanes of the wave that are at cheap leaves on every step.  It is only
        // noted in `pend` and tested after the traversal (media_step), when the lanes of the wave do so
        // together.  The winner rule is order-independent, so the result is the same.
        if ((uint32_t)s >= L.gt.n_prims) {
            pend |= 1u << (NEST ? min((uint32_t)s - L.gt.n_prims, 31u) : (uint32_t)s - L.gt.n_prims); // NEST, bit 31: "one of the media from 31 on" (media_step)
            ok = false;
        } else {
            ok = cc ? prim_root_cached<NEST>(L.gt, L.geo, (uint32_t)s, o, d, 1e-3f, RT_FLT_MAX, *cc, th)
                    : prim_root<NEST>(L.gt, L.geo, (uint32_t)s, o, d, 1e-3f, RT_FLT_MAX, th);
        }
    }
    if (ok && (th < tbest || (th == tbest && s > hit))) {
        tbest = th;
        hit = s;
    }
}

// One traversal step of one lane.  `cur` is an inner node: slab-test the 4 child boxes against [0, tbest], run the
// exact test of the hit leaf children (spheres: Sphere::hit roots, hitable.rs:75-91), continue with the nearest hit
// inner child and push the other hit inner children (unsorted: the visit count is the same as with a full sort,
// 7.32 vs 7.29 node visits per ray on sphere_scene; a binary tree needs 13.2 and measured 6 % slower).
// Returns true when the traversal of this ray has finished.
//
// The winner is the smallest accepted root with ties to the larger sphere index, i.e. exactly what
// the list walk of hitable.rs:117-132 returns (`t_max < root` rejects, so an equal root of a later
// sphere replaces an earlier one), independent of the visiting order.  Boxes are padded at build time
// (rt_bvh.h) so that a box is never culled when the exact test could accept the sphere inside it.
//
// The leaf children of a node are tested inside its node step instead of being pushed: the hit leaves are queued in
// two registers during the four box tests and tested in one loop behind them — one trip of the outer traversal
// loop (pop, branch, re-converge) less per leaf, one instance of the leaf code, and every trip of the leaf loop
// works for all lanes that still have a leaf.  k_intersect: config 2 -8 %, cornell_box -13 %, simple_light_scene
// -5 %, final_scene unchanged.  (Four inlined tests, one per child slot, were 2 % slower on spheres and 10-24 %
// slower on the larger leaf code of general scenes.)
//
// Slab arithmetic: t = b*inv - o*inv as one fused multiply-add per plane (the only place in the
// library that fuses; it is a culling test, not reference arithmetic; 4 % faster than (b-o)*inv.
// Packing two children per v_pk_fma_f32 was measured 4 % SLOWER than the 24 scalar fmas).
// The rounding of the precomputed o*inv puts an ABSOLUTE error of up to 2^-24*|o*inv| on every plane
// distance — large when the ray is nearly perpendicular to an axis — so the per-ray slack
// `eps` = 2.4e-7*max|o*inv| (2x the bound for entry + exit) is added to both limits, next to the 4e-6
// relative widening that covers the rounding of inv and of the fma itself.  For the few rays where that
// slack grows to the size of the scene (a direction component below ~1e-4: o*inv cancels against b*inv
// and nothing is culled any more — harmless for a tree in LDS, milliseconds for one wave walking 1 000
// nodes out of HBM) `exact` selects (b - o)*inv, which has no cancellation, with eps = 0.
// The near / far plane arrays of one ray (SORTED node steps, tree in LDS): per axis, the array of the child-box planes
// the ray reaches first and the array of those it leaves through, chosen once per ray by the sign of 1/d.
struct SlabPlanes {
    const float4 *nx, *ny, *nz, *fx, *fy, *fz;
};
// SORTED: `sp6` holds this ray's near / far plane arrays.  b*inv + c is monotone in b, so for a finite non-zero inv the
// smaller of the two plane distances of a slab is the one of the near plane: the six min / max per child that order them
// (24 of the ~120 vector instructions of a node step) go away.  Only the fma form relies on it — `exact` rays (inv
// infinite or NaN among them, see k_intersect) still order the two distances with min / max, which does not care which
// array a value came from.
template <int BLOCK, bool RECTS, bool SORTED = false, bool NEST = RECTS>
__device__ __forceinline__ bool bvh_step(const BvhLds& L, V3 o, V3 d, float ix, float iy, float iz, float nox, float noy,
                                         float noz, float eps, bool exact, float a, uint32_t& pend, int& cur, int& sp,
                                         float& tbest, int& hit, const SlabPlanes* sp6 = nullptr) {
    { // leaves never reach the stack (they are tested inside the node step below): cur is always a node
        RT_LANE_STAT(2, true);
        const float4 mnx = SORTED ? sp6->nx[cur] : L.pl[0][cur], mny = SORTED ? sp6->ny[cur] : L.pl[1][cur],
                     mnz = SORTED ? sp6->nz[cur] : L.pl[2][cur];
        const float4 mxx = SORTED ? sp6->fx[cur] : L.pl[3][cur], mxy = SORTED ? sp6->fy[cur] : L.pl[4][cur],
                     mxz = SORTED ? sp6->fz[cur] : L.pl[5][cur];
        const int4 id = L.id[cur];
        const float tb = __builtin_fmaf(tbest, 1.000004f, eps);
        uint32_t lq0 = 0u, lq1 = 0u; // the hit leaf children of this node, entry id + 1 in 16 bits each
        float best_t = RT_FLT_MAX;
        int best = (int)0x80000000;
#define RT_CHILD(K, IDK)                                                                                      \
    {                                                                                                         \
        const float x0 = RT_T(mnx.K, ix, nox, o.x), x1 = RT_T(mxx.K, ix, nox, o.x);                           \
        const float y0 = RT_T(mny.K, iy, noy, o.y), y1 = RT_T(mxy.K, iy, noy, o.y);                           \
        const float z0 = RT_T(mnz.K, iz, noz, o.z), z1 = RT_T(mxz.K, iz, noz, o.z);                           \
        const float tn = RT_ORDERED ? fmaxf(fmaxf(x0, y0), fmaxf(z0, 0.0f))                                   \
                                    : fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f)); \
        const float tf = RT_ORDERED ? fminf(fminf(x1, y1), z1)                                                \
                                    : fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));              \
        if (tn <= fminf(__builtin_fmaf(tf, 1.000004f, eps), tb) && IDK != (int)0x80000000) {                  \
            if (IDK < 0) { /* a leaf: queued for the leaf loop behind the four box tests */                   \
                lq1 = (lq1 << 16) | (lq0 >> 16);                                                               \
                lq0 = (lq0 << 16) | (uint32_t)(~IDK + 1);                                                      \
            } else if (tn < best_t) { /* new nearest: the previous nearest (if any) goes on the stack */      \
                if (best != (int)0x80000000) {                                                                \
                    L.stack[sp * BLOCK] = (unsigned short)best;                                               \
                    ++sp;                                                                                     \
                }                                                                                             \
                best_t = tn;                                                                                  \
                best = IDK;                                                                                   \
            } else {                                                                                          \
                L.stack[sp * BLOCK] = (unsigned short)IDK;                                                    \
                ++sp;                                                                                         \
            }                                                                                                 \
        }                                                                                                     \
    }
        if (!exact) {
#define RT_T(B, INV, NO, O) __builtin_fmaf(B, INV, NO)
#define RT_ORDERED SORTED
            RT_CHILD(x, id.x)
            RT_CHILD(y, id.y)
            RT_CHILD(z, id.z)
            RT_CHILD(w, id.w)
#undef RT_ORDERED
#undef RT_T
        } else { // rays almost parallel to an axis plane (see k_intersect): plane distances without cancellation
#define RT_T(B, INV, NO, O) (((B) - (O)) * (INV))
#define RT_ORDERED false
            RT_CHILD(x, id.x)
            RT_CHILD(y, id.y)
            RT_CHILD(z, id.z)
            RT_CHILD(w, id.w)
#undef RT_ORDERED
#undef RT_T
        }
#undef RT_CHILD
        // the leaves of the node in one loop: every trip tests one sphere per lane that still has one, instead of
        // four inlined tests that each run for the few lanes whose child k happens to be a hit leaf
        ChainCache cc;
        cc.xf = RT_NO_XFORM_DEV;
        while (lq0 != 0u) {
            RT_LANE_STAT(4, true);
            const int s = (int)(lq0 & 0xFFFFu) - 1;
            lq0 = (lq0 >> 16) | (lq1 << 16);
            lq1 >>= 16;
            leaf_test<RECTS, NEST>(L, s, o, d, a, pend, tbest, hit, RECTS ? &cc : nullptr);
        }
        if (best != (int)0x80000000) {
            cur = best;
            return false;
        }
    }
    if (sp == 0) return true;
    --sp;
    cur = (int)(short)L.stack[sp * BLOCK];
    return false;
}
// Tests the lowest pending medium of this lane against the best hit so far; returns true when none is left.
// The mask has a bit for each of the first 31 media of a scene.  Bit 31 stands for all the others together: a lane whose walk met one
// of them tests every medium from 31 on — the boundary searches of a medium the ray does not reach find nothing, and the winner rule
// does not depend on the order — so a scene may hold any number of media (the reference's world is a Vec, hitable.rs:104-105) and
// only those with more than 31 pay for it.
template <bool NEST>
__device__ __forceinline__ bool media_step(const BvhLds& L, V3 o, V3 d, const MediumCtx& mc, uint32_t n_media, uint32_t& pend,
                                           float& tbest, int& hit) {
    const uint32_t m0 = (uint32_t)__ffs((int)pend) - 1u;
    pend &= pend - 1u;
    const uint32_t m1 = !NEST || m0 < 31u ? m0 + 1u : n_media;
    for (uint32_t m = m0; m < m1; ++m) { // (one trip unless NEST)
        const int s = (int)(L.gt.n_prims + m);
        float th;
        if (medium_root<NEST>(L.gt, L.geo, m, o, d, 1e-3f, RT_FLT_MAX, tbest, mc, th) && (th < tbest || (th == tbest && s > hit))) {
            tbest = th;
            hit = s;
        }
    }
    return pend == 0u;
}
struct IntersectParams {
    uint32_t nq, cap;
    int depth;
    uint32_t q0, q1; // this launch covers the shards [q0, q1) (one of the two shard groups, rt_api.hip)
};

// Closest hit for every queued ray of the shards q = q0 + blockIdx.x, q0 + blockIdx.x + gridDim.x, ... below q1
// Persistent lanes: a lane whose traversal has finished writes its hit record and, once enough
// lanes of the wave are idle, the wave claims that many fresh rays from the workgroup's LDS work
// counter (the shards of the workgroup are concatenated into one virtual index space).  With no
// shading code the sphere-only kernel needs 58 VGPRs; the general one is held to 64 (second
// __launch_bounds__ argument = waves per SIMD on AMD; 16 B of scratch in the cold path) so that two
// 1024-thread workgroups (8 waves per SIMD) share a CU and hide each other's dependent node fetches
// — measured +15 % on cornell_box and +20 % on final_scene against 7 waves = one workgroup.
// GEN (depth 0): the ray is regenerated from its queue position instead of being loaded.
template <int BLOCK, bool GEN, bool RECTS, bool LDS_NODES, bool GLDS, bool NEST = false>
__global__ __launch_bounds__(BLOCK, 8) void k_intersect(DevScene sc, const float4* __restrict__ qa,
                                                     const float4* __restrict__ qb,
                                                     float2* __restrict__ qh, const uint32_t* __restrict__ in_counts,
                                                     IntersectParams ip, const GenParams* __restrict__ gpd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // virtual index space over this workgroup's shards (at most RT_ISECT_MAX_SHARDS, host-checked)
    uint32_t pre[RT_ISECT_MAX_SHARDS + 1];
    pre[0] = 0;
    uint32_t n_my = 0;
#pragma unroll
    for (uint32_t k = 0; k < RT_ISECT_MAX_SHARDS; ++k) {
        const uint32_t q = ip.q0 + blockIdx.x + k * gridDim.x;
        const uint32_t c = q < ip.q1 ? in_counts[q] : 0u;
        pre[k + 1] = pre[k] + c;
        if (q < ip.q1) n_my = k + 1;
    }
    const uint32_t total = pre[RT_ISECT_MAX_SHARDS];
    if (total == 0) return; // block-uniform
    if (GEN && !RECTS && gpd->lists && *gpd->n_overflow == 0u) return; // every pixel has a list: k_shade<GEN> finds all closest hits of depth 0
    BvhLds L = stage_bvh<BLOCK, LDS_NODES>(sc, smem);
    if (GLDS) L.gt = stage_general<BLOCK>(sc, smem + bvh_lds_bytes(sc, BLOCK, LDS_NODES));
    uint32_t* s_work = reinterpret_cast<uint32_t*>(smem + bvh_lds_bytes(sc, BLOCK, LDS_NODES) - 16u);
    if (threadIdx.x == 0) *s_work = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const bool no_geometry = sc.n_prims == 0u;
    bool exhausted = false; // wave-uniform: the workgroup has no unclaimed rays left
    bool has = false;
    bool exact = false; // this lane's ray uses the cancellation-free slab test (bvh_step)
    bool trav = false;  // general scenes: has && !trav = traversal done, media of `pend` still to test
    uint32_t pend = 0u;
    V3 o = splat(0.0f), d = v3(0.f, 0.f, 1.f);
    float ix = 0.f, iy = 0.f, iz = 0.f, nox = 0.f, noy = 0.f, noz = 0.f, eps = 0.f, a = 1.0f, tbest = RT_FLT_MAX;
    int hit = -1, cur = 0, sp = 0;
    size_t pos = 0;
    MediumCtx mc{0u, 0u, depth_counter_base(ip.depth)};
    // Sorted slab planes for sphere-only scenes with the tree in LDS.  General scenes: the six extra registers spill in the
    // 64-VGPR kernels (cornell_box +3 %, simple_light_scene +4 % measured); a tree in HBM would need six 64-bit pointers.
    constexpr bool SORTED = LDS_NODES && !RECTS;
    SlabPlanes planes{L.pl[0], L.pl[1], L.pl[2], L.pl[3], L.pl[4], L.pl[5]}; // SORTED: this ray's near / far arrays
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (!GEN) RT_LANE_STAT(0, has);
        if (n_idle >= RT_REFILL_MIN && !exhausted) {
            if (!GEN) RT_LANE_STAT(6, !has);
            uint32_t v0 = 0;
            if (lane == 0) v0 = atomicAdd(s_work, n_idle); // LDS atomic: claim n_idle rays
            v0 = __builtin_amdgcn_readfirstlane(v0);
            if (v0 + n_idle >= total) exhausted = true;
            const uint32_t v =
                v0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!has && v < total) {
                uint32_t k = 0;
#pragma unroll
                for (uint32_t t = 1; t < RT_ISECT_MAX_SHARDS; ++t) k += (t < n_my && v >= pre[t]) ? 1u : 0u;
                uint32_t off = v;
#pragma unroll
                for (uint32_t t = 1; t < RT_ISECT_MAX_SHARDS; ++t) off = (k == t) ? v - pre[t] : off;
                const uint32_t shard = ip.q0 + blockIdx.x + k * gridDim.x;
                pos = (size_t)shard * ip.cap + off;
#ifdef RT_DEBUG_QUEUE_BOUNDS
                if (off >= ip.cap || shard >= ip.q1) __builtin_trap();
#endif
                uint4 list = make_uint4(RT_LIST_OVERFLOW, 0u, 0u, 0u);
                // sphere-only scene with candidate lists: k_shade<GEN> finds the closest hit of a listed pixel itself;
                // only the rays of pixels whose list overflowed are traced (and recorded) here
                bool skip = false;
                if (GEN && !RECTS && gpd->lists) {
                    uint32_t pl; // local pixel of the slot (udiv_inv: a true 32-bit modulo is ~40 instructions)
                    udiv_inv(primary_idx_of(ip.nq, shard, off), gpd->npix, gpd->inv_npix, pl);
                    skip = (gpd->lists[pl].x & 0xFFFFu) != RT_LIST_OVERFLOW;
                }
                if (!skip) {
                if (GEN) {
                    uint32_t k0, k1, pl;
                    gen_primary(*gpd, primary_idx_of(ip.nq, shard, off), o, d, k0, k1, pl);
                    if (RECTS) mc.k0 = k0, mc.k1 = k1;
                    if (gpd->lists) list = gpd->lists[pl];
                } else {
                    const float4 ra = qa[RT_QSTRIDE * pos], rb = qb[RT_QSTRIDE * pos];
                    o = v3(ra.x, ra.y, ra.z);
                    d = v3(rb.x, rb.y, rb.z);
                    if (RECTS && sc.n_media) path_key_of_slot(*gpd, __float_as_uint(ra.w), mc.k0, mc.k1); // the free-path draw
                }
                // Slab constants (culling only, never reference arithmetic): v_rcp_f32, 1 ulp, instead of the IEEE division's
                // ~11 instructions per axis.  Both plane terms of a slab use the same ix, so its error is a relative error of
                // the plane distance (1.2e-7 against the 4e-6 the far distance is widened by); a zero or denormal component
                // gives inf, an infinite or NaN eps and so the cancellation-free path below, as 1.0f / d did for zero.
                ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
                nox = -(o.x * ix), noy = -(o.y * iy), noz = -(o.z * iz);
                eps = 2.4e-7f * fmaxf(fmaxf(fabsf(nox), fabsf(noy)), fabsf(noz));
                exact = !(eps <= sc.bvh_exact_eps); // also NaN (0 * inf)
                if (exact) eps = 0.0f;
                if (SORTED) { // a negative 1/d enters the slab of that axis through its max plane
                    planes.nx = ix < 0.0f ? L.pl[3] : L.pl[0], planes.fx = ix < 0.0f ? L.pl[0] : L.pl[3];
                    planes.ny = iy < 0.0f ? L.pl[4] : L.pl[1], planes.fy = iy < 0.0f ? L.pl[1] : L.pl[4];
                    planes.nz = iz < 0.0f ? L.pl[5] : L.pl[2], planes.fz = iz < 0.0f ? L.pl[2] : L.pl[5];
                }
                a = length_squared(d); // hitable.rs:77
                tbest = RT_FLT_MAX;
                hit = -1;
                cur = 0;
                sp = 0;
                has = !no_geometry;
                trav = true;
                pend = 0u;
                if (no_geometry) qh[pos] = make_float2(RT_FLT_MAX, __int_as_float(-1));
                if (GEN && has && (list.x & 0xFFFFu) != RT_LIST_OVERFLOW) {
                    // primary ray of a pixel with a candidate list (k_primary_lists): the listed entries instead of the tree
                    const uint32_t n_list = list.x & 0xFFFFu;
                    if (n_list > 0u) leaf_test<RECTS, NEST>(L, (int)(list.x >> 16), o, d, a, pend, tbest, hit);
                    if (n_list > 1u) leaf_test<RECTS, NEST>(L, (int)(list.y & 0xFFFFu), o, d, a, pend, tbest, hit);
                    if (n_list > 2u) leaf_test<RECTS, NEST>(L, (int)(list.y >> 16), o, d, a, pend, tbest, hit);
                    if (n_list > 3u) leaf_test<RECTS, NEST>(L, (int)(list.z & 0xFFFFu), o, d, a, pend, tbest, hit);
                    if (n_list > 4u) leaf_test<RECTS, NEST>(L, (int)(list.z >> 16), o, d, a, pend, tbest, hit);
                    if (n_list > 5u) leaf_test<RECTS, NEST>(L, (int)(list.w & 0xFFFFu), o, d, a, pend, tbest, hit);
                    if (n_list > 6u) leaf_test<RECTS, NEST>(L, (int)(list.w >> 16), o, d, a, pend, tbest, hit);
                    if (RECTS && pend) {
                        trav = false; // its media are tested in the media phase below
                    } else {
                        qh[pos] = make_float2(tbest, __int_as_float(hit));
                        has = false;
                    }
                }
                } // !skip
            }
        }
        if (!__any(has)) {
            if (exhausted) break;
            continue;
        }
        if (RECTS) {
            // media phase: when no lane of the wave has tree work left, or half the wave is waiting
            const unsigned long long waiting = __ballot(has && !trav);
            if (waiting && (__popcll(waiting) >= 32 || !__any(has && trav))) {
                if (has && !trav && media_step<NEST>(L, o, d, mc, sc.n_media, pend, tbest, hit)) {
                    qh[pos] = make_float2(tbest, __int_as_float(hit));
                    has = false;
                }
                continue;
            }
        }
        if (has && trav && bvh_step<BLOCK, RECTS, SORTED, NEST>(L, o, d, ix, iy, iz, nox, noy, noz, eps, exact, a, pend, cur, sp, tbest, hit, &planes)) {
            if (RECTS && pend) {
                trav = false;
            } else {
                qh[pos] = make_float2(tbest, __int_as_float(hit));
                has = false;
            }
        }
    }
}

// List-walk closest hit: one workgroup per shard, sphere list streamed through LDS tiles.
__global__ __launch_bounds__(256) void k_intersect_list(DevScene sc, const float4* __restrict__ qa,
                                                        const float4* __restrict__ qb,
                                                        float2* __restrict__ qh, const uint32_t* __restrict__ in_counts,
                                                        IntersectParams ip, const GenParams* __restrict__ gpd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_geo = reinterpret_cast<float4*>(smem);
    const uint32_t q = ip.q0 + blockIdx.x;
    const uint32_t count = in_counts[q];
    if (count == 0) return;
    const uint32_t n_sph = sc.n_spheres;
    const bool single_tile = n_sph <= RT_SPHERE_TILE;
    if (single_tile) {
        for (uint32_t i = threadIdx.x; i < n_sph; i += 256u) s_geo[i] = sc.sph_geo[i];
        __syncthreads();
    }
    const size_t qbase = (size_t)q * ip.cap;
    for (uint32_t base = 0; base < count; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < count;
        V3 o = splat(0.0f), d = v3(0.f, 0.f, 1.f);
        if (active) {
            const float4 ra = qa[RT_QSTRIDE * (qbase + i)], rb = qb[RT_QSTRIDE * (qbase + i)];
            o = v3(ra.x, ra.y, ra.z);
            d = v3(rb.x, rb.y, rb.z);
        }
        float tbest = RT_FLT_MAX;
        int hit = -1;
        const float a = length_squared(d);
        if (sc.n_xforms || sc.n_media) {
            closest_hit_spheres_general(sc, o, d, tbest, hit);
        } else if (single_tile) {
            closest_hit_tile(s_geo, n_sph, 0u, o, d, a, tbest, hit);
        } else {
            for (uint32_t t0 = 0; t0 < n_sph; t0 += RT_SPHERE_TILE) {
                const uint32_t n = min(RT_SPHERE_TILE, n_sph - t0);
                __syncthreads();
                for (uint32_t k = threadIdx.x; k < n; k += 256u) s_geo[k] = sc.sph_geo[t0 + k];
                __syncthreads();
                closest_hit_tile(s_geo, n, t0, o, d, a, tbest, hit);
            }
        }
        MediumCtx mc{0u, 0u, depth_counter_base(ip.depth)};
        if (active && sc.n_media) path_key_of_slot(*gpd, __float_as_uint(qa[RT_QSTRIDE * (qbase + i)].w), mc.k0, mc.k1);
        closest_hit_rects(sc, o, d, mc, tbest, hit);
        if (active) qh[qbase + i] = make_float2(tbest, __int_as_float(hit));
    }
}

struct ShadeParams {
    uint32_t nq, cap;
    int depth, max_depth;
    uint32_t sort;             // 1: class-sort every 512-ray block before shading it; 0: queue order (depth 0)
    uint32_t russian_roulette; // main.rs:49-53 (commented out in the reference), RT_FLAG_RUSSIAN_ROULETTE
    uint32_t q0;               // first shard of this launch (shard groups, rt_api.hip)
};

// Phase clocks (diagnostic builds only, -DRT_PROFILE_PHASES; scripts/gpu_phase_stats.py): s_memtime ticks a wave of the
// class-sorting k_shade spends in  [0] the whole kernel  [1] the sort of a block (hit-record loads included)  [2] waiting for the
// first segment's rays  [3] all-miss segments  [4] segments with hits;  [5] blocks  [6] all-miss segments  [7] segments with hits
// [8] (in front of [1]) waiting for the stores of the block before  [9] (inside [1]) the eight hit-record loads, issue to arrival.
#ifdef RT_PROFILE_PHASES
__device__ unsigned long long g_phase_stats[16];
#define RT_PHASE_CLOCK() __builtin_amdgcn_s_memtime()
#else
#define RT_PHASE_CLOCK() 0ull
#endif

// Shading half of the step: workgroup q owns shard q (reads it, appends survivors to shard q of the output queue
// through an LDS counter, publishes the new count with a plain store).
//
// The 13 material branches and 4 texture kinds diverge badly when a wave holds arbitrary rays (VALU lane utilisation
// 34 %, profiles/round1).  From depth 1 on every WAVE therefore takes blocks of RT_SORT_N consecutive rays of the shard
// and counting-sorts their hit records by shading class (DevScene::sph_class; per-wave LDS histogram, scan over the 64
// classes by one lane each, sorted records + their queue positions parked in LDS; no ray data moves), then shades the
// block in class order, 64 rays at a time: a wave runs one material except at class boundaries.  The sort is wave-local
// on purpose — no workgroup barrier and no second pass over the hit records: past depth 0 this kernel is bound by HBM
// (profiles/round2: 4.2-4.9 TB/s of FETCH+WRITE with 70 % of the wave-cycles waiting on memory), so the bytes per ray
// are what counts (8 B hit record + 48 B ray in, 48 B per survivor or 16 B of radiance out) and every wave keeps the
// next segment's ray gathers in flight while it shades the current one.  Results do not depend on the processing order.
// (Tried and measured worse, round 2: sorting in k_intersect's epilogue with the order handed over through HBM, +18 B
// per ray and 3 ms per 128 spp; separate kernels for the cheap and the expensive classes at 8 and 4 waves per SIMD —
// both kernels fetch nearly every line of a chunk, +50 % bytes; 16 bank-conflict-free copies of the Perlin gradients in
// LDS — no change: the turbulence is bound by VALU issue, not by its LDS gathers; an instantiation without the pbr.rs
// materials at 96 VGPRs / 5 waves per SIMD for scenes that use none — no change.)
//
// PERLIN_LDS: the Perlin gradient and permutation tables (texture.rs:53-58; 5.5 KB per set) are staged into LDS.
// GEN (depth 0): T = 1, slot = path index, and the ray is regenerated from its queue position.  In sphere-only
// scenes with candidate lists (k_primary_lists) the closest hit is found right here from the pixel's list — the
// exact Sphere::hit roots of its <= 7 entries, same winner rule as the tree — so a primary ray is generated once and
// no hit record travels through HBM; k_intersect<GEN> then only traces the rays of pixels whose list overflowed.
#define RT_PERLIN_LDS_MAX_SETS 4u
#ifndef RT_SORT_N
#define RT_SORT_N 512u  // rays per wave-local counting sort (8 segments of 64)
#endif
#define RT_NCLASS 64u
#define RT_CLASS_LDS_MAX 4096u // scenes up to this many world entries keep their class table in LDS
#define RT_SHADE_WAVE_LDS (RT_NCLASS * 4u + RT_SORT_N * 8u + RT_SORT_N * 2u) // histogram, sorted records, positions
__host__ __device__ inline size_t shade_lds_bytes(uint32_t n_entries, uint32_t n_perlin_lds, uint32_t n_fused_spheres, bool sort) {
    size_t b = 16u; // survivor counter
    if (sort) b += 4u * RT_SHADE_WAVE_LDS;
    if (sort && n_entries <= RT_CLASS_LDS_MAX) b += ((size_t)n_entries + 15u) & ~(size_t)15u;
    b += (size_t)n_perlin_lds * (4096u + 1536u);
    b += (size_t)n_fused_spheres * 16u;
    return (b + 15u) & ~(size_t)15u;
}
template <bool PERLIN_LDS, bool GEN, bool RECTS, bool NEST = false> // NEST (general scenes only): rt_device.h, wrapper chains and media as loops
#ifndef RT_GEN_WAVES
#define RT_GEN_WAVES 4 // waves per SIMD the depth-0 instantiations are compiled for (98 VGPR: 5 fit).  Round 2: 4 / 5 / 6 no difference.
                       // Round 3 (cheaper draws): alone on the chip, 6 (80 VGPR, 8 B of scratch) is 2.4-2.7 % faster at depth 0 on the
                       // sphere scenes; in the production frame, where the two chains overlap, its dispatches last 12.3 ms instead
                       // of 10.3 and the frame takes the same 56.5 ms (profiles/round3/ab_gen_waves.txt): left at 4
#endif
#ifndef RT_SORTED_WAVES
#define RT_SORTED_WAVES 4 // waves per SIMD the depth >= 1 instantiations are compiled for (5 / 6, with and without the pbr.rs
                          // materials: no gain, profiles/round4/ab_shade_waves_*.txt)
#endif
__global__ __launch_bounds__(256, GEN ? RT_GEN_WAVES : RT_SORTED_WAVES) void k_shade(DevScene sc, Queue qin, const float2* __restrict__ qh, Queue qout,
                                               const uint32_t* __restrict__ in_counts, uint32_t* __restrict__ out_counts,
                                               float* __restrict__ rad, ShadeParams tp,
                                               unsigned long long* __restrict__ stats,
                                               const GenParams* __restrict__ gpd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t q = tp.q0 + blockIdx.x;
    const uint32_t count = in_counts[q];
    if (count == 0) return; // out_counts[q] stays 0 (cleared per slice)
    uint32_t* s_out = reinterpret_cast<uint32_t*>(smem);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool sort = !GEN && tp.sort != 0u;
    size_t lds_off = 16u;
    // this wave's sort scratch
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + lds_off + (size_t)w * RT_SHADE_WAVE_LDS);
    float2* s_rec = reinterpret_cast<float2*>(s_hist + RT_NCLASS);
    unsigned short* s_pos = reinterpret_cast<unsigned short*>(s_rec + RT_SORT_N);
    const uint8_t* cls = sc.sph_class;
    if (sort) {
        lds_off += 4u * RT_SHADE_WAVE_LDS;
        const uint32_t n_ent = sc.n_prims + sc.n_media;
        if (n_ent <= RT_CLASS_LDS_MAX) {
            uint8_t* lc = reinterpret_cast<uint8_t*>(smem + lds_off);
            for (uint32_t i = threadIdx.x; i < n_ent; i += 256u) lc[i] = sc.sph_class[i];
            cls = lc;
            lds_off += ((size_t)n_ent + 15u) & ~(size_t)15u;
        }
    }
    PerlinTables pt{sc.perlin_vec, sc.perlin_perm2};
    if (PERLIN_LDS) {
        float4* lv = reinterpret_cast<float4*>(smem + lds_off);
        unsigned short* lp = reinterpret_cast<unsigned short*>(lv + sc.n_perlin * 256u);
        for (uint32_t i = threadIdx.x; i < sc.n_perlin * 256u; i += 256u) lv[i] = sc.perlin_vec[i];
        for (uint32_t i = threadIdx.x; i < sc.n_perlin * 768u; i += 256u) lp[i] = sc.perlin_perm2[i];
        pt = PerlinTables{lv, lp};
        lds_off += (size_t)sc.n_perlin * (4096u + 1536u);
    }
    // depth 0 with candidate lists in a sphere-only scene: closest hit from the list, sphere geometry in LDS
    const bool fused = GEN && !RECTS && gpd->lists != nullptr;
    const float4* s_geo = reinterpret_cast<const float4*>(smem + lds_off);
    if (fused) {
        float4* g = reinterpret_cast<float4*>(smem + lds_off);
        for (uint32_t i = threadIdx.x; i < sc.n_spheres; i += 256u) g[i] = sc.sph_geo[i];
    }
    if (threadIdx.x == 0) *s_out = 0u;
    __syncthreads();
    uint32_t n_fetch = 0, n_bad = 0;
    const size_t qbase = (size_t)q * tp.cap;
#ifdef RT_PROFILE_PHASES
    unsigned long long ph[10] = {};
    const unsigned long long ph_t0 = RT_PHASE_CLOCK();
#endif
    // blocks of RT_SORT_N rays go round-robin to the four waves; a wave never waits for another one.  A short shard
    // (the deep bounces hold a few hundred rays per shard) is cut into four blocks so that every wave has work.
    const uint32_t bs = count >= 4u * RT_SORT_N ? RT_SORT_N : ((count + 255u) / 256u) * 64u;
    for (uint32_t base = w * bs; base < count; base += 4u * bs) {
        const uint32_t n_here = min(bs, count - base);
#ifdef RT_PROFILE_PHASES
        const unsigned long long ph_a0 = RT_PHASE_CLOCK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (diagnostic: the stores of the block before, on their own)
        const unsigned long long ph_a = RT_PHASE_CLOCK();
        ph[8] += ph_a - ph_a0;
#endif
        if (sort) {
            // ---- wave-local counting sort of the block's hit records by shading class ---------------
            s_hist[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // LDS operations of one wave complete in order; this
                                                                   // keeps the compiler from moving them across
            float2 h[RT_SORT_N / 64u];
            uint32_t kr[RT_SORT_N / 64u]; // key | rank << 8
            // s_waitcnt vmcnt(0) as the BUILTIN, so that the compiler's own wait insertion knows that nothing is in flight: without
            // it the eight loads below are cut in two groups by a wait (their registers were prefetch destinations of the block
            // before), two memory round trips of ~25 000 cycles instead of one.  What is still outstanding here are the stores of
            // the block before, and they have long drained (63 cycles on average: profiles/round5/phases_k_shade.txt).
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (uint32_t k = 0; k < RT_SORT_N / 64u; ++k) {
                const uint32_t j = k * 64u + lane;
                h[k] = j < n_here ? qh[qbase + base + j] : make_float2(0.0f, 0.0f);
            }
#ifdef RT_PROFILE_PHASES
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (diagnostic: the eight hit-record loads, on their own)
            ph[9] += RT_PHASE_CLOCK() - ph_a;
#endif
#pragma unroll
            for (uint32_t k = 0; k < RT_SORT_N / 64u; ++k) {
                const uint32_t j = k * 64u + lane;
                kr[k] = 0u;
                if (j < n_here) {
                    const int hit = __float_as_int(h[k].y);
#ifdef RT_DEBUG_QUEUE_BOUNDS // debug builds: a hit record names a world entry of the scene (a kernel that left qh unwritten would not)
                    if (hit >= (int)(sc.n_prims + sc.n_media)) __builtin_trap();
#endif
                    const uint32_t key = hit < 0 ? sc.key_miss : (uint32_t)cls[hit];
                    kr[k] = key | (atomicAdd(&s_hist[key], 1u) << 8);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            { // exclusive scan of the 64 class counts, one lane per class
                const uint32_t v = s_hist[lane];
                uint32_t incl = v;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = __shfl_up(incl, off);
                    if ((int)lane >= off) incl += t;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                s_hist[lane] = incl - v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
            for (uint32_t k = 0; k < RT_SORT_N / 64u; ++k) {
                const uint32_t j = k * 64u + lane;
                if (j < n_here) {
                    const uint32_t dst = s_hist[kr[k] & 255u] + (kr[k] >> 8);
                    s_rec[dst] = h[k];
                    s_pos[dst] = (unsigned short)j;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
        // ---- shade the block, 64 rays at a time; the next segment's rays are in flight meanwhile ---------
        float2 hA = make_float2(0.0f, 0.0f), rcA = hA;
        float4 raA = make_float4(0.f, 0.f, 0.f, 0.f), rbA = raA;
        auto fetch = [&](uint32_t seg, float2& h, float4& ra, float4& rb, float2& rc) {
            const uint32_t j = seg + lane;
            if (!GEN && j < n_here) {
                uint32_t pj = j;
                if (sort) h = s_rec[j], pj = s_pos[j];
                else h = qh[qbase + base + j];
                const size_t r = qbase + base + pj;
                ra = qin.a[RT_QSTRIDE * r], rb = qin.b[RT_QSTRIDE * r], rc = qin.c[r];
            }
        };
        fetch(0u, hA, raA, rbA, rcA);
#ifdef RT_PROFILE_PHASES
        const unsigned long long ph_b = RT_PHASE_CLOCK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the diagnostic build exposes the first gather's latency as its own phase)
        unsigned long long ph_c = RT_PHASE_CLOCK();
        ph[1] += ph_b - ph_a, ph[2] += ph_c - ph_b, ph[5] += 1;
#endif
        for (uint32_t seg = 0; seg < n_here; seg += 64u) {
            float2 hB = make_float2(0.0f, 0.0f), rcB = hB;
            float4 raB = make_float4(0.f, 0.f, 0.f, 0.f), rbB = raB;
            bool prefetched = false;
            // the next segment's rays: requested from inside shade(), behind the loads of this segment's primitive records
            auto prefetch = [&]() {
                if (!prefetched && seg + 64u < n_here) fetch(seg + 64u, hB, raB, rbB, rcB);
                prefetched = true;
            };
            const uint32_t j = seg + lane;
            RT_LANE_STAT(18, j < n_here);
            // A segment of misses only — after the class sort most segments past depth 0 are (61 % of the rays of sphere_scene miss
            // everything, and "miss" is one class): sky x throughput into the path's radiance slot and nothing else.  No path key (a
            // miss draws no random number), no primitive record, no survivors to compact; the radiance is the one shade() returns.
            if (!GEN && __all(j >= n_here || __float_as_int(hA.y) < 0)) {
                if (j < n_here) {
                    const V3 d = v3(rbA.x, rbA.y, rbA.z);
                    V3 Lr = splat(0.0f);
                    if (!near_one(d)) ++n_bad; // main.rs:39 assert!: the reference panics; the path is dropped
                    else Lr = v3(rbA.w, rcA.x, rcA.y) * sky_value(sc, d, n_fetch); // L = T_n * sky, main.rs:58
                    rad_store(rad, __float_as_uint(raA.w), Lr.x, Lr.y, Lr.z);
                }
                prefetch();
                hA = hB, raA = raB, rbA = rbB, rcA = rcB;
#ifdef RT_PROFILE_PHASES
                { const unsigned long long t = RT_PHASE_CLOCK(); ph[3] += t - ph_c, ph[6] += 1, ph_c = t; }
#endif
                continue;
            }
            bool alive = false;
            Bounce bo;
            bo.o = bo.d = bo.attenuation = splat(0.0f);
            V3 T = splat(0.0f);
            float rr_threshold = 0.0f;
            uint32_t slot = 0, k0 = 0, k1 = 0;
            if (j < n_here) {
                float2 h = hA;
                V3 o, d;
                if (GEN) { // depth 0: T = 1, slot = path index, ray regenerated (bitwise the one k_intersect would trace)
                    uint32_t pl;
                    slot = primary_idx_of(tp.nq, q, base + j);
                    gen_primary(*gpd, slot, o, d, k0, k1, pl);
                    T = splat(1.0f);
                    uint4 list = make_uint4(RT_LIST_OVERFLOW, 0u, 0u, 0u);
                    if (fused) list = gpd->lists[pl];
                    const uint32_t n_list = list.x & 0xFFFFu;
                    RT_LANE_STAT(14, fused && n_list != RT_LIST_OVERFLOW); // waves at depth 0 / lanes with a candidate list
                    if (fused && n_list != RT_LIST_OVERFLOW) {
                        // leaf_test<false> over the listed entries: Sphere::hit roots (hitable.rs:75-91), order-independent accept
                        const SharedRcp ra = shared_rcp(length_squared(d));
                        float tbest = RT_FLT_MAX;
                        int hit = -1;
                        const uint32_t ids[RT_LIST_MAX] = {list.x >> 16, list.y & 0xFFFFu, list.y >> 16, list.z & 0xFFFFu,
                                                           list.z >> 16, list.w & 0xFFFFu, list.w >> 16};
#pragma unroll
                        for (uint32_t t = 0; t < RT_LIST_MAX; ++t) {
                            float th;
                            const int s = (int)ids[t];
                            if (t < n_list) RT_LANE_STAT(12, true); // (profiling builds: trips of the list test and lanes in them)
                            if (t < n_list && sphere_root(s_geo[s], o, d, ra, 1e-3f, RT_FLT_MAX, th) &&
                                (th < tbest || (th == tbest && s > hit))) {
                                tbest = th;
                                hit = s;
                            }
                        }
                        h = make_float2(tbest, __int_as_float(hit));
                    } else {
                        h = qh[qbase + base + j];
                    }
                } else {
                    o = v3(raA.x, raA.y, raA.z), d = v3(rbA.x, rbA.y, rbA.z);
                    T = v3(rbA.w, rcA.x, rcA.y);
                    slot = __float_as_uint(raA.w);
                    path_key_of_slot(*gpd, slot, k0, k1);
                }
                V3 Lr = splat(0.0f);
                if (!near_one(d)) { // main.rs:39 assert!: the reference panics; the path is dropped
                    ++n_bad;
                } else {
#ifdef RT_DEBUG_QUEUE_BOUNDS // debug builds: the record fetch of shade() is indexed by whatever the hit record holds
                    if (__float_as_int(h.y) >= (int)(sc.n_prims + sc.n_media)) __builtin_trap();
#endif
                    Rng rng{k0, k1, depth_counter_base(tp.depth)};
                    bo = shade<RECTS, decltype(prefetch), NEST>(sc, pt, o, d, __float_as_int(h.y), h.x, rng, n_fetch, prefetch);
                    if (bo.alive && tp.russian_roulette) { // main.rs:49-53
                        const float rr = rng.next();
                        rr_threshold = fmaxf(bo.attenuation.x, fmaxf(bo.attenuation.y, bo.attenuation.z)); // max_element
                        if (!(rr < rr_threshold)) bo.alive = false; // the path ends without a further contribution
                    }
                    if (bo.alive) {
                        alive = tp.depth < tp.max_depth; // survivors of the last depth return 0, main.rs:40-42
                    } else {
                        Lr = T * bo.radiance; // L = T_n * (emitted | sky)
                    }
                }
                // one 16 B store: beyond depth 0 the slots of a wave are scattered
                if (!alive) rad_store(rad, slot, Lr.x, Lr.y, Lr.z);
            }
            prefetch(); // (lanes that shaded nothing: beyond the block, or a direction the reference would panic on)
            // wave64 compaction: ballot + prefix popcount; the wave claims its slots from the LDS counter
            const unsigned long long mask = __ballot(alive);
            if (mask) {
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(s_out, (uint32_t)__popcll(mask));
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                if (alive) {
                    const uint32_t rk =
                        __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    const size_t pos = qbase + wbase + rk;
#ifdef RT_DEBUG_QUEUE_BOUNDS // debug builds: a shard never receives more rays than it held (capacity `cap`)
                    if (wbase + rk >= tp.cap) __builtin_trap();
#endif
                    V3 Tn = T * bo.attenuation;
                    if (tp.russian_roulette) Tn = Tn / rr_threshold; // (T * a) / threshold
                    qout.a[RT_QSTRIDE * pos] = make_float4(bo.o.x, bo.o.y, bo.o.z, __uint_as_float(slot));
                    qout.b[RT_QSTRIDE * pos] = make_float4(bo.d.x, bo.d.y, bo.d.z, Tn.x);
                    qout.c[pos] = make_float2(Tn.y, Tn.z);
                }
            }
            hA = hB, raA = raB, rbA = rbB, rcA = rcB;
#ifdef RT_PROFILE_PHASES
            { const unsigned long long t = RT_PHASE_CLOCK(); ph[4] += t - ph_c, ph[7] += 1, ph_c = t; }
#endif
        }
    }
#ifdef RT_PROFILE_PHASES
    if (sort && lane == 0) {
        ph[0] = RT_PHASE_CLOCK() - ph_t0;
        for (int k = 0; k < 10; ++k) atomicAdd(&g_phase_stats[k], ph[k]);
    }
#endif
    __syncthreads();
    if (threadIdx.x == 0) out_counts[q] = *s_out;
    // rare-event counters: one atomic per wave, only when nonzero
    for (int off = 32; off > 0; off >>= 1) {
        n_fetch += __shfl_down(n_fetch, off);
        n_bad += __shfl_down(n_bad, off);
    }
    if (lane == 0) {
        if (n_fetch) atomicAdd(&stats[0], (unsigned long long)n_fetch);
        if (n_bad) atomicAdd(&stats[1], (unsigned long long)n_bad);
    }
}

// Sums the slice's samples of each pixel in sample order (main.rs:95-97) onto the running sum.
__global__ __launch_bounds__(256) void k_resolve(const float* __restrict__ rad, float* __restrict__ acc, uint32_t npix,
                                                 uint32_t s_count) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= npix) return;
    float r = acc[3 * (size_t)p], g = acc[3 * (size_t)p + 1], b = acc[3 * (size_t)p + 2];
    for (uint32_t s = 0; s < s_count; ++s) {
        const float* src = rad + RT_RAD_FLOATS * ((size_t)s * npix + p);
        const float sx = src[0], sy = src[1], sz = src[2];
        r += sx;
        g += sy;
        b += sz;
    }
    acc[3 * (size_t)p] = r, acc[3 * (size_t)p + 1] = g, acc[3 * (size_t)p + 2] = b;
}

// main.rs:98-105,127.  out_f32: linear mean (before gamma), local row order; out_u8: gamma 2,
// *255.99 saturating cast, rows flipped.  sqrtf is the correctly rounded value of powf(x, 0.5).
__global__ __launch_bounds__(256) void k_finalize(const float* __restrict__ acc, float* __restrict__ out_f32,
                                                  uint8_t* __restrict__ out_u8, uint32_t nx, uint32_t rows, uint32_t spp,
                                                  uint32_t tiles_per_row, uint32_t tile_pixels) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x; // output pixel, row-major in the shard's local rows
    if (p >= nx * rows) return;
    const float fs = (float)spp;
    const size_t ap = local_of_pixel(nx, tiles_per_row, tile_pixels, p % nx, p / nx); // where the path slots keep this pixel (GenParams)
    float c[3] = {acc[3 * ap] / fs, acc[3 * ap + 1] / fs, acc[3 * ap + 2] / fs};
    if (out_f32) {
        out_f32[3 * (size_t)p] = c[0], out_f32[3 * (size_t)p + 1] = c[1], out_f32[3 * (size_t)p + 2] = c[2];
    }
    if (out_u8) {
        const uint32_t lj = p / nx, i = p - lj * nx;
        const size_t dst = ((size_t)(rows - 1u - lj) * nx + i) * 3u;
        for (int k = 0; k < 3; ++k) {
            float g = sqrtf(c[k]) * 255.99f;
            uint32_t u = (g == g && g > 0.0f) ? (g >= 255.0f ? 255u : (uint32_t)g) : 0u;
            out_u8[dst + k] = (uint8_t)u;
        }
    }
}

// Sums the per-shard counters of every depth into 64-bit totals (ray statistics).
__global__ __launch_bounds__(256) void k_accum_counts(const uint32_t* __restrict__ counts, uint32_t nq, uint32_t n_depths,
                                                      unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long part[4];
    const uint32_t d = blockIdx.x;
    if (d >= n_depths) return;
    unsigned long long s = 0;
    for (uint32_t k = threadIdx.x; k < nq; k += 256u) s += counts[(size_t)d * nq + k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) totals[d] += part[0] + part[1] + part[2] + part[3];
}

// Test hook (rt_debug_bounce, RT_FLAG_PRODUCTION_KERNELS): caller-given rays into the queue, in the shard lattice of
// the primary rays, T = 1, slot = ray index.  The production k_intersect / k_shade then run on them.
__global__ __launch_bounds__(256) void k_debug_fill(GenParams gp, Queue q, const float* __restrict__ in_o,
                                                    const float* __restrict__ in_d) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= gp.n_rays) return;
    const uint32_t chunk = idx >> 8;
    const size_t pos = (size_t)(chunk % gp.nq) * gp.cap + (size_t)(chunk / gp.nq) * 256u + (idx & 255u);
    q.a[RT_QSTRIDE * pos] = make_float4(in_o[3 * idx], in_o[3 * idx + 1], in_o[3 * idx + 2], __uint_as_float(idx));
    q.b[RT_QSTRIDE * pos] = make_float4(in_d[3 * idx], in_d[3 * idx + 1], in_d[3 * idx + 2], 1.0f);
    q.c[pos] = make_float2(1.0f, 1.0f);
}


// Test hook (rt_debug_arithmetic): div_shared, sqrt_ns and the saturating conversions as the kernels use them.
__global__ __launch_bounds__(256) void k_debug_arithmetic(uint32_t op, uint32_t n, const float* __restrict__ x, const float* __restrict__ a,
                                                          float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (op == 0u) out[i] = div_shared(x[i], shared_rcp(a[i]));
    else if (op == 1u) out[i] = sqrt_ns(x[i]);
    else if (op == 2u) out[i] = __int_as_float(sat_i32(x[i]));
    else out[i] = __uint_as_float(sat_u32(x[i]));
}

// Test hook: one bounce for caller-given rays, no queues (rt_debug_bounce).
template <int BLOCK, bool USE_BVH, bool LDS_NODES>
__global__ __launch_bounds__(BLOCK) void k_debug_bounce(DevScene sc, uint32_t n, int depth, const float* __restrict__ in_o,
                                                        const float* __restrict__ in_d, const uint32_t* __restrict__ in_key,
                                                        int* __restrict__ out_hit, float* __restrict__ out_t,
                                                        float* __restrict__ out_rad, float* __restrict__ out_att,
                                                        float* __restrict__ out_o, float* __restrict__ out_d,
                                                        uint8_t* __restrict__ out_alive) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n;
    V3 o = splat(0.0f), d = v3(0.f, 0.f, 1.f);
    if (active) {
        o = v3(in_o[3 * i], in_o[3 * i + 1], in_o[3 * i + 2]);
        d = v3(in_d[3 * i], in_d[3 * i + 1], in_d[3 * i + 2]);
    }
    float tbest = RT_FLT_MAX;
    int hit = -1;
    const float a = length_squared(d);
    if (USE_BVH) {
        const BvhLds L = stage_bvh<BLOCK, LDS_NODES>(sc, smem);
        __syncthreads();
        if (active && sc.n_prims) {
            const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
            const float nox = -(o.x * ix), noy = -(o.y * iy), noz = -(o.z * iz);
            float eps = 2.4e-7f * fmaxf(fmaxf(fabsf(nox), fabsf(noy)), fabsf(noz));
            const bool exact = !(eps <= sc.bvh_exact_eps);
            if (exact) eps = 0.0f;
            int cur = 0, sp = 0;
            const MediumCtx mc{active ? in_key[2 * i] : 0u, active ? in_key[2 * i + 1] : 0u, depth_counter_base(depth)};
            uint32_t pend = 0u;
            while (!bvh_step<BLOCK, true>(L, o, d, ix, iy, iz, nox, noy, noz, eps, exact, a, pend, cur, sp, tbest, hit)) {
            }
            while (pend && !media_step<true>(L, o, d, mc, sc.n_media, pend, tbest, hit)) {
            }
        }
    } else {
        float4* s_geo = reinterpret_cast<float4*>(smem);
        if (sc.n_xforms || sc.n_media) {
            closest_hit_spheres_general(sc, o, d, tbest, hit);
        } else {
            for (uint32_t t0 = 0; t0 < sc.n_spheres; t0 += RT_SPHERE_TILE) {
                const uint32_t nn = min(RT_SPHERE_TILE, sc.n_spheres - t0);
                __syncthreads();
                for (uint32_t k = threadIdx.x; k < nn; k += BLOCK) s_geo[k] = sc.sph_geo[t0 + k];
                __syncthreads();
                closest_hit_tile(s_geo, nn, t0, o, d, a, tbest, hit);
            }
        }
        const MediumCtx mc{active ? in_key[2 * i] : 0u, active ? in_key[2 * i + 1] : 0u, depth_counter_base(depth)};
        closest_hit_rects(sc, o, d, mc, tbest, hit);
    }
    if (!active) return;
    uint32_t n_fetch = 0;
    Rng rng{in_key[2 * i], in_key[2 * i + 1], depth_counter_base(depth)};
    Bounce bo = shade<true, NoPrefetch, true>(sc, PerlinTables{sc.perlin_vec, sc.perlin_perm2}, o, d, hit, tbest, rng, n_fetch);
    out_hit[i] = hit;
    out_t[i] = hit >= 0 ? tbest : 0.0f;
    out_rad[3 * i] = bo.radiance.x, out_rad[3 * i + 1] = bo.radiance.y, out_rad[3 * i + 2] = bo.radiance.z;
    out_att[3 * i] = bo.attenuation.x, out_att[3 * i + 1] = bo.attenuation.y, out_att[3 * i + 2] = bo.attenuation.z;
    out_o[3 * i] = bo.o.x, out_o[3 * i + 1] = bo.o.y, out_o[3 * i + 2] = bo.o.z;
    out_d[3 * i] = bo.d.x, out_d[3 * i + 1] = bo.d.y, out_d[3 * i + 2] = bo.d.z;
    out_alive[i] = bo.alive ? 1 : 0;
}

} // namespace rt
