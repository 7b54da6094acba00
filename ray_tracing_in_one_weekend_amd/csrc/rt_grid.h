// rt_grid.h — closest hit of the queued (secondary) rays of a sphere-only scene by a 3D-DDA walk over a uniform grid
// that lives in LDS (k_intersect_grid), and the host code that builds the grid at rt_scene_upload.
//
// Why a second search structure.  k_intersect's BVH4 walk is bound by vector issue: ~100 vector instructions per node step,
// 7.3 node steps and 2.3 sphere tests per secondary ray of sphere_scene (round 3 showed that neither
// coherence nor any rearrangement of the node step moves it).  The scenes of this renderer's headline configurations are
// hundreds of similar small spheres on a large one (demo_scene.rs:37-86 sphere_scene; pbr_sweep_scene).  For those a
// uniform grid does the same search with ~20 vector instructions per cell step: on the secondary rays of sphere_scene
// 3.9 cell steps + 2.8 sphere tests in cells + 4 always-tested large spheres (scripts/whatif_grid_sim.py, the CPU what-if
// that preceded this file).  The result is the SAME: every sphere whose exact Sphere::hit root (hitable.rs:75-91) could win
// is tested with the same sphere_root and the same order-independent winner rule as the tree and the list walk
// (hitable.rs:117-132: smallest accepted root, ties to the later sphere), so hit records are bit-identical — except where
// fp32 Sphere::hit reports a root for a ray that misses the sphere in exact arithmetic: a cell or box test may cull such a false
// positive, the list walk cannot (at most 8 of the 1.35e9 rays of config 2; each proven in float64 by the tests: grid == tree ==
// list walk on adversarial rays and on whole frames).
//
// Structure.  Spheres whose padded box would cover more than RT_GRID_BIG_CELLS cells ("large": the r = 1000 ground of
// sphere_scene, its three r = 1 spheres) are tested for every ray when the lane takes the ray (all lanes of a refill
// together); the grid spans the union of the other spheres' boxes with nearly cubic cells (per axis the extent divided
// by a whole number: the 0.4-high layer of sphere_scene's small spheres is ONE layer of cells, not two).
// A cell record is (offset << 12 | count) into a u16 list of sphere ids; a sphere is listed in every cell its box,
// grown by `pad`, overlaps (and whose cell box it actually reaches: corner cells are pruned by distance).
// Testing a sphere that the ray cannot hit never changes the result, so nothing depends on the walk staying inside the
// grid to the last ulp: a step that overshoots the far face reads some other cell (the index is clamped to the array)
// and at worst tests a few spheres for nothing before the exit test ends the walk.
//
// Conservativeness.  The walk visits the cells an fp32 DDA believes the ray crosses; the exact ray may cross a
// neighbouring cell within the DDA's rounding distance of a visited one.  `pad` = 2^-7 of the shortest cell edge bounds that
// distance with margin: plane distances are (plane - o) * (1/d) with a 1-ulp v_rcp (relative 2^-22 of t), advanced by
// at most nx + ny + nz additions of the per-axis increment (each half an ulp of t), for t below the grid's diagonal
// of at most ~100 cells: at most 200 * 2^-24 * 100 cells = 2^-10 cells; rays whose origin coordinates exceed
// pad * 2^20 (where one ulp of the origin is pad / 8) do not walk the grid at all but test every sphere (the
// "all spheres" list behind the cell lists).  The walk ends when the best root so far lies before the plane through
// which the ray leaves the current cell, or when the ray leaves the grid; a step budget of nx + ny + nz cell steps
// (a DDA cannot make more) sends a ray that exceeds it to the all-spheres list as well, so every lane terminates.
#pragma once
#include "rt_kernels.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace rt {

#define RT_GRID_MAX_ALWAYS 4u      // large spheres tested for every ray
#define RT_GRID_CNT_BITS 12u       // cell record: offset << 12 | count
#define RT_GRID_CNT_MASK 0xFFFu
#define RT_GRID_MAX_SPHERES 4095u  // the all-spheres list is one cell record
#ifndef RT_GRID_BIG_CELLS
#define RT_GRID_BIG_CELLS 64.0     // a sphere whose box covers more cells than this is "large"
#endif
#ifndef RT_GRID_MAX_CELLS
#define RT_GRID_MAX_CELLS 16384u
#endif
#ifndef RT_GRID_CELL_FACTOR
#define RT_GRID_CELL_FACTOR 1.4    // cell edge = this * the median sphere diameter (measured: profiles/round4/grid_sweep*.txt)
#endif
#ifndef RT_GRID_REFILL_MIN
#define RT_GRID_REFILL_MIN RT_REFILL_MIN // idle lanes before the wave takes new rays
#endif

struct GridParams {
    float g0[3], b1[3];    // the grid's box: min corner of cell (0, 0, 0), max corner of the last cell
    float cs[3], inv_cs[3]; // cell edges
    uint32_t nx, ny, nz;   // cells per axis
    uint32_t n_cells, n_refs; // n_refs: u16 entries including the all-spheres list, padded to an even count
    uint32_t all_rec;      // cell record of the all-spheres list
    uint32_t n_always;
    uint32_t always[RT_GRID_MAX_ALWAYS];
    float max_coord;       // rays with an origin coordinate beyond this magnitude test every sphere instead of walking
    uint32_t n_spheres;
    const uint32_t* cells;       // [n_cells]
    const unsigned short* refs;  // [n_refs]
    const float4* sph_geo;       // [n_spheres]
};

__host__ __device__ inline size_t grid_lds_bytes(uint32_t n_spheres, uint32_t n_cells, uint32_t n_refs) {
    return (size_t)n_spheres * 16u + (size_t)n_cells * 4u + (((size_t)n_refs * 2u + 15u) & ~(size_t)15u) + 16u;
}

struct HostGrid {
    bool ok = false;
    GridParams gp{};
    std::vector<uint32_t> cells;
    std::vector<unsigned short> refs;
    double refs_per_sphere = 0.0;
};

// Builds the grid over `sph` (cx, cy, cz, r) or leaves out.ok false when the scene does not suit one: too few or too
// many spheres, more than RT_GRID_MAX_ALWAYS large ones at every cell size tried, a grid that does not fit `lds_budget`
// next to the sphere list, or one whose cells are so small against the coordinates that the fp32 walk could not be
// trusted (pad below 2^-20 of the largest coordinate).  cell_factor: cell edge in median sphere diameters, 0 = the default.
inline void build_sphere_grid(const std::vector<float4>& sph, size_t lds_budget, double cell_factor, HostGrid& out) {
    out = HostGrid{};
    const uint32_t n = (uint32_t)sph.size();
    if (n < 16u || n > RT_GRID_MAX_SPHERES) return;
    std::vector<double> diam(n);
    for (uint32_t i = 0; i < n; ++i) diam[i] = 2.0 * std::fabs((double)sph[i].w);
    std::vector<double> sorted(diam);
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const double median = sorted[n / 2];
    if (!(median > 0.0) || !std::isfinite(median)) return;
    double cs_d = (cell_factor > 0.0 ? cell_factor : RT_GRID_CELL_FACTOR) * median;
    for (int attempt = 0; attempt < 16; ++attempt, cs_d *= 1.25) {
        const double pad = cs_d / 128.0; // (the cells come out between 2/3 and 4/3 of cs_d; the bound below uses the real ones)
        std::vector<uint32_t> always, small;
        for (uint32_t i = 0; i < n; ++i) {
            const double k = std::floor((diam[i] + 2.0 * pad) / cs_d) + 2.0;
            (k * k * k > RT_GRID_BIG_CELLS ? always : small).push_back(i);
        }
        if (always.size() > RT_GRID_MAX_ALWAYS || small.size() < 8u) continue;
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, maxabs = 0.0;
        for (uint32_t i : small) {
            const double c[3] = {sph[i].x, sph[i].y, sph[i].z}, r = 0.5 * diam[i];
            for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], c[k] - r - 2.0 * pad), hi[k] = std::max(hi[k], c[k] + r + 2.0 * pad);
        }
        GridParams g{};
        uint64_t total = 1;
        uint32_t dims[3] = {1u, 1u, 1u};
        for (int k = 0; k < 3; ++k) {
            maxabs = std::max(maxabs, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
            const double cells_k = std::max(1.0, std::floor((hi[k] - lo[k]) / cs_d + 0.5)); // whole cells, each within [2/3, 4/3] of cs_d
            if (!(cells_k < 4096.0)) {
                total = ~0ull;
                break;
            }
            dims[k] = (uint32_t)cells_k;
            total *= dims[k];
            g.g0[k] = (float)lo[k];
            g.cs[k] = (float)((hi[k] - lo[k]) / cells_k);
            g.inv_cs[k] = 1.0f / g.cs[k];
            g.b1[k] = (float)((double)g.g0[k] + (double)g.cs[k] * cells_k);
        }
        if (total > RT_GRID_MAX_CELLS) continue;
        // pad must dominate the rounding of the walk: every plane distance is good to a few ulp of t or of a coordinate, and a
        // lane adds its per-axis increment at most nx + ny + nz times (header comment); 4x margin
        {
            double diag2 = 0.0;
            for (int k = 0; k < 3; ++k) diag2 += (hi[k] - lo[k]) * (hi[k] - lo[k]);
            const double steps = (double)dims[0] + dims[1] + dims[2] + 8.0;
            if (!(pad >= 4.0 * steps * (std::sqrt(diag2) + maxabs) / 8388608.0) || !(pad * 1048576.0 >= 2.0 * maxabs)) continue;
        }
        g.nx = dims[0], g.ny = dims[1], g.nz = dims[2];
        g.n_cells = (uint32_t)total;
        // cell lists: count, prefix, fill (cells in x-fastest order; spheres in index order within a cell)
        auto cell_range = [&](uint32_t i, int k, uint32_t& a, uint32_t& b) {
            const double c = k == 0 ? sph[i].x : (k == 1 ? sph[i].y : sph[i].z), r = 0.5 * diam[i] + pad;
            const double fa = std::floor((c - r - (double)g.g0[k]) / (double)g.cs[k]), fb = std::floor((c + r - (double)g.g0[k]) / (double)g.cs[k]);
            a = (uint32_t)std::min<double>(std::max(fa, 0.0), dims[k] - 1.0);
            b = (uint32_t)std::min<double>(std::max(fb, 0.0), dims[k] - 1.0);
        };
        auto reaches = [&](uint32_t i, uint32_t x, uint32_t y, uint32_t z) { // sphere grown by pad against the cell box
            const double c[3] = {sph[i].x, sph[i].y, sph[i].z}, r = 0.5 * diam[i] + pad;
            const uint32_t q[3] = {x, y, z};
            double d2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double mn = (double)g.g0[k] + (double)g.cs[k] * q[k], mx = mn + (double)g.cs[k];
                const double e = c[k] < mn ? mn - c[k] : (c[k] > mx ? c[k] - mx : 0.0);
                d2 += e * e;
            }
            return d2 <= r * r;
        };
        std::vector<uint32_t> count(g.n_cells, 0u);
        out.cells.clear();
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                uint64_t sum = 0;
                bool bad = false;
                out.cells.assign(g.n_cells, 0u);
                for (uint32_t c = 0; c < g.n_cells; ++c) {
                    if (count[c] > RT_GRID_CNT_MASK) bad = true;
                    out.cells[c] = (uint32_t)(sum << RT_GRID_CNT_BITS); // the count is added back while filling
                    sum += count[c];
                }
                if (bad || sum + n + 1u >= (1u << (32u - RT_GRID_CNT_BITS))) {
                    out.cells.clear();
                    break;
                }
                out.refs.assign((size_t)sum, 0);
            }
            for (uint32_t i : small) {
                uint32_t ax, bx, ay, by, az, bz;
                cell_range(i, 0, ax, bx), cell_range(i, 1, ay, by), cell_range(i, 2, az, bz);
                for (uint32_t z = az; z <= bz; ++z)
                    for (uint32_t y = ay; y <= by; ++y)
                        for (uint32_t x = ax; x <= bx; ++x) {
                            if (!reaches(i, x, y, z)) continue;
                            const uint32_t c = (z * g.ny + y) * g.nx + x;
                            if (pass == 0) {
                                ++count[c];
                            } else {
                                uint32_t& rec = out.cells[c];
                                out.refs[(rec >> RT_GRID_CNT_BITS) + (rec & RT_GRID_CNT_MASK)] = (unsigned short)i;
                                ++rec;
                            }
                        }
            }
        }
        if (out.cells.empty()) continue;
        out.refs_per_sphere = (double)out.refs.size() / (double)small.size();
        // Quality gate.  This is the finest grid the scene admits (coarser cells only list more spheres per cell): a ray pays
        // refs / cells sphere tests per cell it crosses.  Spheres of very unequal sizes or in clumps (a 4 x 4 x 4 grid with 16
        // references per cell for radii spread over four decades) are the tree's job.
        if (g.n_cells < 64u || (double)out.refs.size() > 4.0 * (double)g.n_cells || out.refs_per_sphere > 32.0) break;
        g.all_rec = ((uint32_t)out.refs.size() << RT_GRID_CNT_BITS) | n;
        for (uint32_t i = 0; i < n; ++i) out.refs.push_back((unsigned short)i);
        if (out.refs.size() & 1u) out.refs.push_back(0);
        g.n_refs = (uint32_t)out.refs.size();
        g.n_always = (uint32_t)always.size();
        for (uint32_t k = 0; k < g.n_always; ++k) g.always[k] = always[k];
        g.max_coord = (float)(pad * 1048576.0);
        g.n_spheres = n;
        if (grid_lds_bytes(n, g.n_cells, g.n_refs) > lds_budget) continue;
        out.gp = g;
        out.ok = true;
        return;
    }
    out = HostGrid{};
}

// Closest hit for every queued ray of the shards q = q0 + blockIdx.x, q0 + blockIdx.x + gridDim.x, ... below q1 (the
// shard ownership, the LDS work counter and the persistent lanes of k_intersect, rt_kernels.h), by the grid walk.
// Depth >= 1 only: depth 0 of these scenes is answered by the candidate lists inside k_shade<GEN>, and k_intersect<GEN>
// keeps the tree for the pixels whose list overflowed.
//
// A lane that holds a ray is either WALKING (cnt == 0: the cell it is in has no untested sphere left) or READY
// (cnt > 0).  One trip of the main loop lets every walking lane take one cell step and then every ready lane (those that
// just stepped into a non-empty cell included) test one sphere.  (Measured and not kept, profiles/round4/grid_sweep*.txt:
// up to 2 / 4 steps per trip through empty cells +4 / +8 %; holding the test trip back until 8 / 24 / 40 lanes are ready
// +1.5 / +4 / +7 %; refilling at 56 / 40 / 32 / 24 / 16 idle lanes +1 / 0 / +2 / +6 / +14 %.)
// The large spheres are tested when a lane takes its ray, all refilled lanes together.  (One loop over the candidates left
// by a cheap pre-pass — behind the origin, negative discriminant — so that the square root and the divisions run once per
// trip instead of once per large sphere: 3 % slower, the pre-pass repeats 20 instructions per sphere.)
// ONE_LAYER (ny == 1, the usual case: sphere_scene's 0.4-high layer of small spheres is 41 x 1 x 42 cells): the walk never steps in y —
// the plane through which the ray leaves its cell in y is the one through which it leaves the grid, and `texit` already ends the walk
// there — so the y terms of the DDA (six instructions of the set-up, five of every step, three registers) are left out.  Whether the
// general form took that last step (an ulp decides between its tmy and texit) or not, it only ever added redundant tests.
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool ONE_LAYER>
__global__ __launch_bounds__(RT_BVH_BLOCK, 8) void k_intersect_grid(GridParams G, const float4* __restrict__ qa,
                                                                      const float4* __restrict__ qb, float2* __restrict__ qh,
                                                                      const uint32_t* __restrict__ in_counts, IntersectParams ip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t BLOCK = RT_BVH_BLOCK;
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
    float4* s_geo = reinterpret_cast<float4*>(smem);
    uint32_t* s_cells = reinterpret_cast<uint32_t*>(s_geo + G.n_spheres);
    unsigned short* s_refs = reinterpret_cast<unsigned short*>(s_cells + G.n_cells);
    uint32_t* s_work = reinterpret_cast<uint32_t*>(smem + grid_lds_bytes(G.n_spheres, G.n_cells, G.n_refs) - 16u);
    for (uint32_t i = threadIdx.x; i < G.n_spheres; i += BLOCK) s_geo[i] = G.sph_geo[i];
    for (uint32_t i = threadIdx.x; i < G.n_cells; i += BLOCK) s_cells[i] = G.cells[i];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(G.refs);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_refs);
        for (uint32_t i = threadIdx.x; i < G.n_refs / 2u; i += BLOCK) dst[i] = src[i];
    }
    if (threadIdx.x == 0) *s_work = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const int stride_y = (int)G.nx, stride_z = (int)(G.nx * G.ny);
    const uint32_t last_cell = G.n_cells - 1u;
    const uint32_t all_cnt = G.all_rec & RT_GRID_CNT_MASK, all_off = G.all_rec >> RT_GRID_CNT_BITS; // the list of every sphere
    bool exhausted = false; // wave-uniform: the workgroup has no unclaimed rays left
    bool has = false;
    V3 o = splat(0.0f), d = v3(0.f, 0.f, 1.f);
    SharedRcp rcp_a{1.0f, 1.0f}; // |d|^2 (hitable.rs:77) and its refined reciprocal: the divisor of every root of this ray
    float tbest = RT_FLT_MAX, texit = 0.0f;
    float tmx = 0.f, tmy = 0.f, tmz = 0.f, tdx = 0.f, tdy = 0.f, tdz = 0.f;
    int hit = -1, cell = 0, sx = 1, sy = 1, sz = 1;
    uint32_t cnt = 0u, off = 0u, budget = 0u;
    size_t pos = 0;
    auto test_sphere = [&](uint32_t s) {
        float th;
        if (sphere_root(s_geo[s], o, d, rcp_a, 1e-3f, RT_FLT_MAX, th) && (th < tbest || (th == tbest && (int)s > hit))) {
            tbest = th;
            hit = (int)s;
        }
    };
    // (the shard the wave's claims are in: see the refill)
    uint32_t cur_k = 0u, cur_lo = 0u, cur_hi = pre[1];
    size_t cur_base = (size_t)(ip.q0 + blockIdx.x) * ip.cap;
    auto q_finish = [&]() {
        qh[pos] = make_float2(tbest, __int_as_float(hit));
        has = false;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        RT_LANE_STAT(0, has);
        if (n_idle >= RT_GRID_REFILL_MIN && !exhausted) {
            RT_LANE_STAT(6, !has);
            uint32_t v0 = 0;
            if (lane == 0) v0 = atomicAdd(s_work, n_idle); // LDS atomic: claim n_idle rays
            v0 = __builtin_amdgcn_readfirstlane(v0);
            if (v0 + n_idle >= total) exhausted = true;
            const uint32_t v =
                v0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            // The shard of a claimed ray.  A workgroup's claims only move forward through its virtual index space, and the rays of one
            // claim are consecutive: the wave keeps the shard its last claim started in (`cur_k`, the part [cur_lo, cur_hi) of the index
            // space, the queue position `cur_base` of index 0 of that part — all scalar), advances it when a claim starts beyond it, and
            // every lane of a claim that ends inside it gets its position by ONE 64-bit addition.  Only a claim that straddles a shard
            // boundary (at most three per workgroup and wave) searches per lane — which every refill used to do: 20 of its ~175
            // vector instructions.
            while (v0 >= cur_hi && cur_k + 1u < n_my) { // (wave-uniform)
                ++cur_k;
                cur_lo = cur_hi;
                uint32_t nh = total;
#pragma unroll
                for (uint32_t t = 1; t < RT_ISECT_MAX_SHARDS; ++t) nh = (cur_k + 1u == t) ? pre[t] : nh;
                cur_hi = nh;
                cur_base = (size_t)(ip.q0 + blockIdx.x + cur_k * gridDim.x) * ip.cap - cur_lo;
            }
            if (!has && v < total) {
                if (min(v0 + n_idle, total) <= cur_hi) { // (wave-uniform) the whole claim lies in the current shard
                    pos = cur_base + v;
#ifdef RT_DEBUG_QUEUE_BOUNDS
                    if (v - cur_lo >= ip.cap || ip.q0 + blockIdx.x + cur_k * gridDim.x >= ip.q1) __builtin_trap();
#endif
                } else {
                    uint32_t k = 0;
#pragma unroll
                    for (uint32_t t = 1; t < RT_ISECT_MAX_SHARDS; ++t) k += (t < n_my && v >= pre[t]) ? 1u : 0u;
                    uint32_t qoff = v;
#pragma unroll
                    for (uint32_t t = 1; t < RT_ISECT_MAX_SHARDS; ++t) qoff = (k == t) ? v - pre[t] : qoff;
                    const uint32_t shard = ip.q0 + blockIdx.x + k * gridDim.x;
                    pos = (size_t)shard * ip.cap + qoff;
#ifdef RT_DEBUG_QUEUE_BOUNDS
                    if (qoff >= ip.cap || shard >= ip.q1) __builtin_trap();
#endif
                }
                // (with the two halves of a ray's record interleaved, b = a + 1: one address)
                const float4 ra = qa[RT_QSTRIDE * pos], rb = RT_QSTRIDE == 2u ? qa[RT_QSTRIDE * pos + 1u] : qb[RT_QSTRIDE * pos];
                o = v3(ra.x, ra.y, ra.z);
                d = v3(rb.x, rb.y, rb.z);
                rcp_a = shared_rcp(length_squared(d));
                tbest = RT_FLT_MAX;
                hit = -1;
                has = true;
#pragma unroll
                for (uint32_t k2 = 0; k2 < RT_GRID_MAX_ALWAYS; ++k2)
                    if (k2 < G.n_always) test_sphere(G.always[k2]);
                // the ray against the grid's box (culling arithmetic: v_rcp, fused multiply-adds)
                const float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
                const float x0 = (G.g0[0] - o.x) * ix, x1 = (G.b1[0] - o.x) * ix;
                const float y0 = (G.g0[1] - o.y) * iy, y1 = (G.b1[1] - o.y) * iy;
                const float z0 = (G.g0[2] - o.z) * iz, z1 = (G.b1[2] - o.z) * iz;
                const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
                const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
                const float mo = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
                cnt = 0u;
                if (__builtin_expect(!(mo <= G.max_coord), 0)) { // far away (or NaN): every sphere, here and now, no walk (a cold block
                    // of its own, like the step budget below: as a state of the walk it cost every refill six constants and eight copies)
                    for (uint32_t k = 0; k < all_cnt; ++k) test_sphere(s_refs[all_off + k]);
                    q_finish();
                } else if (!(tn <= tf) || !(tn < tbest)) { // misses the grid, or reaches it behind the best large sphere
                    qh[pos] = make_float2(tbest, __int_as_float(hit));
                    has = false;
                } else {
                    const float px = __builtin_fmaf(d.x, tn, o.x), py = __builtin_fmaf(d.y, tn, o.y), pz = __builtin_fmaf(d.z, tn, o.z);
                    const float fx = min_raw(fmaxf(floorf((px - G.g0[0]) * G.inv_cs[0]), 0.0f), (float)(G.nx - 1u));
                    const float fy = ONE_LAYER ? 0.0f : min_raw(fmaxf(floorf((py - G.g0[1]) * G.inv_cs[1]), 0.0f), (float)(G.ny - 1u));
                    const float fz = min_raw(fmaxf(floorf((pz - G.g0[2]) * G.inv_cs[2]), 0.0f), (float)(G.nz - 1u));
                    cell = ONE_LAYER ? (int)fz * (int)G.nx + (int)fx : ((int)fz * (int)G.ny + (int)fy) * (int)G.nx + (int)fx;
                    // a component too small to ever reach the next plane (1/d infinite, NaN or beyond 1e30): never stepped
                    const bool wx = fabsf(ix) < 1e30f, wy = fabsf(iy) < 1e30f, wz = fabsf(iz) < 1e30f;
                    const float bx = __builtin_fmaf(fx + (d.x >= 0.0f ? 1.0f : 0.0f), G.cs[0], G.g0[0]);
                    const float by = __builtin_fmaf(fy + (d.y >= 0.0f ? 1.0f : 0.0f), G.cs[1], G.g0[1]);
                    const float bz = __builtin_fmaf(fz + (d.z >= 0.0f ? 1.0f : 0.0f), G.cs[2], G.g0[2]);
                    tmx = wx ? (bx - o.x) * ix : INFINITY, tdx = wx ? G.cs[0] * fabsf(ix) : INFINITY;
                    if (!ONE_LAYER) tmy = wy ? (by - o.y) * iy : INFINITY, tdy = wy ? G.cs[1] * fabsf(iy) : INFINITY;
                    tmz = wz ? (bz - o.z) * iz : INFINITY, tdz = wz ? G.cs[2] * fabsf(iz) : INFINITY;
                    sx = d.x >= 0.0f ? 1 : -1, sy = d.y >= 0.0f ? stride_y : -stride_y, sz = d.z >= 0.0f ? stride_z : -stride_z;
                    texit = tf;
                    budget = G.nx + G.ny + G.nz;
                    const uint32_t rec = s_cells[cell];
                    cnt = rec & RT_GRID_CNT_MASK, off = rec >> RT_GRID_CNT_BITS;
                }
            }
        }
        if (!__any(has)) {
            if (exhausted) break;
            continue;
        }
        // ---- walk: a lane whose cell is used up steps to the next cell, or finishes
        RT_LANE_STAT(2, has && cnt == 0u);
        if (has && cnt == 0u) {
            // (v_min_f32 itself: fminf() of two loop-carried values costs a canonicalising v_max x, x per operand in front of it, four
            // of the step's ~25 instructions; none of these is ever a NaN — distances, INFINITY, FLT_MAX)
            const float tnext = ONE_LAYER ? min_raw(tmx, tmz) : min_raw(min_raw(tmx, tmy), tmz);
            if (!(tnext < min_raw(tbest, texit))) { // the best root lies inside the cells visited, or the ray has left the grid
                qh[pos] = make_float2(tbest, __int_as_float(hit));
                has = false;
            } else {
                const bool ax = tmx <= tnext, ay = !ONE_LAYER && !ax && tmy <= tnext, az = !ax && !ay;
                cell += ax ? sx : (ay ? sy : sz);
                tmx = ax ? tmx + tdx : tmx;
                if (!ONE_LAYER) tmy = ay ? tmy + tdy : tmy;
                tmz = az ? tmz + tdz : tmz;
                const uint32_t rec = s_cells[min((uint32_t)cell, last_cell)];
                cnt = rec & RT_GRID_CNT_MASK, off = rec >> RT_GRID_CNT_BITS;
                // cannot happen for a DDA; keeps the loop finite whatever the arithmetic did: every sphere, here and now.  (As a state
                // of the walk — the list of all spheres as the lane's cell, infinite plane distances — its five constants were set up
                // on every trip of every lane.)
                if (__builtin_expect(--budget == 0u, 0)) {
                    for (uint32_t k = 0; k < all_cnt; ++k) test_sphere(s_refs[all_off + k]);
                    q_finish();
                    cnt = 0u;
                }
            }
        }
        // ---- test: one sphere per lane that has one
        RT_LANE_STAT(4, has && cnt > 0u);
        if (has && cnt > 0u) {
            const uint32_t s = s_refs[off];
            ++off, --cnt;
            test_sphere(s);
        }
    }
}

} // namespace rt
