// rt_api.hip — implementation of the C-ABI in include/rtow_mi355x.h on top of the gfx950
// wavefront kernels (rt_kernels.h).  One context = one GPU = one HIP stream; the bounce loop
// of a slice is enqueued without any host synchronisation (queue sizes live in HBM).
#include "../../include/rtow_mi355x_debug.h"
#include "rt_kernels.h"
#include "rt_bvh.h"
#include "rt_grid.h"
#include "rt_pool.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace rt;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool arena = false; // carved from the context's pool (rt_pool.h) instead of hipMalloc'ed: never freed on its own
};

} // namespace

struct RtCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // second shard group of a slice (render_impl)
    std::vector<hipStream_t> parked_streams;  // streams that turned out to share a hardware queue with the first chain's (ensure_concurrent_chains)
    hipStream_t paired_with = nullptr;        // the launch stream stream2 was last checked against
    bool chains_concurrent = true;            // what that check found (false: no stream of this process ran beside the launch stream)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    // scene
    bool has_scene = false;
    DevScene ds{};
    // Every device buffer of a context lies in its pool (rt_pool.h): the small persistent ones (scene, accumulator, counters,
    // candidate lists, output images) are carved from the bottom by a bump pointer, the per-frame work buffers start above
    // them.  Only the pool's helper thread ever asks the driver for device memory, so a request that the driver makes wait
    // (seconds, now and then: scripts/micro/alloc_probe stallprobe) holds up the growth of the pool and nothing else —
    // kernel launches, copies from pinned memory and events of the calling thread are not affected by it, its own
    // hipMalloc would be (measured: 5.4 s).  A buffer that is outgrown is left behind (contexts hold a handful; the scene has a
    // region of its own that re-uploads reuse); hipMalloc remains the fallback when the pool cannot serve.
    size_t arena_top = 0;
    DevBuf scene_region;       // the uploaded scene's arrays, bump-allocated (scene_used) and reused by the next upload
    size_t scene_used = 0;
    bool scene_measuring = false; // first pass of rt_scene_upload: upload() only adds up what it would need
    size_t scene_measure = 0;
    char* h_stage = nullptr;   // page-locked staging for host -> device copies (a pageable source costs ~30 ms on first use)
    // work buffers (grown on demand, reused across calls)
    WorkPool pool;    // two ray queues, hit records, radiance slots: one virtual range grown by a helper thread (rt_pool.h)
    DevBuf acc, counts, totals, out_f32, out_u8, dbg, genp, lists;
    std::vector<hipEvent_t> events;
    std::vector<hipEvent_t> depth_events;  // RT_FLAG_TIME_DEPTHS: 3 per depth (k_intersect begin, k_shade begin, k_shade end)
    int timed_depths = 0;
    std::vector<unsigned long long> timed_rays;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    int n_cu = 256;
    size_t lds_limit = 64 * 1024;
    bool use_bvh = false;      // scene BVH fits LDS next to the traversal stacks
    size_t isect_lds = 0;      // k_intersect: nodes + geometry (when they fit) + stack levels + counters
    bool bvh_in_lds = false;   // false: the tree is traversed out of HBM/L2, only the stacks are in LDS
    bool general_lds = false;  // k_intersect<.., GLDS>: the wrapper / medium tables of a general scene are staged in LDS
    bool nest = false;         // k_intersect<.., NEST>: a wrapper chain deeper than RT_MAX_CHAIN, more than 32 media or a wrapper around a medium (rt_device.h)
    bool general_kernels = false; // the general instantiations (rectangles, wrappers, media — or RT_OPT_GENERAL_KERNELS at upload)
    bool use_grid = false;     // sphere-only scene with a uniform grid (rt_grid.h): depth >= 1 runs k_intersect_grid
    GridParams grid{};
    size_t grid_lds = 0;
    uint32_t opt[RT_OPT__COUNT] = {}; // rt_debug_set_option: per context, every setting renders the same bits
    // Page-locked word for the one host decision inside a frame: how many pixels have more primary-ray candidates than a list
    // holds (k_primary_lists counts them; render_impl reads the count back 0.1 ms into the frame).
    uint32_t* h_overflow = nullptr;
    // progressive preview (rt_set_progress): called from rt_render after every slice
    RtProgressFn progress_fn = nullptr;
    void* progress_user = nullptr;
    bool progress_armed = false; // set by rt_render around render_impl, so rt_render_device stays callback-free
    DevBuf preview_u8;
    std::vector<uint8_t> preview_host;
    // Host-side timeline of the last render_impl (rt_debug_render_parts): wall-clock milliseconds between marks.  The first
    // render of a process is where allocations, code-object loads and the queue probe happen; this says which.
    std::vector<std::pair<const char*, double>> parts;
    std::chrono::steady_clock::time_point parts_t{};
    void parts_begin() { parts.clear(), parts_t = std::chrono::steady_clock::now(); }
    void mark(const char* what) {
        const auto t = std::chrono::steady_clock::now();
        parts.emplace_back(what, std::chrono::duration<double, std::milli>(t - parts_t).count());
        parts_t = t;
    }
};

namespace {

int fail(RtCtx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define RT_HIP(ctx, expr)                                                                                  \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) {                                                                           \
            return fail(ctx, e__ == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE,                    \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                               \
        }                                                                                                  \
    } while (0)

constexpr size_t RT_STAGE_BYTES = 4u << 20;
constexpr size_t RT_ARENA_MAX = 8ull << 30; // small buffers beyond this (frames of hundreds of megapixels) are hipMalloc'ed

// `bytes` of the pool behind the buffers carved so far, or nullptr when the pool cannot give them (reservation spent, or the
// device has no memory left for the chunks: the caller falls back to hipMalloc and its error text)
void* arena_alloc(RtCtx* ctx, size_t bytes) {
    const size_t off = (ctx->arena_top + 255u) & ~(size_t)255u;
    const size_t need = off + bytes;
    if (need > RT_ARENA_MAX) return nullptr;
    if (pool_request(ctx->pool, need)) return nullptr;
    if (!pool_wait(ctx->pool, need)) return nullptr;
    ctx->arena_top = need;
    return ctx->pool.base + off;
}

// plain = true: hipMalloc'ed whatever the pool could do (buffers that other devices or libraries address: rt_multi.h)
int ensure(RtCtx* ctx, DevBuf& b, size_t bytes, bool plain = false) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return RT_OK;
    if (b.p && !b.arena) RT_HIP(ctx, hipFree(b.p));
    b = DevBuf{};
    if (!plain)
        if (void* p = arena_alloc(ctx, bytes)) {
            b = DevBuf{p, bytes, true};
            return RT_OK;
        }
    RT_HIP(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return RT_OK;
}

void free_buf(DevBuf& b) {
    if (b.p && !b.arena) (void)hipFree(b.p);
    b = DevBuf{};
}

void free_scene(RtCtx* ctx) { // (the region stays: the next upload carves it again)
    ctx->scene_used = 0;
    ctx->has_scene = false;
    std::memset(&ctx->ds, 0, sizeof(ctx->ds));
}

// host -> device through the page-locked staging buffer, on the context's stream, complete on return
int copy_to_device(RtCtx* ctx, void* dst, const void* src, size_t bytes) {
    for (size_t off = 0; off < bytes; off += RT_STAGE_BYTES) {
        const size_t n = std::min(RT_STAGE_BYTES, bytes - off);
        std::memcpy(ctx->h_stage, (const char*)src + off, n);
        RT_HIP(ctx, hipMemcpyAsync((char*)dst + off, ctx->h_stage, n, hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RT_OK;
}

template <class T>
int upload(RtCtx* ctx, const std::vector<T>& host, const T** dev) {
    const size_t bytes = (std::max<size_t>(host.size() * sizeof(T), 16) + 255u) & ~(size_t)255u;
    *dev = nullptr;
    if (ctx->scene_measuring) {
        ctx->scene_measure += bytes;
        return RT_OK;
    }
    if (ctx->scene_used + bytes > ctx->scene_region.bytes) return fail(ctx, RT_ERR_NOMEM, "rt_scene_upload: scene region too small (internal)");
    void* p = (char*)ctx->scene_region.p + ctx->scene_used;
    ctx->scene_used += bytes;
    if (!host.empty()) {
        const int rc = copy_to_device(ctx, p, host.data(), host.size() * sizeof(T));
        if (rc) return rc;
    }
    *dev = reinterpret_cast<const T*>(p);
    return RT_OK;
}

bool mat_needs_tex0(uint32_t t) {
    return t == RT_MAT_EMISSION || t == RT_MAT_DIFFUSE || t == RT_MAT_LAMBERT || t == RT_MAT_ISOTROPIC ||
           t == RT_MAT_OREN_NAYAR || t == RT_MAT_BURLEY_DIFFUSE || t == RT_MAT_ROUGH_PLASTIC ||
           t == RT_MAT_DISNEY_DIFFUSE || t == RT_MAT_DISNEY_METAL || t == RT_MAT_DISNEY_SHEEN;
}

// The cheap shading classes (1 + material*4 + texture of tex0; 0 = miss): no hit under the gradient / black sky, and
// the material.rs materials over a ConstantTex (Metal and Dielectric have no texture).
bool class_is_light(uint32_t cls, uint32_t sky_type) {
    if (cls == 0) return sky_type != RT_SKY_ENV;
    const uint32_t ty = (cls - 1u) / 4u, tt = (cls - 1u) % 4u;
    return tt == RT_TEX_CONSTANT && ty <= RT_MAT_ISOTROPIC;
}


// ---- two chains need two hardware queues ----------------------------------------------------------------------
// HIP multiplexes the streams of a process onto a small pool of hardware queues (4 by default), and two streams that share one
// run their kernels one after the other: the two chains of a slice then gain nothing from each other (config 2: 54.7 ms instead
// of 48.5; which context of a process draws the shared queue depends on how many streams exist — the "second-listed context
// reads 12 % high" of round 4's A/B runs, profiles/round5/ab_same_binary_four_contexts.txt).  Before the first frame on a
// launch stream, two single-wave spin kernels tell whether `stream2` runs beside it; if not, stream2 is parked (kept alive, so
// the slot stays taken) and a fresh stream tried, a few times.  Speed only: results never depend on it.
__global__ void k_spin(unsigned long long ticks, unsigned long long* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    for (uint32_t n = 0; n < (1u << 20) && t - t0 < ticks; ++n) { // bounded: every wave leaves
        __builtin_amdgcn_s_sleep(16);
        t = __builtin_amdgcn_s_memtime();
    }
    if (sink) *sink = t - t0;
}
int ensure_concurrent_chains(RtCtx* ctx, hipStream_t st) {
    if (ctx->paired_with == st) return RT_OK;
    const unsigned long long ticks = 150000ull; // ~60 us
    hipEvent_t e0 = nullptr, e1 = nullptr; // (its own pair: ev_begin / ev_end bracket the frame this runs inside)
    RT_HIP(ctx, hipEventCreate(&e0));
    RT_HIP(ctx, hipEventCreate(&e1));
    auto timed = [&](bool both, float& ms) -> int {
        RT_HIP(ctx, hipEventRecord(e0, st));
        if (both) {
            RT_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
            RT_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, ctx->stream2, ticks, (unsigned long long*)nullptr);
        }
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, ticks, (unsigned long long*)nullptr);
        if (both) {
            RT_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
        }
        RT_HIP(ctx, hipEventRecord(e1, st));
        RT_HIP(ctx, hipStreamSynchronize(st));
        RT_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        return RT_OK;
    };
    int rc;
    float one = 0.f, both = 0.f;
    if (!(rc = timed(false, one)) && !(rc = timed(false, one))) { // (the first launch of a kernel pays its load)
        for (int attempt = 0;; ++attempt) {
            if ((rc = timed(true, both))) break;
            ctx->chains_concurrent = both < 1.6f * one;
            if (ctx->chains_concurrent || attempt == 6) break;
            hipStream_t fresh = nullptr;
            if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            ctx->parked_streams.push_back(ctx->stream2);
            ctx->stream2 = fresh;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!rc) ctx->paired_with = st;
    return rc;
}

// ---- the two trace-step launches (render_impl and the production-kernel test hook share them) ----------------
struct StepBuffers {
    Queue qi, qo;
    float2* qhit;
    const uint32_t* cin;
    uint32_t* cout;
    float* rad;
    unsigned long long* totals;
    const GenParams* gpd;
};
bool scene_is_general(const RtCtx* ctx) { return ctx->general_kernels; }
bool grid_enabled(const RtCtx* ctx) { return ctx->use_grid && ctx->opt[RT_OPT_GRID] != 1u && !scene_is_general(ctx); }
bool scene_perlin_lds(const RtCtx* ctx) { return ctx->ds.n_perlin > 0 && ctx->ds.n_perlin <= RT_PERLIN_LDS_MAX_SETS; }

// closest hit of the shards [ip.q0, ip.q1): the tree instantiation that matches the scene, or the list walk
void launch_intersect(RtCtx* ctx, hipStream_t sg, bool use_bvh, bool gen, uint32_t grid, const StepBuffers& b, const IntersectParams& ip) {
    const bool rects = scene_is_general(ctx);
    if (use_bvh && !gen && grid_enabled(ctx)) { // sphere-only scene, depth >= 1: the grid walk (rt_grid.h), same hit records
        if (ctx->grid.ny == 1u) hipLaunchKernelGGL(k_intersect_grid<true>, dim3(grid), dim3(RT_BVH_BLOCK), ctx->grid_lds, sg, ctx->grid, b.qi.a, b.qi.b, b.qhit, b.cin, ip);
        else hipLaunchKernelGGL(k_intersect_grid<false>, dim3(grid), dim3(RT_BVH_BLOCK), ctx->grid_lds, sg, ctx->grid, b.qi.a, b.qi.b, b.qhit, b.cin, ip);
        return;
    }
#define RT_LAUNCH_ISECT_X(G, R, N, T, X)                                                                               \
    hipLaunchKernelGGL((k_intersect<RT_BVH_BLOCK, G, R, N, T, X>), dim3(grid), dim3(RT_BVH_BLOCK), ctx->isect_lds, sg, ctx->ds, \
                       b.qi.a, b.qi.b, b.qhit, b.cin, ip, b.gpd)
#define RT_LAUNCH_ISECT(G, R, N, T) RT_LAUNCH_ISECT_X(G, R, N, T, false)
    // general scenes: tables in LDS or not; and (ctx->nest) the instantiation whose wrapper chains and media masks are loops
#define RT_LAUNCH_ISECT_G(G, N)                       \
    do {                                              \
        if (ctx->nest) {                              \
            if (ctx->general_lds) RT_LAUNCH_ISECT_X(G, true, N, true, true); \
            else RT_LAUNCH_ISECT_X(G, true, N, false, true);      \
        } else if (ctx->general_lds) RT_LAUNCH_ISECT(G, true, N, true); \
        else RT_LAUNCH_ISECT(G, true, N, false);      \
    } while (0)
    // trees that do not fit LDS use the general instantiation (R = true works for sphere-only scenes too)
    if (use_bvh && !ctx->bvh_in_lds && gen) RT_LAUNCH_ISECT_G(true, false);
    else if (use_bvh && !ctx->bvh_in_lds) RT_LAUNCH_ISECT_G(false, false);
    else if (use_bvh && gen && rects) RT_LAUNCH_ISECT_G(true, true);
    else if (use_bvh && gen) RT_LAUNCH_ISECT(true, false, true, false);
    else if (use_bvh && rects) RT_LAUNCH_ISECT_G(false, true);
    else if (use_bvh) RT_LAUNCH_ISECT(false, false, true, false);
#undef RT_LAUNCH_ISECT_G
#undef RT_LAUNCH_ISECT
#undef RT_LAUNCH_ISECT_X
    else {
        const size_t list_lds = (size_t)std::min<uint32_t>(std::max<uint32_t>(ctx->ds.n_spheres, 1u), RT_SPHERE_TILE) * sizeof(float4);
        hipLaunchKernelGGL(k_intersect_list, dim3(ip.q1 - ip.q0), dim3(256), list_lds, sg, ctx->ds, b.qi.a, b.qi.b, b.qhit, b.cin, ip, b.gpd);
    }
}

// shading of the shards [sp.q0, sp.q0 + n_shards); `fused_lists`: depth 0 of a sphere-only scene with candidate lists
void launch_shade(RtCtx* ctx, hipStream_t sg, bool gen, bool fused_lists, uint32_t n_shards, const StepBuffers& b, const ShadeParams& sp) {
    const bool rects = scene_is_general(ctx), perlin_lds = scene_perlin_lds(ctx);
    // sphere geometry for the closest hit inside k_shade<GEN>
    const uint32_t n_fused = (gen && fused_lists) ? ctx->ds.n_spheres : 0u;
    const size_t shade_lds = shade_lds_bytes(ctx->ds.n_prims + ctx->ds.n_media, perlin_lds ? ctx->ds.n_perlin : 0u, n_fused, !gen && sp.sort);
#define RT_LAUNCH_SHADE(P, G, R, X) \
    hipLaunchKernelGGL((k_shade<P, G, R, X>), dim3(n_shards), dim3(256), shade_lds, sg, ctx->ds, b.qi, b.qhit, b.qo, b.cin, b.cout, b.rad, sp, b.totals, b.gpd)
#define RT_LAUNCH_SHADE_R(P, G)        \
    do {                               \
        if (rects && ctx->nest) RT_LAUNCH_SHADE(P, G, true, true); \
        else if (rects) RT_LAUNCH_SHADE(P, G, true, false); \
        else RT_LAUNCH_SHADE(P, G, false, false);      \
    } while (0)
    if (perlin_lds && gen) RT_LAUNCH_SHADE_R(true, true);
    else if (perlin_lds) RT_LAUNCH_SHADE_R(true, false);
    else if (gen) RT_LAUNCH_SHADE_R(false, true);
    else RT_LAUNCH_SHADE_R(false, false);
#undef RT_LAUNCH_SHADE_R
#undef RT_LAUNCH_SHADE
}

// Where the work buffers of a slice lie in the pool: two ray queues of n_queue rays (a / b records interleaved when RT_QSTRIDE is
// 2, a separate b array in the RT_QSTRIDE 1 build; the c array), n_queue hit records, n_paths radiance slots.
struct WorkLayout {
    size_t q_ab[2], q_b[2], q_c[2], qhit, rad, total;
};
WorkLayout work_layout(size_t n_queue, size_t n_paths) {
    WorkLayout w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255u) & ~(size_t)255u;
        return o;
    };
    for (int k = 0; k < 2; ++k) {
        w.q_ab[k] = take(RT_QSTRIDE * n_queue * sizeof(float4));
        w.q_b[k] = RT_QSTRIDE == 1u ? take(n_queue * sizeof(float4)) : w.q_ab[k] + sizeof(float4);
        w.q_c[k] = take(n_queue * sizeof(float2));
    }
    w.qhit = take(n_queue * sizeof(float2));
    w.rad = take(n_paths * RT_RAD_FLOATS * sizeof(float));
    w.total = off;
    return w;
}
struct WorkView {
    Queue Q[2];
    float2* qhit;
    float* rad;
};
WorkView work_view(const RtCtx* ctx, const WorkLayout& w, size_t base) {
    char* b = ctx->pool.base + base;
    WorkView v;
    for (int k = 0; k < 2; ++k) v.Q[k] = Queue{(float4*)(b + w.q_ab[k]), (float4*)(b + w.q_b[k]), (float2*)(b + w.q_c[k])};
    v.qhit = (float2*)(b + w.qhit), v.rad = (float*)(b + w.rad);
    return v;
}

// Queue geometry of a slice of n_max rays: shard count, k_intersect grid, shard capacity.
struct QueueGeom {
    uint32_t nq, isect_grid, cap;
};
QueueGeom queue_geom(const RtCtx* ctx, uint32_t n_max) {
    QueueGeom g;
    // Queue shards: one per k_shade workgroup (8 workgroups of 256 threads per CU), each owned by one
    // workgroup per kernel so that queue positions come from LDS counters (rt_kernels.h).
    g.nq = (uint32_t)ctx->n_cu * 8u;
    if (ctx->opt[RT_OPT_QUEUE_SHARDS]) g.nq = ctx->opt[RT_OPT_QUEUE_SHARDS];
    // k_intersect: as many 1024-thread workgroups per CU as LDS admits (two at <= 64 VGPRs); every
    // workgroup owns nq / isect_grid shards
    const size_t isect_lds = grid_enabled(ctx) ? std::max(ctx->isect_lds, ctx->grid_lds) : ctx->isect_lds;
    const uint32_t isect_wg_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2, ctx->lds_limit / std::max<size_t>(isect_lds, 1)));
    // ... in TWO rounds when the tree is more than a handful of nodes: a workgroup's persistent lanes end in a drain phase (its
    // work counter is empty, the last rays finish in ever emptier waves), and with exactly one resident round all workgroups
    // drain together; with twice as many, half as long, the drain of the first round runs under the bulk of the second.
    // config 2: 61.9 -> 59.8 ms (-3.4 %), pbr_sweep_scene -3.6 %; scenes of a few primitives (simple_light_scene +5.6 %) and
    // trees read through L2 (final_scene +1.9 %) do better with one round (profiles/round3/nq_sweep*.txt; non-multiples of
    // the resident count lose outright: 1 280 / 1 792 workgroups 60.6 / 62.0 ms).
    const uint32_t rounds = (ctx->bvh_in_lds && ctx->ds.n_bvh4_nodes >= 64u) ? 2u : 1u;
    g.isect_grid = std::min(g.nq, (uint32_t)ctx->n_cu * isect_wg_per_cu * rounds);
    if (ctx->opt[RT_OPT_ISECT_WORKGROUPS]) g.isect_grid = std::min(g.nq, ctx->opt[RT_OPT_ISECT_WORKGROUPS]);
    while ((g.nq + g.isect_grid - 1) / g.isect_grid > RT_ISECT_MAX_SHARDS) g.isect_grid *= 2;
    g.isect_grid = std::min(g.isect_grid, g.nq);
    const uint32_t nchunks = (n_max + 255u) / 256u;
    g.cap = ((nchunks + g.nq - 1) / g.nq) * 256u;
    return g;
}

// ---- slice sizing (rt_prepare and render_impl) ----------------------------------------------------------------
// A ray of a slice costs 100 B of work buffers (two 40 B queues, 8 B hit record, 12 B radiance slot).  Few, large slices amortise
// the short-queue tail of the bounce loop (depths > ~12 hold a few thousand rays: config 2 measured 99 / 88 / 82 / 79.5 ms per
// frame with 8 / 4 / 2 / 1 slices), so the library's own choice is up to 1 280 Mi rays (134 GB of the 288 GB HBM) and never more
// than half of what the device has free.
#ifndef RT_MAX_SLICE_MI_RAYS
#define RT_MAX_SLICE_MI_RAYS 1280 // Mi rays of the largest slice the library chooses by itself (134 GB of work buffers).  640 until round 6:
                                  // config 3 (4K, 1024 spp) in 7 slices instead of 13 is 1.6 % faster, config 5 1 % (the tail of a slice costs 3.7 ms)
#endif
uint32_t queue_shards(const RtCtx* ctx) { return ctx->opt[RT_OPT_QUEUE_SHARDS] ? ctx->opt[RT_OPT_QUEUE_SHARDS] : (uint32_t)ctx->n_cu * 8u; }
size_t slice_bytes(const RtCtx* ctx, uint32_t npix, uint32_t sc) {
    const uint32_t nq = queue_shards(ctx), n_max = npix * sc;
    const uint32_t cap = (((n_max + 255u) / 256u + nq - 1u) / nq) * 256u; // (queue_geom's shard capacity)
    return work_layout((size_t)nq * cap, n_max).total;
}
// the most samples per pixel (<= sc_max) whose slice fits `bytes`; 0 when not even one does
uint32_t slice_fit(const RtCtx* ctx, uint32_t npix, size_t bytes, uint32_t sc_max) {
    uint32_t sc = (uint32_t)std::min<uint64_t>(sc_max, bytes / (100ull * npix) + 1u);
    while (sc > 0u && slice_bytes(ctx, npix, sc) > bytes) --sc;
    return sc;
}
struct SlicePlan {
    uint32_t S_cap;    // most samples per pixel a slice may hold
    bool by_library;   // RtParams.spp_slice == 0: the library chose, and may start a frame with smaller slices while the pool grows
    size_t want_bytes; // pool size that holds such a slice
};
// (the work buffers of a slice start at work_base(ctx): behind the small persistent buffers carved from the pool so far)
size_t work_base(const RtCtx* ctx) { return (ctx->arena_top + 4095u) & ~(size_t)4095u; }
int plan_slices(RtCtx* ctx, const RtParams* prm, uint64_t npix64, SlicePlan& pl) {
    const uint32_t spp = prm->spp;
    uint32_t S = prm->spp_slice;
    pl.by_library = S == 0u;
    if (pl.by_library) S = (uint32_t)std::max<uint64_t>(1, ((uint64_t)RT_MAX_SLICE_MI_RAYS << 20) / npix64);
    S = std::min(std::min(S, spp), 1u << 20); // udiv_inv (slot -> sample, pixel) wants quotients below 2^21
    if ((uint64_t)S * npix64 > 0xFFFFFF00ull) S = (uint32_t)(0xFFFFFF00ull / npix64);
    if (S == 0) return fail(ctx, RT_ERR_UNSUPPORTED, "render: shard has more than 2^32 pixels");
    const uint32_t npix = (uint32_t)npix64;
    if (pl.by_library) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t half = (free_b + ctx->pool.mapped.load()) / 2u, base = work_base(ctx);
            if (base + slice_bytes(ctx, npix, S) > half) S = std::max(1u, slice_fit(ctx, npix, half > base ? half - base : 0u, S));
        }
    }
    pl.S_cap = S;
    pl.want_bytes = work_base(ctx) + slice_bytes(ctx, npix, S);
    return RT_OK;
}

} // namespace

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char* rt_build_id(void) { return RT_BUILD_ID; }

const char* rt_last_error(const RtCtx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_ctx_create(int device_id, RtCtx** out_ctx) {
    if (!out_ctx) {
        g_create_error = "rt_ctx_create: out_ctx is NULL";
        return RT_ERR_INVALID;
    }
    *out_ctx = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_create_error = std::string("rt_ctx_create: no HIP device available (") + hipGetErrorString(e) + ")";
        return RT_ERR_DEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) {
        g_create_error = "rt_ctx_create: device_id out of range";
        return RT_ERR_INVALID;
    }
    RtCtx* ctx = new (std::nothrow) RtCtx();
    if (!ctx) {
        g_create_error = "rt_ctx_create: out of host memory";
        return RT_ERR_NOMEM;
    }
    ctx->device = device_id;
    ctx->pool.device = device_id;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = std::string("rt_ctx_create: ") + what + ": " + hipGetErrorString(err);
        if (ctx->h_overflow) (void)hipHostFree(ctx->h_overflow);
        pool_destroy(ctx->pool);
        delete ctx;
        return RT_ERR_DEVICE;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->lds_limit = prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 64 * 1024; // 160 KiB on gfx950
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    // Kernels whose dynamic LDS can exceed the 64 KB default: the attribute is process-global per function, so it is
    // set to the device limit once (a per-scene value would be lowered by the next context's smaller scene).
    {
        const void* variants[] = {
#define RT_ISECT_VARIANTS(G, R, N, T) reinterpret_cast<const void*>(&k_intersect<RT_BVH_BLOCK, G, R, N, T>)
            RT_ISECT_VARIANTS(false, false, true, false), RT_ISECT_VARIANTS(false, true, true, false),
            RT_ISECT_VARIANTS(true, false, true, false),  RT_ISECT_VARIANTS(true, true, true, false),
            RT_ISECT_VARIANTS(false, true, false, false), RT_ISECT_VARIANTS(true, true, false, false),
            RT_ISECT_VARIANTS(false, true, true, true),   RT_ISECT_VARIANTS(true, true, true, true),
            RT_ISECT_VARIANTS(false, true, false, true),  RT_ISECT_VARIANTS(true, true, false, true),
#undef RT_ISECT_VARIANTS
#define RT_ISECT_VARIANTS(G, N, T) reinterpret_cast<const void*>(&k_intersect<RT_BVH_BLOCK, G, true, N, T, true>)
            RT_ISECT_VARIANTS(false, true, false), RT_ISECT_VARIANTS(true, true, false), RT_ISECT_VARIANTS(false, false, false), RT_ISECT_VARIANTS(true, false, false),
            RT_ISECT_VARIANTS(false, true, true),  RT_ISECT_VARIANTS(true, true, true),  RT_ISECT_VARIANTS(false, false, true),  RT_ISECT_VARIANTS(true, false, true),
#undef RT_ISECT_VARIANTS
            reinterpret_cast<const void*>(&k_intersect_grid<true>), reinterpret_cast<const void*>(&k_intersect_grid<false>),
            reinterpret_cast<const void*>(&k_debug_bounce<RT_BVH_BLOCK, true, true>),
            reinterpret_cast<const void*>(&k_debug_bounce<RT_BVH_BLOCK, true, false>),
#define RT_SHADE_VARIANTS(P, G) reinterpret_cast<const void*>(&k_shade<P, G, false>), reinterpret_cast<const void*>(&k_shade<P, G, true>), reinterpret_cast<const void*>(&k_shade<P, G, true, true>)
            RT_SHADE_VARIANTS(true, true), RT_SHADE_VARIANTS(true, false), RT_SHADE_VARIANTS(false, true), RT_SHADE_VARIANTS(false, false),
#undef RT_SHADE_VARIANTS
        };
        for (const void* fn : variants)
            if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_limit)) != hipSuccess)
                return bail("hipFuncSetAttribute(MaxDynamicSharedMemorySize)", e);
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev_begin)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev_end)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipHostMalloc((void**)&ctx->h_overflow, sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    if ((e = hipHostMalloc((void**)&ctx->h_stage, RT_STAGE_BYTES, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    // the pool's helper thread starts backing the bottom of the range now (the scene and the small buffers will lie there): the
    // first mapping of a process costs 10-30 ms, which overlap the host's scene construction this way
    (void)pool_request(ctx->pool, 256u << 20);
    *out_ctx = ctx;
    return RT_OK;
}

void rt_ctx_destroy(RtCtx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    free_scene(ctx);
    free_buf(ctx->scene_region);
    free_buf(ctx->acc), free_buf(ctx->counts), free_buf(ctx->totals);
    free_buf(ctx->out_f32), free_buf(ctx->out_u8), free_buf(ctx->dbg), free_buf(ctx->genp);
    free_buf(ctx->preview_u8), free_buf(ctx->lists);
    for (auto ev : ctx->events) (void)hipEventDestroy(ev);
    for (auto ev : ctx->depth_events) (void)hipEventDestroy(ev);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    for (auto sp : ctx->parked_streams) (void)hipStreamDestroy(sp);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->h_overflow) (void)hipHostFree(ctx->h_overflow);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    pool_destroy(ctx->pool); // (after every buffer that may lie in it)
    delete ctx;
}

uint32_t rt_shard_rows(uint32_t ny, uint32_t shard_band, uint32_t shard_count, uint32_t shard_id) {
    if (shard_count <= 1) return ny;
    if (shard_band == 0) shard_band = 1;
    if (shard_id >= shard_count) return 0;
    // rows j with (j / band) % count == id: `band` rows out of every cycle of band * count, plus what the last partial cycle holds
    const uint64_t cycle = (uint64_t)shard_band * shard_count;
    const uint64_t rem = ny % cycle, lo = (uint64_t)shard_id * shard_band;
    return (uint32_t)((ny / cycle) * shard_band + (rem > lo ? std::min<uint64_t>(rem - lo, shard_band) : 0u));
}

uint32_t rt_shard_row_to_image_row(uint32_t local_row, uint32_t shard_band, uint32_t shard_count, uint32_t shard_id) {
    if (shard_count <= 1) return local_row;
    if (shard_band == 0) shard_band = 1;
    return ((local_row / shard_band) * shard_count + shard_id) * shard_band + (local_row % shard_band);
}

int rt_scene_upload(RtCtx* ctx, const RtFlatScene* s) {
    if (!ctx) return RT_ERR_INVALID;
    if (!s) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: scene is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    // ---- validate ---------------------------------------------------------------------------
    if (s->n_spheres && (!s->sph_cx || !s->sph_cy || !s->sph_cz || !s->sph_r || !s->sph_mat))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: sphere arrays missing");
    if (s->n_materials && (!s->mat_type || !s->mat_color || !s->mat_p0 || !s->mat_p1 || !s->mat_p2 || !s->mat_p3 ||
                           !s->mat_tex0 || !s->mat_tex1))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: material arrays missing");
    if (s->n_textures && (!s->tex_type || !s->tex_color0 || !s->tex_color1 || !s->tex_scale || !s->tex_aux))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: texture arrays missing");
    if (s->n_perlin && (!s->perlin_vec || !s->perlin_perm))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: perlin tables missing");
    if (s->n_images && (!s->img_w || !s->img_h || !s->img_offset || !s->texels))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: image arrays missing");
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        if (s->sph_mat[i] >= s->n_materials)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: sphere " + std::to_string(i) + " has material index out of range");
        // non-finite geometry would reach the tree builder's sort comparators and bin casts (host-side UB)
        if (!std::isfinite(s->sph_cx[i]) || !std::isfinite(s->sph_cy[i]) || !std::isfinite(s->sph_cz[i]) || !std::isfinite(s->sph_r[i]))
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: sphere " + std::to_string(i) + " has a centre or radius that is not finite");
    }
    for (uint32_t i = 0; i < 3u * s->n_rects && s->rect_min && s->rect_max; ++i)
        if (!std::isfinite(s->rect_min[i]) || !std::isfinite(s->rect_max[i]))
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: rectangle " + std::to_string(i / 3u) + " has a bound that is not finite");
    for (uint32_t i = 0; i < 4u * s->n_xforms && s->xf_param; ++i)
        if (!std::isfinite(s->xf_param[i]))
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: transform " + std::to_string(i / 4u) + " has a parameter that is not finite");
    if (s->n_xforms && (!s->xf_type || !s->xf_param || !s->xf_parent))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: transform arrays missing");
    for (uint32_t i = 0; i < s->n_xforms; ++i) {
        if (s->xf_type[i] > RT_XF_ROTATE_Y) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: unknown transform type");
        if (s->xf_parent[i] != RT_NO_XFORM && s->xf_parent[i] >= i)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: transform parent must precede its child");
    }
    // (a parent precedes its child, so every chain ends; its length is xf_depth below: any number of nested wrappers, hitable.rs:404-520)
    for (uint32_t i = 0; i < s->n_spheres && s->sph_xform; ++i)
        if (s->sph_xform[i] != RT_NO_XFORM && s->sph_xform[i] >= s->n_xforms) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: bad sphere transform chain");
    for (uint32_t i = 0; i < s->n_rects && s->rect_xform; ++i)
        if (s->rect_xform[i] != RT_NO_XFORM && s->rect_xform[i] >= s->n_xforms) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: bad rectangle transform chain");
    if (s->n_media > RT_MAX_MEDIA) return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: more than RT_MAX_MEDIA media");
    if (s->n_media && (!s->med_neg_inv_density || !s->med_mat))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: medium arrays missing");
    for (uint32_t m = 0; m < s->n_media; ++m) {
        if (s->med_xform && s->med_xform[m] != RT_NO_XFORM && s->med_xform[m] >= s->n_xforms)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: bad wrapper around medium " + std::to_string(m));
        if (s->med_mat[m] >= s->n_materials) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: medium material out of range");
        const uint32_t mt = s->mat_type[s->med_mat[m]];
        // a medium writes neither uv nor tang (hitable.rs:574-576): materials that read them see stale record state
        if (mt == RT_MAT_DISNEY_METAL) return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: DisneyMetal as a phase function reads a stale HitRecord.tang");
        if (mat_needs_tex0(mt) && s->mat_tex0[s->med_mat[m]] < s->n_textures && s->tex_type[s->mat_tex0[s->med_mat[m]]] == RT_TEX_IMAGE)
            return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: an image texture on a medium reads a stale HitRecord.uv");
    }
    for (uint32_t i = 0; i < s->n_spheres && s->sph_medium; ++i)
        if (s->sph_medium[i] != RT_NO_MEDIUM && s->sph_medium[i] >= s->n_media) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: bad sphere medium tag");
    for (uint32_t i = 0; i < s->n_rects && s->rect_medium; ++i)
        if (s->rect_medium[i] != RT_NO_MEDIUM && s->rect_medium[i] >= s->n_media) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: bad rectangle medium tag");
    if (s->n_rects && (!s->rect_axis || !s->rect_min || !s->rect_max || !s->rect_mat))
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: rectangle arrays missing");
    for (uint32_t i = 0; i < s->n_rects; ++i) {
        if (s->rect_axis[i] > RT_RECT_XY) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: unknown rectangle axis");
        if (s->rect_mat[i] >= s->n_materials)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: rectangle " + std::to_string(i) + " has material index out of range");
        // DisneyMetal reads rec.tang, which a rectangle never writes (hitable.rs:262-269): the reference then uses
        // whatever an earlier candidate left in the record — not reproducible outside its traversal order
        if (s->mat_type[s->rect_mat[i]] == RT_MAT_DISNEY_METAL)
            return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: DisneyMetal on a rectangle reads a stale HitRecord.tang in the reference");
    }
    for (uint32_t m = 0; m < s->n_materials; ++m) {
        uint32_t t = s->mat_type[m];
        if (t >= RT_MAT__COUNT) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: unknown material type");
        if (mat_needs_tex0(t) && s->mat_tex0[m] >= s->n_textures)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: material " + std::to_string(m) + " needs tex0");
        if (t == RT_MAT_ROUGH_PLASTIC && s->mat_tex1[m] >= s->n_textures)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: RoughPlastic material needs tex1");
    }
    for (uint32_t t = 0; t < s->n_textures; ++t) {
        uint32_t ty = s->tex_type[t];
        if (ty >= RT_TEX__COUNT) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: unknown texture type");
        if (ty == RT_TEX_PERLIN && s->tex_aux[t] >= s->n_perlin)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: Perlin texture references a missing table set");
        if (ty == RT_TEX_IMAGE && s->tex_aux[t] >= s->n_images)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: image texture references a missing image");
    }
    for (uint32_t k = 0; k < s->n_perlin * 3u * RT_PERLIN_POINTS; ++k)
        if (s->perlin_perm[k] >= RT_PERLIN_POINTS)
            return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: perlin permutation entry out of range");
    uint64_t n_texels = 0;
    for (uint32_t k = 0; k < s->n_images; ++k) {
        if (s->img_w[k] == 0 || s->img_h[k] == 0) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: empty image");
        if (s->img_offset[k] % 3u) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: image offset not a multiple of 3");
        uint64_t end = s->img_offset[k] + 3ull * s->img_w[k] * s->img_h[k];
        if (end > s->n_texel_floats) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: image exceeds texel pool");
        n_texels = std::max(n_texels, end / 3u);
    }
    if (s->sky_type > RT_SKY_ENV) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: unknown sky type");
    if (s->sky_type == RT_SKY_ENV && s->sky_image >= s->n_images)
        return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: env sky references a missing image");

    // ---- pack -------------------------------------------------------------------------------
    std::vector<float4> geo(s->n_spheres);
    std::vector<uint32_t> smat(s->n_spheres);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        geo[i] = make_float4(s->sph_cx[i], s->sph_cy[i], s->sph_cz[i], s->sph_r[i]);
        smat[i] = s->sph_mat[i];
    }
    std::vector<MatRec> mats(s->n_materials);
    for (uint32_t m = 0; m < s->n_materials; ++m) {
        MatRec r{};
        r.type = s->mat_type[m];
        r.tex0 = s->mat_tex0[m];
        r.tex1 = s->mat_tex1[m];
        r.cr = s->mat_color[3 * m], r.cg = s->mat_color[3 * m + 1], r.cb = s->mat_color[3 * m + 2];
        r.p0 = s->mat_p0[m], r.p1 = s->mat_p1[m], r.p2 = s->mat_p2[m], r.p3 = s->mat_p3[m];
        mats[m] = r;
    }
    std::vector<TexRec> texs(s->n_textures);
    for (uint32_t t = 0; t < s->n_textures; ++t) {
        TexRec r{};
        r.type = s->tex_type[t];
        r.aux = s->tex_aux[t];
        r.scale = s->tex_scale[t];
        r.c0r = s->tex_color0[3 * t], r.c0g = s->tex_color0[3 * t + 1], r.c0b = s->tex_color0[3 * t + 2];
        r.c1r = s->tex_color1[3 * t], r.c1g = s->tex_color1[3 * t + 1], r.c1b = s->tex_color1[3 * t + 2];
        texs[t] = r;
    }
    std::vector<float4> pvec((size_t)s->n_perlin * 256);
    std::vector<uint8_t> pperm((size_t)s->n_perlin * 768);
    for (size_t k = 0; k < pvec.size(); ++k)
        pvec[k] = make_float4(s->perlin_vec[3 * k], s->perlin_vec[3 * k + 1], s->perlin_vec[3 * k + 2], 0.0f);
    for (size_t k = 0; k < pperm.size(); ++k) pperm[k] = (uint8_t)s->perlin_perm[k];
    std::vector<unsigned short> pperm2(pperm.size()); // entry i with its successor on the 256-ring (PerlinTables)
    for (size_t k = 0; k < pperm.size(); ++k)
        pperm2[k] = (unsigned short)(pperm[k] | (pperm[(k & ~(size_t)255) + ((k + 1) & 255)] << 8));
    std::vector<ImgRec> imgs(s->n_images);
    // The texel pool: RGBA8 when every component of every image is exactly k/255 (what image::open(..).to_rgb32f() makes of
    // an 8-bit file, texture.rs:176-177; image_value() divides k by 255 again and gets the same float), float4 otherwise.
    auto as_u8 = [](float t, uint32_t& k) {
        if (!(t >= 0.0f && t <= 1.0f)) return false;
        k = (uint32_t)std::lrintf(t * 255.0f);
        const float back = (float)k / 255.0f;
        return k <= 255u && std::memcmp(&back, &t, sizeof(float)) == 0;
    };
    bool all_u8 = s->n_images > 0 && ctx->opt[RT_OPT_TEXEL_POOL] != 1u;
    for (uint32_t k = 0; k < s->n_images && all_u8; ++k) {
        const float* src = s->texels + s->img_offset[k];
        const size_t nc = (size_t)s->img_w[k] * s->img_h[k] * 3u;
        uint32_t q;
        for (size_t c = 0; c < nc && all_u8; ++c) all_u8 = as_u8(src[c], q);
    }
    std::vector<float4> texels(all_u8 ? 0 : (size_t)n_texels);
    std::vector<uint32_t> texels8(all_u8 ? (size_t)n_texels : 0);
    for (uint32_t k = 0; k < s->n_images; ++k) {
        uint64_t off = s->img_offset[k] / 3u;
        imgs[k] = ImgRec{s->img_w[k], s->img_h[k], (uint32_t)off, (uint32_t)(off >> 32)};
        const float* src = s->texels + s->img_offset[k];
        const size_t np = (size_t)s->img_w[k] * s->img_h[k];
        for (size_t p = 0; p < np; ++p) {
            if (all_u8) {
                uint32_t r = 0, g = 0, b = 0;
                as_u8(src[3 * p], r), as_u8(src[3 * p + 1], g), as_u8(src[3 * p + 2], b);
                texels8[off + p] = r | (g << 8) | (b << 16);
            } else {
                texels[off + p] = make_float4(src[3 * p], src[3 * p + 1], src[3 * p + 2], 0.0f);
            }
        }
    }

    // rectangles: device geometry (k, u0, u1, v0), (v1, axis) with (u, v) the uv axes of hitable.rs:262-263 etc.
    const uint32_t n_prims = s->n_spheres + s->n_rects;
    std::vector<float4> rgeo((size_t)s->n_rects * 2);
    std::vector<PrimBox> pboxes(n_prims);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const float c[3] = {geo[i].x, geo[i].y, geo[i].z};
        const float r = std::fabs(geo[i].w);
        for (int k = 0; k < 3; ++k) pboxes[i].mn[k] = c[k] - r, pboxes[i].mx[k] = c[k] + r;
    }
    auto fbits = [](uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    for (uint32_t i = 0; i < s->n_rects; ++i) {
        const uint32_t ax = s->rect_axis[i];
        const float* mn = s->rect_min + 3 * (size_t)i;
        const float* mx = s->rect_max + 3 * (size_t)i;
        const int ua = ax == 0 ? 1 : 0, va = ax == 2 ? 1 : 2;
        rgeo[2 * (size_t)i] = make_float4(mn[ax], mn[ua], mx[ua], mn[va]);
        rgeo[2 * (size_t)i + 1] = make_float4(mx[va], fbits(ax), 0.0f, 0.0f);
        PrimBox& b = pboxes[s->n_spheres + i];
        for (int k = 0; k < 3; ++k) b.mn[k] = std::min(mn[k], mx[k]), b.mx[k] = std::max(mn[k], mx[k]);
        b.mn[ax] = mn[ax] - 0.0001f, b.mx[ax] = mn[ax] + 0.0001f; // the plane the hit test uses (hitable.rs:253)
    }
    // instance wrappers: per-primitive innermost wrapper, and world-space bounds through the chain (corners
    // through every wrapper from the inside out, as RotateY::new does at hitable.rs:455-473)
    std::vector<uint32_t> pxf(n_prims, RT_NO_XFORM);
    std::vector<float4> xparam(s->n_xforms);
    std::vector<uint2> xmeta(s->n_xforms);
    // xf_meta[x] = (type, parent).  A chain of more than RT_MAX_CHAIN wrappers is listed once more behind the table, outermost wrapper
    // first, as (wrapper, chain length), and xf_param[x].w holds where — 0 for a short chain (rt_device.h: the kernels walk short
    // chains through the parent links into registers and long ones through the list).
    uint32_t n_xf_listed = 0;
    std::vector<uint32_t> xf_depth(s->n_xforms);
    std::vector<uint8_t> xf_is_innermost(s->n_xforms, 0); // only the wrapper a primitive or a medium names is ever looked up by a kernel
    for (uint32_t i = 0; i < s->n_spheres && s->sph_xform; ++i)
        if (s->sph_xform[i] != RT_NO_XFORM) xf_is_innermost[s->sph_xform[i]] = 1;
    for (uint32_t i = 0; i < s->n_rects && s->rect_xform; ++i)
        if (s->rect_xform[i] != RT_NO_XFORM) xf_is_innermost[s->rect_xform[i]] = 1;
    for (uint32_t m = 0; m < s->n_media && s->med_xform; ++m)
        if (s->med_xform[m] != RT_NO_XFORM && s->med_xform[m] < s->n_xforms) xf_is_innermost[s->med_xform[m]] = 1;
    for (uint32_t i = 0; i < s->n_xforms; ++i) {
        const uint32_t depth = xf_depth[i] = s->xf_parent[i] == RT_NO_XFORM ? 1u : xf_depth[s->xf_parent[i]] + 1u;
        xparam[i] = make_float4(s->xf_param[4 * i], s->xf_param[4 * i + 1], s->xf_param[4 * i + 2], 0.0f);
        xmeta[i] = make_uint2(s->xf_type[i], s->xf_parent[i]);
        if (depth > (uint32_t)RT_MAX_CHAIN && xf_is_innermost[i]) {
            if ((uint64_t)s->n_xforms + n_xf_listed + depth > 0x7FFFFFFFull) return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: wrapper chains too long to list");
            xparam[i].w = fbits(s->n_xforms + n_xf_listed);
            n_xf_listed += depth;
        }
    }
    xmeta.resize((size_t)s->n_xforms + n_xf_listed);
    for (uint32_t i = 0; i < s->n_xforms; ++i) {
        if (xf_depth[i] <= (uint32_t)RT_MAX_CHAIN || !xf_is_innermost[i]) continue;
        uint32_t p0, k = xf_depth[i];
        std::memcpy(&p0, &xparam[i].w, 4);
        for (uint32_t x = i; x != RT_NO_XFORM; x = s->xf_parent[x]) xmeta[p0 + --k] = make_uint2(x, xf_depth[i]);
    }
    std::vector<float4> world_sphere(s->n_spheres); // instanced spheres: centre and radius in world space (culling only)
    for (uint32_t i = 0; i < n_prims; ++i) {
        const uint32_t x0 = i < s->n_spheres ? (s->sph_xform ? s->sph_xform[i] : RT_NO_XFORM)
                                             : (s->rect_xform ? s->rect_xform[i - s->n_spheres] : RT_NO_XFORM);
        pxf[i] = x0;
        if (i < s->n_spheres && x0 != RT_NO_XFORM) {
            // A sphere below Translate / RotateY wrappers is still a sphere: its world box is centre' +- r, not the
            // box of the rotated box that RotateY::new computes (22 % wider per axis at 15 degrees, and the cloud of
            // final_scene is 1 000 overlapping instanced spheres).  The exact test runs in object space on a ray
            // whose transform rounds at the magnitude of the WORLD coordinates, so the box gets that slack.
            double c[3] = {geo[i].x, geo[i].y, geo[i].z}, mag = std::fabs(geo[i].w);
            for (uint32_t x = x0; x != RT_NO_XFORM; x = s->xf_parent[x]) {
                const float* q = s->xf_param + 4 * (size_t)x;
                for (int k = 0; k < 3; ++k) mag = std::max(mag, std::fabs(c[k]));
                if (s->xf_type[x] == RT_XF_TRANSLATE) {
                    for (int k = 0; k < 3; ++k) c[k] += (double)q[k];
                } else {
                    const double sn = q[0], cs = q[1], cx = c[0], cz = c[2];
                    c[0] = cs * cx + sn * cz, c[2] = -sn * cx + cs * cz;
                }
                for (int k = 0; k < 3; ++k) mag = std::max(mag, std::fabs(c[k]));
            }
            const double r = std::fabs((double)geo[i].w), slack = 8e-6 * mag + 1e-30;
            for (int k = 0; k < 3; ++k) pboxes[i].mn[k] = (float)(c[k] - r - slack), pboxes[i].mx[k] = (float)(c[k] + r + slack);
            world_sphere[i] = make_float4((float)c[0], (float)c[1], (float)c[2], (float)(r + 2.0 * slack));
            continue;
        }
        for (uint32_t x = x0; x != RT_NO_XFORM; x = s->xf_parent[x]) {
            PrimBox& b = pboxes[i];
            const float* q = s->xf_param + 4 * (size_t)x;
            if (s->xf_type[x] == RT_XF_TRANSLATE) {
                for (int k = 0; k < 3; ++k) b.mn[k] += q[k], b.mx[k] += q[k];
            } else {
                const float sn = q[0], cs = q[1];
                float mn[3] = {FLT_MAX, b.mn[1], FLT_MAX}, mx[3] = {-FLT_MAX, b.mx[1], -FLT_MAX};
                for (int ci = 0; ci < 4; ++ci) {
                    const float x0c = (ci & 1) ? b.mx[0] : b.mn[0], z0c = (ci & 2) ? b.mx[2] : b.mn[2];
                    const float nx = cs * x0c + sn * z0c, nz = -sn * x0c + cs * z0c;
                    mn[0] = std::min(mn[0], nx), mx[0] = std::max(mx[0], nx);
                    mn[2] = std::min(mn[2], nz), mx[2] = std::max(mx[2], nz);
                }
                // the rotation itself rounds: widen by a few ulp of the coordinate magnitude
                for (int k = 0; k < 3; k += 2) {
                    const float e = 1e-6f * std::max(std::fabs(mn[k]), std::fabs(mx[k]));
                    b.mn[k] = mn[k] - e, b.mx[k] = mx[k] + e;
                }
            }
        }
    }
    // media: boundary primitive lists; world entries = primitives that are not a boundary, then the media
    const uint32_t n_entries = n_prims + s->n_media;
    std::vector<uint32_t> pmed(n_prims, RT_NO_MEDIUM), med_prims;
    std::vector<uint2> med_range(s->n_media);
    std::vector<float> med_nid(s->n_media);
    std::vector<uint2> med_xf(s->n_media, make_uint2(RT_NO_XFORM, RT_NO_XFORM)); // (.x: the chain all boundary primitives share, .y: the wrapper around the medium)
    for (uint32_t i = 0; i < n_prims; ++i)
        pmed[i] = i < s->n_spheres ? (s->sph_medium ? s->sph_medium[i] : RT_NO_MEDIUM)
                                   : (s->rect_medium ? s->rect_medium[i - s->n_spheres] : RT_NO_MEDIUM);
    std::vector<PrimBox> eboxes;
    std::vector<uint32_t> entry_ids;
    for (uint32_t i = 0; i < n_prims; ++i)
        if (pmed[i] == RT_NO_MEDIUM) eboxes.push_back(pboxes[i]), entry_ids.push_back(i);
    for (uint32_t m = 0; m < s->n_media; ++m) {
        med_range[m] = make_uint2((uint32_t)med_prims.size(), 0u);
        med_nid[m] = s->med_neg_inv_density[m];
        PrimBox mb;
        for (int k = 0; k < 3; ++k) mb.mn[k] = FLT_MAX, mb.mx[k] = -FLT_MAX;
        for (uint32_t i = 0; i < n_prims; ++i)
            if (pmed[i] == m) {
                // one wrapper chain for the whole boundary (a GBox under RotateY/Translate): medium_root moves the ray once
                if (med_range[m].y == 0) med_xf[m].x = pxf[i];
                else if (med_xf[m].x != pxf[i]) med_xf[m].x = RT_MED_XF_MIXED;
                if (s->med_xform && s->med_xform[m] != RT_NO_XFORM) { // the wrapper around the medium lies on the chain of every boundary primitive
                    uint32_t x = pxf[i];
                    while (x != RT_NO_XFORM && x != s->med_xform[m]) x = s->xf_parent[x];
                    if (x == RT_NO_XFORM) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: the wrapper around medium " + std::to_string(m) + " is not on the chain of its boundary primitive " + std::to_string(i));
                }
                med_prims.push_back(i);
                ++med_range[m].y;
                for (int k = 0; k < 3; ++k) mb.mn[k] = std::min(mb.mn[k], pboxes[i].mn[k]), mb.mx[k] = std::max(mb.mx[k], pboxes[i].mx[k]);
            }
        if (med_range[m].y == 0) return fail(ctx, RT_ERR_INVALID, "rt_scene_upload: medium " + std::to_string(m) + " has no boundary primitives");
        if (s->med_xform) med_xf[m].y = s->med_xform[m];
        if (med_range[m].y > RT_MED_COUNT_MASK) return fail(ctx, RT_ERR_UNSUPPORTED, "rt_scene_upload: medium " + std::to_string(m) + " has too many boundary primitives");
        {   // boundaries whose two searches are answered from one evaluation of their primitives (medium_root, rt_kernels.h)
            bool all_rects = true;
            for (uint32_t k = 0; k < med_range[m].y; ++k) all_rects = all_rects && med_prims[med_range[m].x + k] >= s->n_spheres;
            uint32_t kind = 0u;
            if (med_xf[m].x != RT_MED_XF_MIXED && all_rects && med_range[m].y <= 6u) kind = RT_MED_KIND_RECTS;
            else if (med_xf[m].x != RT_MED_XF_MIXED && med_range[m].y == 1u && med_prims[med_range[m].x] < s->n_spheres) kind = RT_MED_KIND_SPHERE;
            if (ctx->opt[RT_OPT_MEDIUM_SEARCH] == 1u) kind = 0u; // test hook: the two searches as the reference makes them
            med_range[m].y |= kind << 24;
        }
        eboxes.push_back(mb);
        entry_ids.push_back(n_prims + m);
    }
    // bounding spheres of the world entries (k_primary_lists): a bare sphere is its own, anything else gets the
    // sphere around its (padded, world-space) box
    std::vector<float4> ent_bs(eboxes.size());
    for (size_t e = 0; e < eboxes.size(); ++e) {
        const uint32_t id = entry_ids[e];
        if (id < s->n_spheres && pxf[id] == RT_NO_XFORM) {
            ent_bs[e] = make_float4(s->sph_cx[id], s->sph_cy[id], s->sph_cz[id], std::fabs(s->sph_r[id]));
        } else if (id < s->n_spheres) {
            ent_bs[e] = world_sphere[id];
        } else {
            const PrimBox& b = eboxes[e];
            const double hx = 0.5 * ((double)b.mx[0] - b.mn[0]), hy = 0.5 * ((double)b.mx[1] - b.mn[1]), hz = 0.5 * ((double)b.mx[2] - b.mn[2]);
            ent_bs[e] = make_float4((float)(0.5 * ((double)b.mx[0] + b.mn[0])), (float)(0.5 * ((double)b.mx[1] + b.mn[1])),
                                    (float)(0.5 * ((double)b.mx[2] + b.mn[2])), (float)(std::sqrt(hx * hx + hy * hy + hz * hz) * 1.0001));
        }
    }
    HostBvh bvh;
    build_prim_bvh(eboxes, RT_BVH_MAX_DEPTH, bvh);
    for (auto& dd : bvh.d) { // leaf ids: index into eboxes -> world entry id
        if (dd.x < 0 && dd.x != INT_MIN) dd.x = ~(int)entry_ids[(size_t)~dd.x];
        if (dd.y < 0 && dd.y != INT_MIN) dd.y = ~(int)entry_ids[(size_t)~dd.y];
    }
    std::vector<uint8_t> sclass(n_entries);
    // (at least one record: shade() issues the loads of record 0 for a miss too, see rt_device.h, so an empty scene
    // still needs 5 readable float4)
    std::vector<float4> srec((size_t)std::max<uint32_t>(n_entries, 1u) * 5, make_float4(0.f, 0.f, 0.f, 0.f));
    bool class_present[RT_NCLASS] = {};
    class_present[0] = true; // "miss"
    for (uint32_t i = 0; i < n_entries; ++i) {
        const bool is_med = i >= n_prims;
        const bool is_rect = !is_med && i >= s->n_spheres;
        const uint32_t m = is_med ? s->med_mat[i - n_prims] : (is_rect ? s->rect_mat[i - s->n_spheres] : s->sph_mat[i]);
        const uint32_t ty = s->mat_type[m];
        const bool has_t0 = mat_needs_tex0(ty) && s->mat_tex0[m] < s->n_textures;
        const uint32_t t0 = has_t0 ? s->mat_tex0[m] : 0u;
        const uint32_t tt = has_t0 ? s->tex_type[t0] : 0u;
        sclass[i] = (uint8_t)(1u + ty * 4u + tt); // < RT_NCLASS
        class_present[sclass[i]] = true;
        // colour slot: the texture's colour 0 for textured materials, the albedo for Metal
        const float* col = has_t0 ? s->tex_color0 + 3 * (size_t)t0 : s->mat_color + 3 * (size_t)m;
        srec[5 * (size_t)i + 0] = is_med ? make_float4(0.f, 0.f, 0.f, 1.f) : (is_rect ? rgeo[2 * (size_t)(i - s->n_spheres)] : geo[i]);
        srec[5 * (size_t)i + 1] = make_float4(fbits(ty), fbits(tt), fbits(has_t0 ? s->tex_aux[t0] : 0u), fbits(s->mat_tex1[m]));
        srec[5 * (size_t)i + 2] = make_float4(col[0], col[1], col[2], s->mat_p0[m]);
        srec[5 * (size_t)i + 3] = make_float4(s->mat_p1[m], s->mat_p2[m], has_t0 ? s->tex_scale[t0] : 0.0f, fbits(s->mat_tex0[m]));
        srec[5 * (size_t)i + 4] = is_rect ? rgeo[2 * (size_t)(i - s->n_spheres) + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // sort keys of k_shade's class sort: the classes present in the scene ranked cheap classes first (a miss and the
    // constant-texture material.rs materials, then the textured and pbr.rs ones), so that the long Perlin / PBR
    // segments of a block sit together at its end
    uint32_t class_key[RT_NCLASS] = {}, n_keys = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (uint32_t c = 0; c < RT_NCLASS; ++c)
            if (class_present[c] && class_is_light(c, s->sky_type) == (pass == 0)) class_key[c] = n_keys++;
    for (uint32_t i = 0; i < n_entries; ++i) sclass[i] = (uint8_t)class_key[sclass[i]];

    HostBvh4 bvh4;
    collapse_bvh4(bvh, bvh4);

    // the arrays of the scene before lie in the region the new ones are about to be written to: nothing may still read them
    RT_HIP(ctx, hipDeviceSynchronize());
    free_scene(ctx);
    DevScene ds{};
    ds.n_xforms = s->n_xforms;
    ds.n_media = s->n_media;
    ds.n_rects = s->n_rects;
    ds.n_prims = n_prims;
    ds.n_bvh4_nodes = (uint32_t)bvh4.id.size();
    ds.n_entries = (uint32_t)eboxes.size();
    ds.n_med_prims = (uint32_t)med_prims.size();
    ds.n_xf_listed = n_xf_listed;
    // what only the NEST instantiations of k_intersect evaluate (rt_device.h): a listed chain, the media of mask bit 31, a wrapper around a medium
    ctx->nest = n_xf_listed > 0 || s->n_media > 32u;
    for (uint32_t m = 0; m < s->n_media && s->med_xform; ++m) ctx->nest = ctx->nest || s->med_xform[m] != RT_NO_XFORM;
    {   // rays whose slab slack exceeds 2^-10 of the scene extent use the cancellation-free slab test (bvh_step)
        double ext2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            float lo = FLT_MAX, hi = -FLT_MAX;
            for (const PrimBox& b : eboxes) lo = std::min(lo, b.mn[k]), hi = std::max(hi, b.mx[k]);
            if (hi > lo) ext2 += ((double)hi - lo) * ((double)hi - lo);
        }
        ds.bvh_exact_eps = (float)(std::sqrt(ext2) / 1024.0);
    }
    ds.bvh4_depth = bvh4.depth;
    ds.key_miss = class_key[0];
    ds.n_spheres = s->n_spheres, ds.n_materials = s->n_materials, ds.n_textures = s->n_textures;
    ds.n_perlin = s->n_perlin, ds.n_images = s->n_images, ds.sky_type = s->sky_type, ds.sky_image = s->sky_image;
    int rc;
    std::vector<float4> pgeo(geo);
    pgeo.insert(pgeo.end(), rgeo.begin(), rgeo.end());
    auto upload_all = [&]() -> int {
        int rc;
        if ((rc = upload(ctx, pgeo, &ds.prim_geo)) || (rc = upload(ctx, smat, &ds.sph_mat)) || (rc = upload(ctx, mats, &ds.mats)) ||
            (rc = upload(ctx, texs, &ds.texs)) || (rc = upload(ctx, pvec, &ds.perlin_vec)) ||
            (rc = upload(ctx, pperm2, &ds.perlin_perm2)) || (rc = upload(ctx, imgs, &ds.imgs)) ||
            (rc = upload(ctx, texels, &ds.texels)) || (rc = upload(ctx, texels8, &ds.texels8)) || (rc = upload(ctx, pmed, &ds.prim_medium)) || (rc = upload(ctx, med_prims, &ds.med_prims)) || (rc = upload(ctx, med_range, &ds.med_range)) || (rc = upload(ctx, med_xf, &ds.med_xform)) ||
            (rc = upload(ctx, ent_bs, &ds.ent_bs)) || (rc = upload(ctx, entry_ids, &ds.ent_leaf)) ||
            (rc = upload(ctx, med_nid, &ds.med_neg_inv_density)) || (rc = upload(ctx, pxf, &ds.prim_xform)) || (rc = upload(ctx, xparam, &ds.xf_param)) ||
            (rc = upload(ctx, xmeta, &ds.xf_meta)) || (rc = upload(ctx, sclass, &ds.sph_class)) || (rc = upload(ctx, srec, &ds.sph_rec)) || (rc = upload(ctx, bvh4.id, &ds.bvh4_id)) ||
            (rc = upload(ctx, bvh4.p[0], &ds.bvh4_p[0])) || (rc = upload(ctx, bvh4.p[1], &ds.bvh4_p[1])) ||
            (rc = upload(ctx, bvh4.p[2], &ds.bvh4_p[2])) || (rc = upload(ctx, bvh4.p[3], &ds.bvh4_p[3])) ||
            (rc = upload(ctx, bvh4.p[4], &ds.bvh4_p[4])) || (rc = upload(ctx, bvh4.p[5], &ds.bvh4_p[5]))) return rc;
        return RT_OK;
    };
    // first pass: what the arrays need (upload() only adds up); then ONE region for them, the one of the scene before when it
    // is large enough, else a new one from the pool (+ 25 %, + room for the grid's cell arrays) — no hipMalloc per array, none at
    // all in the common case, and the copies go through page-locked staging
    ctx->scene_measuring = true, ctx->scene_measure = 0;
    (void)upload_all();
    ctx->scene_measuring = false;
    {
        const size_t need = ctx->scene_measure + (1u << 20);
        if (need > ctx->scene_region.bytes && (rc = ensure(ctx, ctx->scene_region, need + need / 4u))) return rc;
        ctx->scene_used = 0;
    }
    if ((rc = upload_all())) {
        free_scene(ctx);
        return rc;
    }
    ds.sph_geo = ds.prim_geo;
    ds.rect_geo = ds.prim_geo + s->n_spheres;
    if (!all_u8) ds.texels8 = nullptr; // (upload() hands out a 16 B allocation even for an empty pool)
    ctx->ds = ds;
    ctx->has_scene = true;
    // k_intersect keeps nodes + geometry + one u16 stack column per lane in LDS when that fits 160 KB;
    // larger trees are traversed out of HBM/L2 with only the stacks in LDS; the list walk is the last resort
    const bool bvh_ok = ds.n_prims > 0 && ds.n_bvh4_nodes > 0 && ds.n_bvh4_nodes < 32768 && n_entries <= 32768 &&
                        bvh.depth <= RT_BVH_MAX_DEPTH;
    const bool force_hbm = ctx->opt[RT_OPT_TREE_PLACEMENT] == 1u; // test hook: traverse out of HBM even when LDS would fit
    // General scenes (wrappers, rectangles, media): the tree goes to LDS only when TWO workgroups per CU still fit.  Their
    // traversal is bound by dependent loads, and 8 waves per SIMD reading the tree through L2 beat 4 waves reading it from
    // LDS: a final_scene-like scene of 600-1 300 primitives runs 16-19 % faster with its tree in L2 and two workgroups than
    // with tree AND wrapper tables in LDS and one (profiles/round3/final_like.txt) — which is also why final_scene's tree was
    // not squeezed into LDS with quantised boxes: 1 150 nodes x 48 B + 56 KB of stacks leave room for one workgroup only.
    // Sphere-only scenes keep their faster LDS-only kernel (sorted slab planes) even at one workgroup per CU.
    const bool general = ds.n_rects > 0 || ds.n_xforms > 0 || ds.n_media > 0 || ctx->opt[RT_OPT_GENERAL_KERNELS] == 1u;
    ctx->general_kernels = general;
    const size_t lds_budget = general ? ctx->lds_limit / 2 : ctx->lds_limit;
    ctx->bvh_in_lds = bvh_ok && bvh_lds_bytes(ds, RT_BVH_BLOCK, true) <= lds_budget && !force_hbm;
    ctx->isect_lds = bvh_lds_bytes(ds, RT_BVH_BLOCK, ctx->bvh_in_lds);
    ctx->use_bvh = bvh_ok && ctx->isect_lds <= ctx->lds_limit;
    // general scenes: wrapper / medium tables behind the tree carve, when two workgroups per CU still fit
    ctx->general_lds = false;
    if (ctx->use_bvh && (ds.n_xforms || ds.n_media) && ctx->isect_lds + general_lds_bytes(ds) <= ctx->lds_limit / 2 &&
        ctx->opt[RT_OPT_GENERAL_LDS] != 1u) {
        ctx->isect_lds += general_lds_bytes(ds);
        ctx->general_lds = true;
    }
    // Sphere-only scenes: a uniform grid over the spheres for the rays of depth >= 1 (rt_grid.h), when the scene suits one and
    // two workgroups per CU still fit.  The tree stays: depth 0 (candidate-list overflow), the single-kernel test hook and
    // RT_OPT_GRID = 1 use it, and the tests hold the two searches against each other bit for bit.
    ctx->use_grid = false;
    if (ctx->use_bvh && ctx->bvh_in_lds && !general && ctx->opt[RT_OPT_GRID] != 1u) {
        HostGrid hg;
        build_sphere_grid(geo, ctx->lds_limit / 2, (double)ctx->opt[RT_OPT_GRID_CELL] * 1e-3, hg);
        if (hg.ok) {
            if ((rc = upload(ctx, hg.cells, &hg.gp.cells)) || (rc = upload(ctx, hg.refs, &hg.gp.refs))) {
                free_scene(ctx);
                return rc;
            }
            hg.gp.sph_geo = ds.sph_geo;
            ctx->grid = hg.gp;
            ctx->grid_lds = grid_lds_bytes(hg.gp.n_spheres, hg.gp.n_cells, hg.gp.n_refs);
            ctx->use_grid = true;
        }
    }
    return RT_OK;
}

static int check_params(RtCtx* ctx, const RtCamera* cam, const RtParams* p) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_STATE, "render: no scene uploaded");
    if (!cam || !p) return fail(ctx, RT_ERR_INVALID, "render: camera/params NULL");
    if (p->nx == 0 || p->ny == 0 || p->spp == 0) return fail(ctx, RT_ERR_INVALID, "render: nx, ny and spp must be > 0");
    if (p->max_depth < 0 || p->max_depth > 4096) return fail(ctx, RT_ERR_INVALID, "render: max_depth out of range [0, 4096]");
    if ((uint64_t)p->nx * p->ny > 0xFFFFFFFFull) return fail(ctx, RT_ERR_INVALID, "render: image too large");
    if (p->shard_count > 1 && p->shard_id >= p->shard_count) return fail(ctx, RT_ERR_INVALID, "render: shard_id >= shard_count");
    // slot -> (sample, row, column) goes through udiv_inv (rt_kernels.h), exact for quotients below 2^21: the sample index is
    // capped by the slice size, the row index by this
    if (rt_shard_rows(p->ny, p->shard_band ? p->shard_band : 1u, p->shard_count, p->shard_id) >= (1u << 21))
        return fail(ctx, RT_ERR_UNSUPPORTED, "render: a shard of 2^21 or more image rows is not supported");
    return RT_OK;
}

// h_out_*: host destinations of the two images (rt_render) or NULL; the copies are enqueued behind the last kernel, before the
// one synchronisation that reads the counters.
static int render_impl(RtCtx* ctx, const RtCamera* cam, const RtParams* prm, void* d_out_rgb_f32, void* d_out_rgb8,
                       void* stream_v, RtStats* stats, float* h_out_f32 = nullptr, uint8_t* h_out_u8 = nullptr) {
    int rc = check_params(ctx, cam, prm);
    if (rc) return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream_v ? (hipStream_t)stream_v : ctx->stream;
    const auto wall0 = std::chrono::steady_clock::now();
    ctx->parts_begin();

    const uint32_t nx = prm->nx, ny = prm->ny, spp = prm->spp;
    const uint32_t band = prm->shard_band ? prm->shard_band : 1u;
    const uint32_t scount = prm->shard_count <= 1 ? 1u : prm->shard_count;
    const uint32_t rows = rt_shard_rows(ny, band, scount, prm->shard_id);
    const uint64_t npix64 = (uint64_t)rows * nx;
    if (npix64 == 0) {
        if (stats) std::memset(stats, 0, sizeof(*stats));
        return RT_OK;
    }
    const uint32_t npix = (uint32_t)npix64;
    const int n_depths = prm->max_depth + 1;
    const uint32_t nq = queue_shards(ctx);
    const bool use_bvh = ctx->use_bvh && !(prm->flags & RT_FLAG_BRUTE_FORCE);
    // depth 0 regenerates the primary ray in both kernels instead of materialising the queue
    const bool fuse_gen = use_bvh && ctx->opt[RT_OPT_MATERIALISE_PRIMARIES] != 1u;
    // Candidate lists of the primary rays, once per frame (k_primary_lists): worth it when the samples of a pixel
    // share them (>= 4 spp) and pixels see few entries.  Measured per 128-spp slice: sphere_scene (533 entries)
    // 46.7 -> 41.9 ms, pbr_sweep_scene 41.2 -> 39.3, test_sphere 15.5 -> 14.8, cornell_box unchanged (its walls'
    // bounding spheres cover every pixel: overflow); final_scene (3 408 entries, most pixels overflow) would pay
    // 3 ms for nothing, hence the cap.
    const bool want_lists = use_bvh && fuse_gen && spp >= 4 && ctx->ds.n_entries > 0 && ctx->ds.n_entries <= 2048 && ctx->opt[RT_OPT_PRIMARY_LISTS] != 1u;

    // the small persistent buffers first (they lie at the bottom of the pool); the work buffers of the slices start above them
    ctx->pool.chunk_delay_us.store(ctx->opt[RT_OPT_POOL_CHUNK_DELAY_US]);
    if ((rc = ensure(ctx, ctx->acc, (size_t)npix * 3 * sizeof(float)))) return rc;
    const size_t counts_bytes = (size_t)(n_depths + 1) * nq * sizeof(uint32_t); // queue sizes [depth][shard]
    if ((rc = ensure(ctx, ctx->counts, counts_bytes))) return rc;
    const size_t totals_bytes = (size_t)(n_depths + 2) * sizeof(unsigned long long);
    if ((rc = ensure(ctx, ctx->totals, totals_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->genp, sizeof(GenParams)))) return rc;
    if (want_lists && (rc = ensure(ctx, ctx->lists, ((size_t)npix + 1u) * sizeof(uint4)))) return rc; // + the overflow counter behind the lists
    GenParams* gpd = (GenParams*)ctx->genp.p;
    float* acc = (float*)ctx->acc.p;
    uint32_t* counts = (uint32_t*)ctx->counts.p;
    unsigned long long* totals = (unsigned long long*)ctx->totals.p; // [0]=tex fetches [1]=bad dirs [2..]=rays per depth
    SlicePlan plan;
    if ((rc = plan_slices(ctx, prm, npix64, plan))) return rc;
    if (const char* e = pool_request(ctx->pool, plan.want_bytes)) return fail(ctx, RT_ERR_NOMEM, std::string("render: ") + e);
    const size_t wbase = work_base(ctx);
    const uint32_t isect_grid = queue_geom(ctx, npix * plan.S_cap).isect_grid;
    ctx->mark("small_buffers_and_pool_request");

    // (before the frame's own clock starts: once per launch stream, 0.3 ms of spin kernels; see ensure_concurrent_chains)
    if (nq >= 2u * RT_ISECT_MAX_SHARDS && (rc = ensure_concurrent_chains(ctx, st))) return rc;
    ctx->mark("queue_probe"); // (the first kernel launch of a process: the code object is loaded here)
    RT_HIP(ctx, hipEventRecord(ctx->ev_begin, st));
    RT_HIP(ctx, hipMemsetAsync(acc, 0, (size_t)npix * 3 * sizeof(float), st));
    RT_HIP(ctx, hipMemsetAsync(totals, 0, totals_bytes, st));

    GenParams gp{};
    for (int k = 0; k < 3; ++k) {
        gp.cam_origin[k] = cam->origin[k];
        gp.cam_horizontal[k] = cam->horizontal[k];
        gp.cam_vertical[k] = cam->vertical[k];
        gp.cam_llc[k] = cam->lower_left_corner[k];
    }
    gp.nx = nx, gp.ny = ny, gp.npix = npix;
    gp.shard_band = band, gp.shard_count = scount, gp.shard_id = prm->shard_id;
    gp.nq = nq, gp.cap = 0u; // (cap: per slice)
    gp.seed_lo = (uint32_t)prm->seed, gp.seed_hi = (uint32_t)(prm->seed >> 32);
    gp.lists = nullptr;
    gp.n_overflow = nullptr;
    bool no_overflow = false; // every pixel of this frame has a candidate list (read back from k_primary_lists below)
    {   // udiv_inv: reciprocals that keep the float quotient at or below the true one
        auto inv = [](uint32_t d) { return (float)((1.0 / (double)d) * (1.0 - 1.0 / 4194304.0)); };
        gp.inv_npix = inv(npix), gp.inv_nx = inv(nx), gp.inv_band = inv(band);
        // 8 x 8 pixel tiles per wave of depth 0 when the shard's frame allows it (rt_kernels.h GenParams): a strip of 64 x 1 pixels
        // crosses more silhouettes than a block of 8 x 8, and at depth 0 a wave runs the union of what its lanes hit.  Frames are
        // bit-identical (keys and the resolve order are functions of (pixel, sample)).  sphere_scene: depth-0 shading 7.17 -> 6.67 ms
        // per 128 spp, frame 58.6 -> 57.4 ms; pbr_sweep_scene -1.5 %; cornell_box +-0; final_scene +1.7 % (its deeper bounces,
        // 42.0 -> 43.6 ms of k_intersect), so general scenes keep the rows (profiles/round3/ab_tiles.txt).
        const uint32_t po = ctx->opt[RT_OPT_PIXEL_ORDER]; // 1 = rows, 2 = tiles wherever the frame allows
        const bool want_tiles = po ? po == 2u : !scene_is_general(ctx);
        gp.tiles_per_row = (nx % 8u == 0u && rows >= 8u && want_tiles) ? nx / 8u : 0u;
        gp.tile_pixels = gp.tiles_per_row * 64u * (rows / 8u); // the last rows % 8 rows stay row-major
        gp.inv_tpr = gp.tiles_per_row ? inv(gp.tiles_per_row) : 0.0f;
    }
    if (want_lists) {
        gp.lists = (const uint4*)ctx->lists.p;
        uint32_t* n_overflow = (uint32_t*)((uint4*)ctx->lists.p + npix);
        gp.n_overflow = n_overflow;
        RT_HIP(ctx, hipMemsetAsync(n_overflow, 0, sizeof(uint4), st));
        hipLaunchKernelGGL(k_primary_lists, dim3((npix + 255u) / 256u), dim3(256), (size_t)ctx->ds.n_entries * sizeof(float4) + 4u * RT_LIST_WAVE_CAP * 2u, st,
                           ctx->ds, gp, (uint4*)ctx->lists.p, n_overflow);
        // The one host decision of a frame: with no overflowing list (every headline configuration) depth 0 of a sphere-only scene
        // needs no closest-hit launch at all — k_shade<GEN> finds every hit from the lists.  An empty launch is not free: each of
        // its workgroups waits for 66 KB of LDS behind the other chain's shading waves and holds its own chain's shading back
        // (config 2: 1.6 ms of 47.6).  The count is there 0.1 ms into the frame; reading it costs one stream synchronisation while
        // nothing else of the frame is enqueued yet, and the first frame of a view is as fast as any later one (the reference
        // renders exactly one, main.rs:62-129).
        if (!scene_is_general(ctx)) {
            RT_HIP(ctx, hipMemcpyAsync(ctx->h_overflow, n_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            RT_HIP(ctx, hipStreamSynchronize(st));
            no_overflow = *ctx->h_overflow == 0u;
        }
        ctx->mark("primary_lists_and_in_frame_sync");
    }

    uint32_t n_trace_launches = 0;
    const bool time_depths = (prm->flags & RT_FLAG_TIME_DEPTHS) != 0;
    ctx->timed_depths = 0;
    if (time_depths) {
        while (ctx->depth_events.size() < 3 * (size_t)n_depths) {
            hipEvent_t ev;
            RT_HIP(ctx, hipEventCreate(&ev));
            ctx->depth_events.push_back(ev);
        }
    }
    IntersectParams ip{nq, 0u, 0, 0u, nq};
    // Two shard groups on two streams: k_intersect is bound by VALU issue and leaves HBM idle, k_shade past depth 0
    // waits on HBM and leaves the VALUs idle (profiles/round2).  Each half of the shards runs its own chain
    // intersect -> shade -> intersect ... on its own stream; the two chains drift apart because the kernels differ in
    // length, and the dispatcher fills every CU with waves of both kinds.  Same kernels, same results; config 2 runs
    // 2.2 % faster than with one chain over all shards (68.0 -> 66.5 ms; three, four and eight groups: 83 / 82 / 104 ms;
    // forcing the two chains half a step apart with events, so that one always intersects while the other shades: 75 ms,
    // half-size grids in lockstep).  The two half-grid launches of a depth run side by side, so a "launch" in RtStats is
    // the logical one — one kernel, one depth, all shards — and rocprof shows each half lasting about that long.
    // Per-depth timing needs the single chain.
    // Scenes whose k_intersect waits on dependent loads like k_shade does — wrapper chains, media, a tree read through L2 —
    // run ONE chain: two would only share the same pipes (final_scene 41.4 -> 39.3 ms, cornell_box 32.7 -> 30.8 ms per 64 spp
    // with one; sphere_scene 18.0 -> 18.6, simple_light_scene's bare rectangles likewise prefer two:
    // profiles/round3/one_stream.txt).  RT_OPT_CHAINS = 1 / 2 force either.
    const uint32_t chains_opt = ctx->opt[RT_OPT_CHAINS];
    const bool one_chain = chains_opt ? chains_opt == 1u : (ctx->ds.n_xforms > 0 || ctx->ds.n_media > 0 || (use_bvh && !ctx->bvh_in_lds));
    const uint32_t n_groups = (nq >= 2u * RT_ISECT_MAX_SHARDS && !time_depths && !one_chain) ? 2u : 1u;
    const uint32_t shards_per_wg = (nq + isect_grid - 1u) / isect_grid;
    // selects the "general scene" kernel instantiations (rectangles and Translate / RotateY wrappers)
    const bool rects = scene_is_general(ctx);
    // Slices: as many samples per pixel as the plan allows and the pool holds NOW.  While the helper thread is still backing the
    // pool (the first frame of a process, rt_pool.h) a frame whose slice size the library chose starts with what is mapped — at
    // least 1/32 of its samples, one such slice ahead of the device at most — instead of waiting for the rest; slices are
    // independent and the frame is bit-identical for any slicing.  A slice size the caller asked for is waited for.
    uint32_t n_slices = 0, s0 = 0;
    const uint32_t S_progress = std::min(plan.S_cap, std::max(1u, (spp + 31u) / 32u));
    double pool_wait_ms = 0.0;
    while (s0 < spp) {
        const uint32_t sl = n_slices;
        const uint32_t sc_max = std::min(plan.S_cap, spp - s0);
        uint32_t sc = 0;
        for (;;) {
            const size_t mapped = ctx->pool.mapped.load();
            const uint32_t fits = slice_fit(ctx, npix, mapped > wbase ? mapped - wbase : 0u, sc_max);
            if (fits >= sc_max) { sc = sc_max; break; }
            const bool growing = pool_growing(ctx->pool);
            if (!growing) { // the device had no more to give: the frame takes more slices; nothing at all is an error
                if (ctx->pool.mapped.load() != mapped) continue;
                if (fits == 0u) return fail(ctx, RT_ERR_NOMEM, "render: no device memory for the work buffers of one sample per pixel (" +
                                                                    std::to_string(slice_bytes(ctx, npix, 1u) >> 20) + " MB): " + pool_error(ctx->pool));
                sc = fits;
                break;
            }
            // one small slice ahead of the device at most: the next is cut when the device is about to run dry
            const bool device_idle = sl == 0u || hipEventQuery(ctx->events[2 * (size_t)sl - 1]) == hipSuccess;
            if (plan.by_library && fits >= std::min(S_progress, sc_max) && device_idle) { sc = fits; break; }
            const auto tw = std::chrono::steady_clock::now();
            pool_wait_progress(ctx->pool, mapped, 200u);
            pool_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
        }
        (void)hipGetLastError(); // (hipEventQuery: hipErrorNotReady is not an error)
        const uint32_t n_max = npix * sc;
        const uint32_t cap = queue_geom(ctx, n_max).cap;
        const WorkView wv = work_view(ctx, work_layout((size_t)nq * cap, n_max), wbase);
        const Queue* Q = wv.Q;
        float2* qhit = wv.qhit;
        float* rad = wv.rad;
        gp.cap = cap, ip.cap = cap;
        while (ctx->events.size() < 2 * (size_t)(sl + 1u)) {
            hipEvent_t ev;
            RT_HIP(ctx, hipEventCreate(&ev));
            ctx->events.push_back(ev);
        }
        ++n_slices;
        gp.s0 = s0;
        gp.n_rays = npix * sc;
        RT_HIP(ctx, hipMemsetAsync(counts, 0, counts_bytes, st));
        hipLaunchKernelGGL(k_init_counts, dim3((nq + 255u) / 256u), dim3(256), 0, st, gp, counts, gpd);
        if (!fuse_gen) hipLaunchKernelGGL(k_gen_primary, dim3((gp.n_rays + 255u) / 256u), dim3(256), 0, st, gp, Q[0]);
        RT_HIP(ctx, hipEventRecord(ctx->events[2 * sl], st));
        if (n_groups > 1u) {
            RT_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
            RT_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        }
        // (Starting the second chain k depths behind the first — one wait on the first chain's depth-k shading, then free —
        // so that the chains are in different phases: config 2 62.9 ms -> 69.6 / 71.4 / 71.0 / 72.9 ms for k = 0 / 1 / 2 / 4,
        // profiles/round3/stagger_sphere_scene.txt.  The chains do best side by side.)
        for (int depth = 0; depth < n_depths; ++depth)
        for (uint32_t grp = 0; grp < n_groups; ++grp) {
            hipStream_t sg = grp ? ctx->stream2 : st;
            const uint32_t q0 = grp * (nq / n_groups), q1 = grp + 1u == n_groups ? nq : q0 + nq / n_groups;
            const uint32_t isect_grid_g = (q1 - q0 + shards_per_wg - 1u) / shards_per_wg;
            const Queue& qi = Q[depth & 1];
            const Queue& qo = Q[(depth + 1) & 1];
            const uint32_t* cin = counts + (size_t)depth * nq;
            uint32_t* cout = counts + (size_t)(depth + 1) * nq;
            const bool td = time_depths && sl == 0;
            if (td) RT_HIP(ctx, hipEventRecord(ctx->depth_events[3 * (size_t)depth], st));
            const bool gen = fuse_gen && depth == 0;
            ip.depth = depth;
            ip.q0 = q0, ip.q1 = q1;
            const StepBuffers sb{qi, qo, qhit, cin, cout, rad, totals, gpd};
            // depth 0 of a sphere-only scene whose pixels all have a candidate list: k_shade<GEN> finds every closest hit itself
            const bool no_primary_trace = gen && !rects && gp.lists != nullptr && no_overflow;
            if (!no_primary_trace) launch_intersect(ctx, sg, use_bvh, gen, isect_grid_g, sb, ip);
            if (td) RT_HIP(ctx, hipEventRecord(ctx->depth_events[3 * (size_t)depth + 1], st));
            // class sort from depth 1 on: primary rays are coherent already (measured: sorting depth 0 costs 8 %)
            const ShadeParams sp{nq, cap, depth, prm->max_depth, depth > 0 ? 1u : 0u,
                                 (prm->flags & RT_FLAG_RUSSIAN_ROULETTE) ? 1u : 0u, q0};
            launch_shade(ctx, sg, gen, !rects && gp.lists != nullptr, q1 - q0, sb, sp);
            if (td) RT_HIP(ctx, hipEventRecord(ctx->depth_events[3 * (size_t)depth + 2], st));
            if (grp == 0u) n_trace_launches += no_primary_trace ? 1u : 2u;
        }
        if (n_groups > 1u) {
            RT_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
            RT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
        }
        if (time_depths && sl == 0) {
            ctx->timed_depths = n_depths;
            ctx->timed_rays.assign((size_t)n_depths, 0ull);
            std::vector<uint32_t> hc((size_t)(n_depths + 1) * nq);
            RT_HIP(ctx, hipMemcpyAsync(hc.data(), counts, hc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            RT_HIP(ctx, hipStreamSynchronize(st));
            for (int d = 0; d < n_depths; ++d)
                for (uint32_t k = 0; k < nq; ++k) ctx->timed_rays[(size_t)d] += hc[(size_t)d * nq + k];
        }
        RT_HIP(ctx, hipEventRecord(ctx->events[2 * sl + 1], st));
        hipLaunchKernelGGL(k_resolve, dim3((npix + 255u) / 256u), dim3(256), 0, st, rad, acc, npix, sc);
        hipLaunchKernelGGL(k_accum_counts, dim3((unsigned)n_depths), dim3(256), 0, st, counts, nq, (uint32_t)n_depths, totals + 2);
        if (ctx->progress_armed && ctx->progress_fn && s0 + sc < spp) { // the last slice is the final image itself
            const uint32_t done = s0 + sc;
            hipLaunchKernelGGL(k_finalize, dim3((npix + 255u) / 256u), dim3(256), 0, st, acc, (float*)nullptr,
                               (uint8_t*)ctx->preview_u8.p, nx, rows, done, gp.tiles_per_row, gp.tile_pixels);
            RT_HIP(ctx, hipMemcpyAsync(ctx->preview_host.data(), ctx->preview_u8.p, (size_t)npix * 3, hipMemcpyDeviceToHost, st));
            RT_HIP(ctx, hipStreamSynchronize(st));
            ctx->progress_fn(ctx->progress_user, done, spp, ctx->preview_host.data(), nx, rows);
        }
        s0 += sc;
    }
    if (pool_wait_ms > 0.0) ctx->parts.emplace_back("of_which_waiting_for_the_pool", pool_wait_ms);
    hipLaunchKernelGGL(k_finalize, dim3((npix + 255u) / 256u), dim3(256), 0, st, acc, (float*)d_out_rgb_f32,
                       (uint8_t*)d_out_rgb8, nx, rows, spp, gp.tiles_per_row, gp.tile_pixels);
    RT_HIP(ctx, hipEventRecord(ctx->ev_end, st));
    RT_HIP(ctx, hipGetLastError());
    ctx->mark("enqueue_all_launches");
    // Device -> host.  A destination in pinned host memory (rt_host_alloc, or memory the caller registered with HIP) is written
    // by the copy engine at PCIe rate while this thread goes on; a pageable one makes hipMemcpyAsync stage and wait, as
    // hipMemcpy would (24.9 + 6.2 MB of a 1920 x 1080 frame: ~3 ms against ~0.6 ms).
    if (h_out_f32) RT_HIP(ctx, hipMemcpyAsync(h_out_f32, d_out_rgb_f32, (size_t)npix * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    if (h_out_u8) RT_HIP(ctx, hipMemcpyAsync(h_out_u8, d_out_rgb8, (size_t)npix * 3, hipMemcpyDeviceToHost, st));
    if ((h_out_f32 || h_out_u8) && !stats) RT_HIP(ctx, hipStreamSynchronize(st));

    if (stats) {
        RT_HIP(ctx, hipStreamSynchronize(st));
        ctx->mark("wait_for_the_device");
        std::vector<unsigned long long> h((size_t)n_depths + 2);
        RT_HIP(ctx, hipMemcpy(h.data(), totals, totals_bytes, hipMemcpyDeviceToHost));
        std::memset(stats, 0, sizeof(*stats));
        stats->n_paths = (uint64_t)npix * spp;
        stats->n_texture_fetches = h[0];
        stats->n_bad_dir = h[1];
        for (int d = 0; d < n_depths; ++d) {
            stats->n_rays += h[(size_t)d + 2];
            if (d < 64) stats->rays_per_depth[d] = h[(size_t)d + 2];
        }
        stats->n_rays_secondary = stats->n_rays - std::min(stats->n_rays, stats->n_paths);
        float ms = 0.0f;
        double trace_ms = 0.0;
        for (uint32_t sl = 0; sl < n_slices; ++sl) {
            RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->events[2 * sl], ctx->events[2 * sl + 1]));
            trace_ms += ms;
        }
        RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        stats->seconds_trace = trace_ms * 1e-3;
        stats->seconds_device = ms * 1e-3;
        stats->bytes_algorithmic = 96ull * stats->n_rays + 24ull * stats->n_paths + 12ull * stats->n_texture_fetches;
        stats->bytes_trace_algorithmic = 48ull * stats->n_rays + 48ull * stats->n_rays_secondary + 12ull * stats->n_paths +
                                         12ull * stats->n_texture_fetches;
        stats->n_trace_launches = n_trace_launches;
        stats->n_slices = n_slices;
        stats->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    }
    return RT_OK;
}

int rt_prepare(RtCtx* ctx, const RtParams* prm) {
    if (!ctx) return RT_ERR_INVALID;
    if (!prm || prm->nx == 0 || prm->ny == 0 || prm->spp == 0) return fail(ctx, RT_ERR_INVALID, "rt_prepare: params NULL, or nx, ny or spp 0");
    if (prm->shard_count > 1 && prm->shard_id >= prm->shard_count) return fail(ctx, RT_ERR_INVALID, "rt_prepare: shard_id >= shard_count");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t npix64 = (uint64_t)rt_shard_rows(prm->ny, prm->shard_band ? prm->shard_band : 1u, prm->shard_count, prm->shard_id) * prm->nx;
    if (npix64 == 0) return RT_OK;
    if (npix64 > 0xFFFFFFFFull) return fail(ctx, RT_ERR_INVALID, "rt_prepare: image too large");
    SlicePlan plan;
    int rc = plan_slices(ctx, prm, npix64, plan);
    if (rc) return rc;
    ctx->pool.chunk_delay_us.store(ctx->opt[RT_OPT_POOL_CHUNK_DELAY_US]);
    // + what the frame's small buffers will take below the work area (accumulator, candidate lists, the two output images: 43 B a
    // pixel) and the scene (an estimate: a pool that turns out a little short simply keeps growing while the frame starts)
    const size_t small = (size_t)npix64 * 48u + (64u << 20);
    if (const char* e = pool_request(ctx->pool, plan.want_bytes + small)) return fail(ctx, RT_ERR_NOMEM, std::string("rt_prepare: ") + e);
    return RT_OK;
}

int rt_render_device(RtCtx* ctx, const RtCamera* cam, const RtParams* prm, void* d_out_rgb_f32, void* stream_v,
                     RtStats* stats) {
    if (ctx && !d_out_rgb_f32) return fail(ctx, RT_ERR_INVALID, "rt_render_device: output pointer is NULL");
    return render_impl(ctx, cam, prm, d_out_rgb_f32, nullptr, stream_v, stats);
}

int rt_render(RtCtx* ctx, const RtCamera* cam, const RtParams* prm, float* out_rgb_f32, uint8_t* out_rgb8, RtStats* stats) {
    int rc = check_params(ctx, cam, prm);
    if (rc) return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t band = prm->shard_band ? prm->shard_band : 1u;
    const uint32_t rows = rt_shard_rows(prm->ny, band, prm->shard_count, prm->shard_id);
    const size_t n = (size_t)rows * prm->nx * 3;
    if ((rc = ensure(ctx, ctx->out_f32, n * sizeof(float)))) return rc;
    // the u8 image is produced only on request
    if (out_rgb8 && (rc = ensure(ctx, ctx->out_u8, n))) return rc;
    if (ctx->progress_fn) {
        if ((rc = ensure(ctx, ctx->preview_u8, n))) return rc;
        ctx->preview_host.resize(n);
    }
    RtStats local;
    ctx->progress_armed = true;
    rc = render_impl(ctx, cam, prm, ctx->out_f32.p, out_rgb8 ? ctx->out_u8.p : nullptr, nullptr, stats ? stats : &local,
                     n ? out_rgb_f32 : nullptr, n ? out_rgb8 : nullptr);
    ctx->progress_armed = false;
    return rc;
}

void* rt_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void rt_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int rt_debug_set_option(RtCtx* ctx, uint32_t option, uint32_t value) {
    if (!ctx) return RT_ERR_INVALID;
    if (option >= RT_OPT__COUNT) return fail(ctx, RT_ERR_INVALID, "rt_debug_set_option: unknown option");
    ctx->opt[option] = value;
    return RT_OK;
}

int rt_debug_render_parts(const RtCtx* ctx, char* buf, uint32_t cap) {
    if (!ctx) return RT_ERR_INVALID;
    std::string js = "{";
    char num[64];
    for (size_t k = 0; k < ctx->parts.size(); ++k) {
        std::snprintf(num, sizeof(num), "\": %.3f", ctx->parts[k].second);
        js += (k ? ", \"" : "\"") + std::string(ctx->parts[k].first) + num;
    }
    std::snprintf(num, sizeof(num), "%s\"pool_mapped_mb\": %zu", ctx->parts.empty() ? "" : ", ", ctx->pool.mapped.load() >> 20);
    js += num;
    std::snprintf(num, sizeof(num), ", \"pool_chunks_grown\": %u", ctx->pool.n_grown.load());
    js += num;
    std::snprintf(num, sizeof(num), ", \"pool_slowest_chunk_ms\": %.3f", ctx->pool.slowest_chunk_ms.load());
    js += num;
    js += "}";
    if (buf && cap) {
        const size_t n = std::min<size_t>(js.size(), cap - 1u);
        std::memcpy(buf, js.data(), n);
        buf[n] = 0;
    }
    return (int)js.size() + 1;
}

int rt_debug_get_option(const RtCtx* ctx, uint32_t option, uint32_t* value) {
    if (!ctx || !value || option >= RT_OPT__COUNT) return RT_ERR_INVALID;
    *value = ctx->opt[option];
    return RT_OK;
}

int rt_debug_grid_build(const RtFlatScene* s, uint32_t cell_per_mille, uint32_t lds_budget, float grid[8], uint32_t dims[3], uint32_t* cells,
                        uint32_t* n_cells, uint16_t* refs, uint32_t* n_refs, uint32_t large[4], uint32_t* n_large) {
    if (!s || !grid || !dims || !n_cells || !n_refs || !large || !n_large) return RT_ERR_INVALID;
    if (s->n_rects || s->n_xforms || s->n_media || (s->n_spheres && (!s->sph_cx || !s->sph_cy || !s->sph_cz || !s->sph_r))) return RT_ERR_UNSUPPORTED;
    std::vector<float4> geo(s->n_spheres);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        if (!std::isfinite(s->sph_cx[i]) || !std::isfinite(s->sph_cy[i]) || !std::isfinite(s->sph_cz[i]) || !std::isfinite(s->sph_r[i])) return RT_ERR_INVALID;
        geo[i] = make_float4(s->sph_cx[i], s->sph_cy[i], s->sph_cz[i], s->sph_r[i]);
    }
    HostGrid hg;
    build_sphere_grid(geo, lds_budget ? lds_budget : 80u * 1024u, (double)cell_per_mille * 1e-3, hg);
    if (!hg.ok) return RT_ERR_UNSUPPORTED;
    const uint32_t nc = hg.gp.n_cells, nr = hg.gp.all_rec >> RT_GRID_CNT_BITS; // (the all-spheres list behind the cell lists is not reported)
    const bool fits = cells && refs && *n_cells >= nc && *n_refs >= nr;
    *n_cells = nc, *n_refs = nr;
    if (!fits) return RT_ERR_INVALID;
    for (int k = 0; k < 3; ++k) grid[k] = hg.gp.g0[k], grid[3 + k] = hg.gp.cs[k];
    grid[7] = hg.gp.max_coord;
    grid[6] = hg.gp.max_coord / 1048576.0f; // pad (build_sphere_grid: max_coord = pad * 2^20)
    dims[0] = hg.gp.nx, dims[1] = hg.gp.ny, dims[2] = hg.gp.nz;
    std::copy(hg.cells.begin(), hg.cells.end(), cells);
    std::copy(hg.refs.begin(), hg.refs.begin() + nr, refs);
    *n_large = hg.gp.n_always;
    for (uint32_t k = 0; k < hg.gp.n_always; ++k) large[k] = hg.gp.always[k];
    return RT_OK;
}

int rt_debug_scene_info(const RtCtx* ctx, RtSceneInfo* info) {
    if (!ctx || !info) return RT_ERR_INVALID;
    std::memset(info, 0, sizeof(*info));
    if (!ctx->has_scene) return RT_ERR_STATE;
    info->n_entries = ctx->ds.n_entries;
    info->n_tree_nodes = ctx->ds.n_bvh4_nodes;
    info->tree_depth = ctx->ds.bvh4_depth;
    info->tree_in_lds = ctx->use_bvh && ctx->bvh_in_lds;
    info->general_kernels = scene_is_general(ctx);
    info->closest_hit_lds_bytes = (uint32_t)ctx->isect_lds;
    info->grid = ctx->use_grid;
    info->general_tables_in_lds = ctx->general_lds;
    info->nest = ctx->nest;
    if (ctx->use_grid) {
        info->grid_cells[0] = ctx->grid.nx, info->grid_cells[1] = ctx->grid.ny, info->grid_cells[2] = ctx->grid.nz;
        info->grid_refs = ctx->grid.all_rec >> RT_GRID_CNT_BITS;
        info->grid_always = ctx->grid.n_always;
        info->grid_lds_bytes = (uint32_t)ctx->grid_lds;
        for (int k = 0; k < 3; ++k) info->grid_cell_size[k] = ctx->grid.cs[k];
    }
    return RT_OK;
}

int rt_set_progress(RtCtx* ctx, RtProgressFn fn, void* user) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->progress_fn = fn;
    ctx->progress_user = user;
    return RT_OK;
}

int rt_get_depth_timings(RtCtx* ctx, uint32_t max_n, float* isect_ms, float* shade_ms, uint64_t* rays) {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int n = std::min<int>(ctx->timed_depths, (int)max_n);
    for (int d = 0; d < n; ++d) {
        float a_ms = 0.f, b_ms = 0.f;
        RT_HIP(ctx, hipEventElapsedTime(&a_ms, ctx->depth_events[3 * (size_t)d], ctx->depth_events[3 * (size_t)d + 1]));
        RT_HIP(ctx, hipEventElapsedTime(&b_ms, ctx->depth_events[3 * (size_t)d + 1], ctx->depth_events[3 * (size_t)d + 2]));
        if (isect_ms) isect_ms[d] = a_ms;
        if (shade_ms) shade_ms[d] = b_ms;
        if (rays) rays[d] = ctx->timed_rays[(size_t)d];
    }
    return n;
}

// rt_debug_bounce with RT_FLAG_PRODUCTION_KERNELS: the rays go through the queue and the kernels rt_render launches for a
// depth >= 1 (the scene's k_intersect instantiation, then the class-sorting k_shade), and the per-ray outcome is read
// back from the queues: hit records by queue position, radiance slots of the finished paths, survivors by slot.
static int debug_bounce_production(RtCtx* ctx, const RtBounceIO* io) {
    const uint32_t n = io->n;
    hipStream_t st = ctx->stream;
    const QueueGeom qg = queue_geom(ctx, n);
    const uint32_t nq = qg.nq, cap = qg.cap;
    int rc;
    if ((rc = ensure(ctx, ctx->counts, (size_t)2 * nq * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->totals, 4 * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(ctx, ctx->genp, sizeof(GenParams)))) return rc;
    if ((rc = ensure(ctx, ctx->dbg, (size_t)n * 6 * sizeof(float)))) return rc;
    const WorkLayout wl = work_layout((size_t)nq * cap, n);
    const size_t wbase = work_base(ctx);
    if (const char* e = pool_request(ctx->pool, wbase + wl.total)) return fail(ctx, RT_ERR_NOMEM, std::string("rt_debug_bounce: ") + e);
    if (!pool_wait(ctx->pool, wbase + wl.total)) return fail(ctx, RT_ERR_NOMEM, "rt_debug_bounce: no device memory for the work buffers: " + pool_error(ctx->pool));
    const WorkView wv = work_view(ctx, wl, wbase);
    const Queue* Q = wv.Q;
    uint32_t* counts = (uint32_t*)ctx->counts.p;
    float* d_o = (float*)ctx->dbg.p;
    float* d_d = d_o + 3 * (size_t)n;
    RT_HIP(ctx, hipMemcpyAsync(d_o, io->in_o, 3 * (size_t)n * 4, hipMemcpyHostToDevice, st));
    RT_HIP(ctx, hipMemcpyAsync(d_d, io->in_d, 3 * (size_t)n * 4, hipMemcpyHostToDevice, st));
    RT_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)2 * nq * sizeof(uint32_t), st));
    RT_HIP(ctx, hipMemsetAsync(ctx->totals.p, 0, 4 * sizeof(unsigned long long), st));
    RT_HIP(ctx, hipMemsetAsync(wv.rad, 0xFF, (size_t)n * RT_RAD_FLOATS * sizeof(float), st)); // NaN pattern: a survivor has no slot
    // one "image row" of n pixels, one sample: slot i is pixel i, so its key is path_key(seed 0, pix i, sample 0)
    GenParams gp{};
    gp.nx = n, gp.ny = 1, gp.npix = n, gp.n_rays = n, gp.s0 = 0;
    gp.shard_band = 1, gp.shard_count = 1, gp.shard_id = 0;
    gp.nq = nq, gp.cap = cap;
    gp.inv_npix = gp.inv_nx = (float)((1.0 / (double)n) * (1.0 - 1.0 / 4194304.0));
    gp.inv_band = (float)(1.0 - 1.0 / 4194304.0);
    hipLaunchKernelGGL(k_init_counts, dim3((nq + 255u) / 256u), dim3(256), 0, st, gp, counts, (GenParams*)ctx->genp.p);
    hipLaunchKernelGGL(k_debug_fill, dim3((n + 255u) / 256u), dim3(256), 0, st, gp, Q[0], d_o, d_d);
    const bool use_bvh = ctx->use_bvh && !(io->flags & RT_FLAG_BRUTE_FORCE);
    const StepBuffers sb{Q[0], Q[1], wv.qhit, counts, counts + nq, wv.rad,
                         (unsigned long long*)ctx->totals.p, (const GenParams*)ctx->genp.p};
    const IntersectParams ip{nq, cap, (int)io->depth, 0u, nq};
    launch_intersect(ctx, st, use_bvh, false, qg.isect_grid, sb, ip);
    const ShadeParams sp{nq, cap, (int)io->depth, 0x7FFFFFFF, 1u, 0u, 0u};
    launch_shade(ctx, st, false, false, nq, sb, sp);
    RT_HIP(ctx, hipGetLastError());
    std::vector<float4> ha((size_t)nq * cap), hb((size_t)nq * cap);
    std::vector<float> hrad((size_t)n * RT_RAD_FLOATS);
    std::vector<float2> hc((size_t)nq * cap), hh((size_t)nq * cap);
    std::vector<uint32_t> hcnt((size_t)2 * nq);
    RT_HIP(ctx, hipMemcpyAsync(hh.data(), wv.qhit, hh.size() * sizeof(float2), hipMemcpyDeviceToHost, st));
    std::vector<float4> hab; // RT_QSTRIDE 2: the interleaved a / b records, split on the host below
    if (RT_QSTRIDE == 1u) {
        RT_HIP(ctx, hipMemcpyAsync(ha.data(), Q[1].a, ha.size() * sizeof(float4), hipMemcpyDeviceToHost, st));
        RT_HIP(ctx, hipMemcpyAsync(hb.data(), Q[1].b, hb.size() * sizeof(float4), hipMemcpyDeviceToHost, st));
    } else {
        hab.resize(2 * ha.size());
        RT_HIP(ctx, hipMemcpyAsync(hab.data(), Q[1].a, hab.size() * sizeof(float4), hipMemcpyDeviceToHost, st));
    }
    RT_HIP(ctx, hipMemcpyAsync(hc.data(), Q[1].c, hc.size() * sizeof(float2), hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(hrad.data(), wv.rad, hrad.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(hcnt.data(), counts, hcnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < hab.size() / 2; ++i) ha[i] = hab[2 * i], hb[i] = hab[2 * i + 1];
    auto bits = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    };
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t chunk = i >> 8;
        const size_t pos = (size_t)(chunk % nq) * cap + (size_t)(chunk / nq) * 256u + (i & 255u);
        const int hit = (int)bits(hh[pos].y);
        io->out_hit[i] = hit;
        io->out_t[i] = hit >= 0 ? hh[pos].x : 0.0f;
        io->out_alive[i] = 0;
        for (int k = 0; k < 3; ++k) io->out_attenuation[3 * (size_t)i + k] = io->out_o[3 * (size_t)i + k] = io->out_d[3 * (size_t)i + k] = 0.0f;
        for (int c = 0; c < 3; ++c) io->out_radiance[3 * (size_t)i + c] = hrad[RT_RAD_FLOATS * (size_t)i + c];
    }
    size_t n_out = 0;
    for (uint32_t q = 0; q < nq; ++q) {
        if (hcnt[nq + q] > cap) return fail(ctx, RT_ERR_DEVICE, "rt_debug_bounce: shard overflow");
        for (uint32_t k = 0; k < hcnt[nq + q]; ++k, ++n_out) {
            const size_t pos = (size_t)q * cap + k;
            const uint32_t slot = bits(ha[pos].w);
            if (slot >= n || io->out_alive[slot]) return fail(ctx, RT_ERR_DEVICE, "rt_debug_bounce: bad or repeated slot in the output queue");
            io->out_alive[slot] = 1;
            io->out_o[3 * (size_t)slot] = ha[pos].x, io->out_o[3 * (size_t)slot + 1] = ha[pos].y, io->out_o[3 * (size_t)slot + 2] = ha[pos].z;
            io->out_d[3 * (size_t)slot] = hb[pos].x, io->out_d[3 * (size_t)slot + 1] = hb[pos].y, io->out_d[3 * (size_t)slot + 2] = hb[pos].z;
            io->out_attenuation[3 * (size_t)slot] = hb[pos].w, io->out_attenuation[3 * (size_t)slot + 1] = hc[pos].x, io->out_attenuation[3 * (size_t)slot + 2] = hc[pos].y;
            for (int c = 0; c < 3; ++c) io->out_radiance[3 * (size_t)slot + c] = 0.0f; // emitted = 0 on every scattering material
        }
    }
    return RT_OK;
}

int rt_debug_bounce(RtCtx* ctx, const RtBounceIO* io) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_STATE, "rt_debug_bounce: no scene uploaded");
    if (!io || !io->in_o || !io->in_d || !io->in_key || !io->out_hit || !io->out_t || !io->out_radiance ||
        !io->out_attenuation || !io->out_o || !io->out_d || !io->out_alive)
        return fail(ctx, RT_ERR_INVALID, "rt_debug_bounce: NULL array");
    if (io->n == 0) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (io->flags & RT_FLAG_PRODUCTION_KERNELS) return debug_bounce_production(ctx, io);
    const size_t n = io->n;
    // layout of the scratch buffer (floats unless noted)
    const size_t off_o = 0, off_d = off_o + 3 * n, off_key = off_d + 3 * n, off_hit = off_key + 2 * n, off_t = off_hit + n,
                 off_rad = off_t + n, off_att = off_rad + 3 * n, off_so = off_att + 3 * n, off_sd = off_so + 3 * n,
                 off_alive = off_sd + 3 * n, total = off_alive + (n + 3) / 4 + 4;
    int rc;
    if ((rc = ensure(ctx, ctx->dbg, total * 4))) return rc;
    float* base = (float*)ctx->dbg.p;
    hipStream_t st = ctx->stream;
    RT_HIP(ctx, hipMemcpyAsync(base + off_o, io->in_o, 3 * n * 4, hipMemcpyHostToDevice, st));
    RT_HIP(ctx, hipMemcpyAsync(base + off_d, io->in_d, 3 * n * 4, hipMemcpyHostToDevice, st));
    RT_HIP(ctx, hipMemcpyAsync(base + off_key, io->in_key, 2 * n * 4, hipMemcpyHostToDevice, st));
    const bool use_bvh = ctx->use_bvh && !(io->flags & RT_FLAG_BRUTE_FORCE);
    if (use_bvh && !ctx->bvh_in_lds) {
        hipLaunchKernelGGL((k_debug_bounce<RT_BVH_BLOCK, true, false>), dim3((unsigned)((n + RT_BVH_BLOCK - 1) / RT_BVH_BLOCK)),
                           dim3(RT_BVH_BLOCK), ctx->isect_lds, st, ctx->ds, (uint32_t)n, (int)io->depth, base + off_o, base + off_d,
                           (const uint32_t*)(base + off_key), (int*)(base + off_hit), base + off_t, base + off_rad, base + off_att,
                           base + off_so, base + off_sd, (uint8_t*)(base + off_alive));
    } else if (use_bvh) {
        hipLaunchKernelGGL((k_debug_bounce<RT_BVH_BLOCK, true, true>), dim3((unsigned)((n + RT_BVH_BLOCK - 1) / RT_BVH_BLOCK)),
                           dim3(RT_BVH_BLOCK), ctx->isect_lds, st, ctx->ds, (uint32_t)n, (int)io->depth, base + off_o, base + off_d,
                           (const uint32_t*)(base + off_key), (int*)(base + off_hit), base + off_t, base + off_rad, base + off_att,
                           base + off_so, base + off_sd, (uint8_t*)(base + off_alive));
    } else {
        const size_t lds_bytes = (size_t)std::min<uint32_t>(std::max<uint32_t>(ctx->ds.n_spheres, 1u), RT_SPHERE_TILE) * sizeof(float4);
        hipLaunchKernelGGL((k_debug_bounce<256, false, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), lds_bytes, st, ctx->ds,
                           (uint32_t)n, (int)io->depth, base + off_o, base + off_d, (const uint32_t*)(base + off_key),
                           (int*)(base + off_hit), base + off_t, base + off_rad, base + off_att, base + off_so, base + off_sd,
                           (uint8_t*)(base + off_alive));
    }
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(io->out_hit, base + off_hit, n * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(io->out_t, base + off_t, n * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(io->out_radiance, base + off_rad, 3 * n * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(io->out_attenuation, base + off_att, 3 * n * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(io->out_o, base + off_so, 3 * n * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(io->out_d, base + off_sd, 3 * n * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipMemcpyAsync(io->out_alive, base + off_alive, n, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipStreamSynchronize(st));
    return RT_OK;
}

} // extern "C"

int rt_debug_arithmetic(RtCtx* ctx, uint32_t op, uint32_t n, const float* x, const float* a, float* out) {
    if (!ctx) return RT_ERR_INVALID;
    if (op > RT_ARITH_TO_U32) return fail(ctx, RT_ERR_INVALID, "rt_debug_arithmetic: unknown operation");
    if (!x || !out || (op == RT_ARITH_SHARED_DIVISION && !a)) return fail(ctx, RT_ERR_INVALID, "rt_debug_arithmetic: NULL array");
    if (n == 0) return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->dbg, (size_t)n * 3 * sizeof(float)))) return rc;
    float* dx = (float*)ctx->dbg.p;
    float* da = dx + n;
    float* dq = da + n;
    hipStream_t st = ctx->stream;
    RT_HIP(ctx, hipMemcpyAsync(dx, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (op == RT_ARITH_SHARED_DIVISION) RT_HIP(ctx, hipMemcpyAsync(da, a, (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_debug_arithmetic, dim3((n + 255u) / 256u), dim3(256), 0, st, op, n, dx, da, dq);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(out, dq, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    RT_HIP(ctx, hipStreamSynchronize(st));
    return RT_OK;
}

#ifdef RT_PROFILE_LANES
// Diagnostic builds only (not declared in include/rtow_mi355x.h, absent from the product library): the lane statistics
// of rt_kernels.h, optionally reset after reading.
extern "C" int rt_debug_lane_stats(unsigned long long* out24, int reset) { // (RT_LANE_STAT_N values, 48 since round 6)
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(rt::g_lane_stats), RT_LANE_STAT_N * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long zero[RT_LANE_STAT_N] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_lane_stats), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef RT_PROFILE_PHASES
// Diagnostic builds only: the phase clocks of the class-sorting k_shade (rt_kernels.h), optionally reset after reading.
extern "C" int rt_debug_phase_stats(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rt::g_phase_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long zero[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rt::g_phase_stats), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#include "rt_multi.h"
