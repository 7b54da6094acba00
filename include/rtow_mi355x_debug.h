/*
 * rtow_mi355x_debug.h — test hooks and diagnostics of librtow_mi355x.so.
 *
 * Nothing here is part of the drop-in boundary: a host that replaces main.rs:62-129 binds include/rtow_mi355x.h alone
 * (lifecycle, scene upload, render, multi-GPU, progress).  The entry points below live in the same library so that the
 * parity tests can drive single bounces, hold equivalent search structures against each other and read what upload built;
 * the benchmark reads per-depth and first-frame timings through them.  Same conventions as rtow_mi355x.h.
 */
#ifndef RTOW_MI355X_DEBUG_H
#define RTOW_MI355X_DEBUG_H

#include "rtow_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* -- diagnostic RtParams.flags ------------------------------------------------------------------------------------- */
/* Record HIP events around the kernels of every depth of the FIRST slice; read them back with
 * rt_get_depth_timings().  Diagnostic only (adds two event records per depth). */
#define RT_FLAG_TIME_DEPTHS 4u
/* rt_debug_bounce only: run the rays through the ray queue and the SAME kernels rt_render launches for a depth >= 1
 * (the scene's closest-hit kernel with persistent lanes, then the class-sorting shading kernel with its wave64
 * compaction) instead of the unsorted single-kernel test path.  The per-path RNG key is then the one the renderer
 * derives from the slot: ray i gets path_key(seed 0, pixel i, sample 0) and in_key is ignored; out_attenuation,
 * out_o and out_d are filled for surviving rays only (a finished path keeps no ray), out_radiance for finished ones. */
#define RT_FLAG_PRODUCTION_KERNELS 8u

/* Per-depth device times of the first slice of the last render that had RT_FLAG_TIME_DEPTHS set:
 * isect_ms[d] = closest-hit kernel, shade_ms[d] = shading kernel, rays[d] = rays traced at depth d
 * in that slice.
 * Returns the number of depths written (<= max_n), or a negative RT_ERR_*. */
int rt_get_depth_timings(RtCtx* ctx, uint32_t max_n, float* isect_ms, float* shade_ms, uint64_t* rays);

/* Host-side timeline of the last rt_render / rt_render_device of `ctx`: a JSON object {"label": milliseconds, ...} in call order
 * (work-buffer allocations one by one, the hardware-queue probe = the first kernel launch of a process, the candidate lists with
 * the one in-frame synchronisation, enqueueing every launch, waiting for the device).  The first frame of a process is what the
 * reference's own timer covers (main.rs:62-129, utils.rs:15-18); bench.py reports its parts from here.  Writes at most cap bytes
 * (NUL-terminated) and returns the size needed, or a negative RT_ERR_*. */
int rt_debug_render_parts(const RtCtx* ctx, char* buf, uint32_t cap);

/* -- debug / tuning options (test hooks) ---------------------------------------------------------------------------
 * Per context (not per process: the library reads no environment variable); every setting renders the same image — the
 * options select between equivalent search structures, placements and orders so that tests can hold them against each
 * other, and so that measurements can vary one thing.  Same BITS with one stated exception: the closest-hit searches
 * (list walk, tree, grid, candidate lists) agree on every ray except those for which fp32 Sphere::hit (hitable.rs:75-91)
 * reports a root although the ray misses the sphere in exact arithmetic (cancellation at grazing incidence); a box or cell
 * test may cull such a false positive, the list walk cannot.  Measured: at most 8 of the 1.35e9 rays of config 2, each
 * proven a false positive in float64 by the tests.  Which of them the reference's own binary BvhNode would cull is unpinned.  0 is the library's own choice for every option.  Options marked
 * (upload) take effect at the next rt_scene_upload, the others at the next render. */
enum RtDebugOption {
    RT_OPT_TREE_PLACEMENT = 0,        /* (upload) 1: the BVH is read through L2 even when it would fit LDS */
    RT_OPT_PRIMARY_LISTS = 1,         /* 1: no per-pixel candidate lists, depth 0 walks the tree */
    RT_OPT_PIXEL_ORDER = 2,           /* 1: path slots enumerate pixels row by row, 2: in 8 x 8 tiles wherever the frame allows */
    RT_OPT_TEXEL_POOL = 3,            /* (upload) 1: float4 texel pool even when every texel is k/255 */
    RT_OPT_GRID = 4,                  /* 1: no uniform grid, sphere-only scenes walk the tree at every depth */
    RT_OPT_GRID_CELL = 5,             /* (upload) grid cell edge in 1/1000 of the median sphere diameter */
    RT_OPT_CHAINS = 6,                /* 1: one chain of launches per slice, 2: two shard groups on two streams */
    RT_OPT_GENERAL_KERNELS = 7,       /* (upload) 1: the general-scene kernel instantiations on a sphere-only scene */
    RT_OPT_GENERAL_LDS = 8,           /* (upload) 1: wrapper / medium tables stay in HBM */
    RT_OPT_QUEUE_SHARDS = 9,          /* n: queue shards (default 8 per CU) */
    RT_OPT_ISECT_WORKGROUPS = 10,     /* n: closest-hit workgroups per launch */
    RT_OPT_MATERIALISE_PRIMARIES = 11,/* 1: primary rays are written to the queue by their own kernel instead of regenerated */
    RT_OPT_MEDIUM_SEARCH = 12,        /* (upload) 1: ConstantMedium::hit evaluates its boundary twice, as the reference does, also where one
                                       * evaluation answers both searches (a box, a sphere) */
    RT_OPT_POOL_CHUNK_DELAY_US = 13,  /* n: the helper thread that backs the work-buffer pool (csrc/rt_pool.h) takes n microseconds longer per
                                       * 128 MB chunk — a device that hands out memory slowly, for the test of frames that start in what
                                       * has arrived so far */
    RT_OPT__COUNT = 14
};
int rt_debug_set_option(RtCtx* ctx, uint32_t option, uint32_t value);
int rt_debug_get_option(const RtCtx* ctx, uint32_t option, uint32_t* value);

/* What rt_scene_upload built for the closest-hit search of the uploaded scene. */
typedef struct RtSceneInfo {
    uint32_t n_entries;            /* world entries: primitives that are not a medium boundary + media */
    uint32_t n_tree_nodes, tree_depth;
    uint32_t tree_in_lds;          /* 1: the BVH4 is staged in LDS, 0: read through L2 */
    uint32_t general_kernels;      /* 1: rectangles / wrappers / media (or forced) */
    uint32_t closest_hit_lds_bytes;
    uint32_t grid;                 /* 1: depth >= 1 walks a uniform grid (sphere-only scenes, csrc/rt_grid.h) */
    uint32_t grid_cells[3];
    uint32_t grid_refs;            /* sphere references in the cell lists */
    uint32_t grid_always;          /* large spheres tested for every ray */
    uint32_t grid_lds_bytes;
    float grid_cell_size[3];
    uint32_t general_tables_in_lds; /* 1: the wrapper / medium tables of a general scene are staged in LDS beside the tree or its stacks */
    uint32_t nest;                  /* 1: the scene nests beyond what the kernels keep in registers (a wrapper chain of more than 4, more
                                     * than 32 media, a wrapper around a medium): the instantiations with the loops run (csrc/rt_device.h) */
} RtSceneInfo;
int rt_debug_scene_info(const RtCtx* ctx, RtSceneInfo* info);

/* The uniform grid rt_scene_upload would build over the spheres of `scene` (host code only: no context, no GPU), for tests of
 * its construction.  cell_per_mille as RT_OPT_GRID_CELL (0 = default), lds_budget in bytes (0 = 80 KiB, two workgroups per CU).
 * Returns RT_ERR_UNSUPPORTED when the scene gets no grid, RT_ERR_INVALID when a buffer is too small (the needed sizes are
 * then in *n_cells / *n_refs), else RT_OK with: grid[0..2] = min corner, grid[3..5] = cell edges, grid[6] = pad, grid[7] =
 * origin-coordinate limit; dims[0..2] = cells per axis; cells[c] = offset << 12 | count (x fastest); refs = sphere ids of the
 * cell lists; large[0..*n_large) = the spheres tested for every ray. */
int rt_debug_grid_build(const RtFlatScene* scene, uint32_t cell_per_mille, uint32_t lds_budget, float grid[8], uint32_t dims[3],
                        uint32_t* cells, uint32_t* n_cells, uint16_t* refs, uint32_t* n_refs, uint32_t large[4], uint32_t* n_large);

#ifdef RT_PROFILE_LANES
/* Diagnostic builds only (-DRT_PROFILE_LANES; absent from the product library): the lane-occupancy counters of
 * csrc/rt_kernels.h, optionally reset after reading. */
int rt_debug_lane_stats(unsigned long long* out24, int reset);
#endif

/* -- single-bounce evaluation (test hook) --------------------------------------------------
 * Runs ONE closest-hit + shade step (main.rs:44-58 for one depth) over `n` caller-given
 * rays on the GPU without queue compaction and returns the per-ray outcome, so that each
 * material / texture / sky branch can be compared with the CPU oracle function by
 * function.  Arrays are host pointers, n entries each (vec3 as 3 floats).
 */
typedef struct RtBounceIO {
    uint32_t n;
    uint32_t depth;            /* RNG counter block = depth (see DESIGN.md "RNG") */
    const float* in_o;         /* [3n] */
    const float* in_d;         /* [3n] */
    const uint32_t* in_key;    /* [2n] per-path RNG key (k0,k1) */
    int32_t* out_hit;          /* [n] primitive index (sphere i, or n_spheres + rect i) or -1 */
    float* out_t;              /* [n] */
    float* out_radiance;       /* [3n] emitted or sky term of this segment (untinted) */
    float* out_attenuation;    /* [3n] */
    float* out_o;              /* [3n] scattered ray */
    float* out_d;              /* [3n] */
    uint8_t* out_alive;        /* [n] 1 = scatter returned true */
    uint32_t flags;            /* RT_FLAG_* (RT_FLAG_BRUTE_FORCE selects the list walk) */
} RtBounceIO;
int rt_debug_bounce(RtCtx* ctx, const RtBounceIO* io);

/* Test hook for the arithmetic routines of the kernels that are not the compiler's operators (csrc/rt_device.h; host arrays
 * of n floats):
 *   RT_ARITH_SHARED_DIVISION  out[i] = x[i] / a[i] as the kernels compute the roots of a ray (divisor |d|^2, hitable.rs:85-89) and
 *       the normal of a sphere (divisor r, hitable.rs:95): the compiler's own fp32 division sequence with the refined reciprocal
 *       of the divisor shared between the quotients and without the operand scaling that only extreme exponents need;
 *   RT_ARITH_SQRT             out[i] = sqrt(x[i]) as the kernels take it of a discriminant and of a squared length: the compiler's
 *       own sequence without the scaling of arguments below 2^-96 (`a` is not read);
 *   RT_ARITH_TO_I32, _TO_U32  out[i] = the BITS of `x[i] as i32` / `x[i] as u32` with Rust's rule (toward zero, saturating, NaN -> 0;
 *       math.rs:137-152 offset_hit_point, texture.rs:183-193): v_cvt_i32_f32 / v_cvt_u32_f32 (`a` is not read).
 * The test holds the first two against IEEE bit for bit over the operand range the kernels use them on and maps where they may
 * differ, the conversions against the Rust rule over every kind of argument. */
#define RT_ARITH_SHARED_DIVISION 0u
#define RT_ARITH_SQRT 1u
#define RT_ARITH_TO_I32 2u
#define RT_ARITH_TO_U32 3u
int rt_debug_arithmetic(RtCtx* ctx, uint32_t op, uint32_t n, const float* x, const float* a, float* out);


#ifdef __cplusplus
}
#endif
#endif /* RTOW_MI355X_DEBUG_H */
