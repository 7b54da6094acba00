/*
 * rtow_mi355x.h — C-ABI of the MI355X (gfx950) wavefront path tracer.
 *
 * This is the drop-in boundary for the per-pixel integration loop of
 * zhouhang95/ray_tracing_in_one_weekend.  The reference has no FFI of its own: the seam is
 * the pair (scene builder -> renderer) described in SURVEY.md §8(b).  Every entry point
 * below names the reference interface it replaces (paths relative to /root/reference).
 *
 *   scene side : demo_scene.rs:37,229  fn(aspect_ratio) -> (Vec<Arc<dyn Hitable>>, Camera)
 *                lib.rs:11             SKY_COLOR (global sky fn pointer)
 *   render side: main.rs:77-108        the per-column pixel loop
 *                main.rs:38-60         ray_color (closest hit -> emitted/scatter -> recurse)
 *   output     : main.rs:98-105,127    /spp, gamma 2, *255.99 as u8, vertical flip
 *
 * Conventions
 *   - plain C structs, pointers and sizes; no C++/torch types; no exceptions cross it.
 *   - every function returns 0 on success, a negative RT_ERR_* otherwise; text via
 *     rt_last_error().
 *   - the caller owns all buffers; rt_scene_upload() copies and retains nothing host-side.
 *   - a context (RtCtx) is bound to ONE GPU and is driven by one host thread; multi-GPU sharding is
 *     expressed through RtParams.shard_* (one process per GPU, the host gathers — bench.py does it with
 *     torch.distributed) or handled inside the library by an RtMulti (one process, n GPUs, RCCL gather).
 *   - fp32 throughout; image row 0 is the BOTTOM row in the f32 output (reference `j`
 *     order, main.rs:84) and the TOP row in the RGB8 output (after the flip, main.rs:127).
 */
#ifndef RTOW_MI355X_H
#define RTOW_MI355X_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 9u /* 9: RT_OPT_MEDIUM_SEARCH, rt_debug_arithmetic (additions; every v8 layout and prototype unchanged) */

/* error codes */
#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / inconsistent scene            */
#define RT_ERR_DEVICE (-2)    /* HIP runtime error (text in rt_last_error)   */
#define RT_ERR_NOMEM (-3)     /* host or device allocation failed            */
#define RT_ERR_UNSUPPORTED (-4)
#define RT_ERR_STATE (-5)     /* e.g. render before scene upload             */

/* Material tags.  Order follows SURVEY.md §3.4; one tag per `impl Material`. */
enum RtMatType {
    RT_MAT_EMISSION = 0,         /* material.rs:17-29   tex0 = emit                          */
    RT_MAT_DIFFUSE = 1,          /* material.rs:31-46   tex0 = albedo                        */
    RT_MAT_LAMBERT = 2,          /* material.rs:48-59   tex0 = albedo                        */
    RT_MAT_METAL = 3,            /* material.rs:61-73   color = albedo, p0 = fuzz            */
    RT_MAT_DIELECTRIC = 4,       /* material.rs:75-97   p0 = ior                             */
    RT_MAT_ISOTROPIC = 5,        /* material.rs:99-113  tex0 = albedo                        */
    RT_MAT_OREN_NAYAR = 6,       /* pbr.rs:12-42        tex0 = albedo, p0 = roughness        */
    RT_MAT_BURLEY_DIFFUSE = 7,   /* pbr.rs:45-69        tex0 = albedo, p0 = roughness        */
    RT_MAT_ROUGH_PLASTIC = 8,    /* pbr.rs:153-189      tex0 = spec, tex1 = diff, p0 = roughness, p1 = eta */
    RT_MAT_DISNEY_DIFFUSE = 9,   /* pbr.rs:192-222      tex0 = albedo, p0 = roughness, p1 = subsurface */
    RT_MAT_DISNEY_METAL = 10,    /* pbr.rs:224-278      tex0 = albedo, p0 = roughness, p1 = anisotropic, p2 = rot */
    RT_MAT_DISNEY_SHEEN = 11,    /* pbr.rs:281-308      tex0 = albedo, p0 = tint             */
    RT_MAT_DISNEY_CLEARCOAT = 12,/* pbr.rs:310-335      p0 = clearcoat_gloss                 */
    RT_MAT__COUNT = 13
};

/* Texture tags, one per `impl Texture` (texture.rs). */
enum RtTexType {
    RT_TEX_CONSTANT = 0, /* texture.rs:15-23    color0 = col                                  */
    RT_TEX_CHECKER = 1,  /* texture.rs:25-49    color0 = odd, color1 = even                    */
    RT_TEX_PERLIN = 2,   /* texture.rs:153-168  scale, aux = index of the Perlin table set     */
    RT_TEX_IMAGE = 3,    /* texture.rs:170-193  aux = image index                              */
    RT_TEX__COUNT = 4
};

/* Sky models (demo_scene.rs:22-35; selected through lib.rs:11 SKY_COLOR). */
enum RtSkyType {
    RT_SKY_GRADIENT = 0, /* sky_color      demo_scene.rs:28-31 */
    RT_SKY_BLACK = 1,    /* black_sky      demo_scene.rs:33-35 */
    RT_SKY_ENV = 2       /* tex_sky_color  demo_scene.rs:22-26, image = sky_image */
};

/* Axis-aligned rectangle kinds (hitable.rs:244-362): the axis that is constant on the rectangle.
 * The plane coordinate is min[axis] (the reference reads `self.min.z` etc. and ignores max[axis]). */
enum RtRectAxis {
    RT_RECT_YZ = 0, /* YZRect hitable.rs:324-362  x = min.x, normal +X, uv = (y, z) */
    RT_RECT_XZ = 1, /* XZRect hitable.rs:284-322  y = min.y, normal +Y, uv = (x, z) */
    RT_RECT_XY = 2  /* XYRect hitable.rs:244-282  z = min.z, normal +Z, uv = (x, y) */
};

/* Instance transforms (hitable.rs:404-520).  A primitive below `Translate { offset, ptr }` /
 * `RotateY::new(ptr, angle)` wrappers carries the index of its INNERMOST wrapper; xf_parent links
 * each wrapper to the next one outwards (RT_NO_XFORM at the top).  hit() applies the chain to the
 * ray from the outside in and fixes the HitRecord from the inside out, exactly like the nested
 * trait objects do (including RotateY's face-normal quirk, hitable.rs:505). */
enum RtXformType {
    RT_XF_TRANSLATE = 0, /* param = offset.x, offset.y, offset.z, 0   hitable.rs:404-418 */
    RT_XF_ROTATE_Y = 1   /* param = sin_theta, cos_theta, angle_degrees, 0   hitable.rs:438-510 */
};
#define RT_NO_XFORM 0xFFFFFFFFu
#define RT_MAX_XFORM_CHAIN 4u /* nesting depth the kernels support */

#define RT_NO_MEDIUM 0xFFFFFFFFu
#define RT_MAX_MEDIA 32u /* the medium's random draw uses counter slot 224 + medium index of its depth block */

#define RT_NO_TEX 0xFFFFFFFFu
#define RT_PERLIN_POINTS 256u /* texture.rs:51 */

/*
 * Flattened structure-of-arrays scene.  Replaces `Vec<Arc<dyn Hitable>>` + the trait
 * objects behind it (hitable.rs:57-62, material.rs, pbr.rs, texture.rs).  Spheres (SURVEY.md
 * §8(a) a4), axis-aligned rectangles / boxes, the Translate / RotateY instance wrappers (§8(f)
 * rank 1) and ConstantMedium (§8(f) rank 2) are on the accelerated path.
 */
typedef struct RtFlatScene {
    /* spheres: hitable.rs:57-62 `Sphere { c, r, mat, name }` in world-list order */
    uint32_t n_spheres;
    const float* sph_cx;   /* [n_spheres] */
    const float* sph_cy;
    const float* sph_cz;
    const float* sph_r;
    const uint32_t* sph_mat; /* [n_spheres] index into the material table */

    /* axis-aligned rectangles: hitable.rs:244-362 `XYRect/XZRect/YZRect { min, max, mat }`; a GBox
     * (hitable.rs:364-383) flattens to its 6 sides in the order of GBox::new.  Rect i has primitive
     * index n_spheres + i: rects come after the spheres in the closest-hit tie order (= world-list
     * order when the scene lists its spheres first, as simple_light_scene does). */
    uint32_t n_rects;
    const uint8_t* rect_axis;  /* [n_rects] RtRectAxis: the constant coordinate */
    const float* rect_min;     /* [3*n_rects] `min` as written in the scene */
    const float* rect_max;     /* [3*n_rects] `max` */
    const uint32_t* rect_mat;  /* [n_rects] */

    /* instance transforms */
    uint32_t n_xforms;
    const uint8_t* xf_type;     /* [n_xforms] RtXformType */
    const float* xf_param;      /* [4*n_xforms] */
    const uint32_t* xf_parent;  /* [n_xforms] next wrapper outwards or RT_NO_XFORM */
    const uint32_t* sph_xform;  /* [n_spheres] innermost wrapper or RT_NO_XFORM; NULL = none */
    const uint32_t* rect_xform; /* [n_rects]   likewise */

    /* homogeneous media: hitable.rs:523-588 `ConstantMedium::new(boundary, density, phase_tex)`.
     * The boundary's primitives are flattened like any others but tagged with the medium that owns
     * them (sph_medium / rect_medium); tagged primitives are NOT hit directly — medium i is primitive
     * n_spheres + n_rects + i and its hit() does the two boundary queries and the one random draw of
     * hitable.rs:540-568.  Its material is the Isotropic phase function (material.rs:99-113). */
    uint32_t n_media;
    const float* med_neg_inv_density; /* [n_media] -1/(k*density), k = how many times the scene's own tree calls the
                                       * medium per visit: 2 for each enclosing BvhNode that holds it as its only
                                       * object (left == right, hitable.rs:188, 236-237), else 1 */
    const uint32_t* med_mat;          /* [n_media] material index (RT_MAT_ISOTROPIC) */
    const uint32_t* sph_medium;       /* [n_spheres] owning medium or RT_NO_MEDIUM; NULL = none */
    const uint32_t* rect_medium;      /* [n_rects] likewise */

    /* materials */
    uint32_t n_materials;
    const uint8_t* mat_type;   /* [n_materials] RtMatType */
    const float* mat_color;    /* [3*n_materials] Metal albedo; unused otherwise */
    const float* mat_p0;       /* [n_materials] see RtMatType comments */
    const float* mat_p1;
    const float* mat_p2;
    const float* mat_p3;
    const uint32_t* mat_tex0;  /* [n_materials] texture index or RT_NO_TEX */
    const uint32_t* mat_tex1;

    /* textures */
    uint32_t n_textures;
    const uint8_t* tex_type;   /* [n_textures] RtTexType */
    const float* tex_color0;   /* [3*n_textures] */
    const float* tex_color1;   /* [3*n_textures] */
    const float* tex_scale;    /* [n_textures] PerlinTex.scale */
    const uint32_t* tex_aux;   /* [n_textures] perlin set index / image index */

    /* Perlin table sets (texture.rs:53-91): rand_vec then perm_x, perm_y, perm_z */
    uint32_t n_perlin;
    const float* perlin_vec;    /* [n_perlin*256*3] xyz interleaved */
    const uint16_t* perlin_perm;/* [n_perlin*3*256] values 0..255 */

    /* images (texture.rs:170-181): Rgb<f32> = u8/255, row 0 = top of the file */
    uint32_t n_images;
    const uint32_t* img_w;      /* [n_images] */
    const uint32_t* img_h;
    const uint64_t* img_offset; /* [n_images] offset in floats into `texels` */
    const float* texels;        /* RGB interleaved */
    uint64_t n_texel_floats;

    /* sky (lib.rs:11) */
    uint32_t sky_type;  /* RtSkyType */
    uint32_t sky_image; /* image index for RT_SKY_ENV (demo_scene.rs:19 ENV_TEX) */
} RtFlatScene;

/* camera.rs:7-10 — the four derived vectors of `Camera` (private fields there). */
typedef struct RtCamera {
    float origin[3];
    float horizontal[3];
    float vertical[3];
    float lower_left_corner[3];
} RtCamera;

/*
 * Render parameters.  Replaces the constants captured by the closure at main.rs:80:
 * nx, ny (main.rs:66-67), samples_per_pixel (main.rs:64), MAX_DEPTH (main.rs:36) and
 * the seed base 95 (main.rs:82).
 */
typedef struct RtParams {
    uint32_t nx, ny;
    uint32_t spp;
    int32_t max_depth;   /* reference MAX_DEPTH = 50: segments with depth 0..=max_depth are traced */
    uint64_t seed;       /* reference: 95 */
    /* Row-interleaved sharding: image row j belongs to shard
     * (j / shard_band) % shard_count.  shard_count <= 1 renders every row. */
    uint32_t shard_band;
    uint32_t shard_count;
    uint32_t shard_id;
    uint32_t spp_slice;  /* samples per pixel traced per wavefront slice; 0 = library default */
    uint32_t flags;      /* RT_FLAG_* */
    uint32_t reserved;
} RtParams;

#define RT_FLAG_NONE 0u
/* Closest hit by the plain list walk (HitableList::hit order, hitable.rs:117-132) instead of the
 * LDS-resident BVH.  Results are identical either way; the flag exists for cross-checking. */
#define RT_FLAG_BRUTE_FORCE 1u
/* Russian roulette, the estimator the reference keeps commented out at main.rs:49-53: after a
 * successful scatter draw one more f32; the path continues with probability
 * threshold = attenuation.max_element() and its attenuation is divided by the threshold.
 * Unbiased, NOT the reference's default estimator (opt-in; paths get ~2x shorter at depth 50). */
#define RT_FLAG_RUSSIAN_ROULETTE 2u
/* Record HIP events around the kernels of every depth of the FIRST slice; read them back with
 * rt_get_depth_timings().  Diagnostic only (adds two event records per depth). */
#define RT_FLAG_TIME_DEPTHS 4u
/* rt_debug_bounce only: run the rays through the ray queue and the SAME kernels rt_render launches for a depth >= 1
 * (the scene's closest-hit kernel with persistent lanes, then the class-sorting shading kernel with its wave64
 * compaction) instead of the unsorted single-kernel test path.  The per-path RNG key is then the one the renderer
 * derives from the slot: ray i gets path_key(seed 0, pixel i, sample 0) and in_key is ignored; out_attenuation,
 * out_o and out_d are filled for surviving rays only (a finished path keeps no ray), out_radiance for finished ones. */
#define RT_FLAG_PRODUCTION_KERNELS 8u

typedef struct RtStats {
    uint64_t n_paths;          /* nx_rows_local * nx * spp                                   */
    uint64_t n_rays;           /* closest-hit queries = ray_color calls passing main.rs:40   */
    uint64_t n_rays_secondary; /* n_rays - n_paths                                           */
    uint64_t n_texture_fetches;/* ImageTex texel fetches (0 when not counted)                */
    uint64_t n_bad_dir;        /* paths dropped where the reference would assert (main.rs:39) */
    double seconds_total;      /* wall time of the render call, host clock                   */
    double seconds_trace;      /* sum of trace+shade kernel time (HIP events, device clock)   */
    double seconds_device;     /* all kernels of the call (HIP events)                       */
    uint64_t bytes_algorithmic;/* 96*n_rays + 24*n_paths (+12*n_texture_fetches), SURVEY §8(d) */
    uint64_t bytes_trace_algorithmic; /* trace kernel share: 48*n_rays + 48*n_secondary + 12*n_paths + 12*n_texture_fetches */
    uint32_t n_trace_launches;
    uint32_t n_slices;
    uint64_t rays_per_depth[64]; /* rays traced at depth d (d < 64) */
} RtStats;

typedef struct RtCtx RtCtx;

/* -- lifecycle ------------------------------------------------------------------------ */
uint32_t rt_abi_version(void);
/* 16 hex digits over the device sources and compiler flags the library was built from ("unknown" for a hand-made build).
 * Measurement records (profiles/) carry it so that a number is never quoted for kernels other than the ones it was taken on. */
const char* rt_build_id(void);
/* Creates a context on HIP device `device_id`.  Replaces main.rs:72-73 (thread pool setup). */
int rt_ctx_create(int device_id, RtCtx** out_ctx);
void rt_ctx_destroy(RtCtx* ctx);
/* Last error text of `ctx` (or of the calling thread's last failed rt_ctx_create if NULL). */
const char* rt_last_error(const RtCtx* ctx);

/* -- scene ---------------------------------------------------------------------------- */
/* Validates and copies the flat scene to HBM (packed device records, Perlin tables,
 * texel pool).  Replaces `let (world, cam) = scene(aspect)` hand-over at main.rs:74 and
 * `SKY_COLOR.set` (demo_scene.rs:38).  May be called again to replace the scene. */
int rt_scene_upload(RtCtx* ctx, const RtFlatScene* scene);

/* Number of image rows owned by a shard, and the mapping local row -> image row. */
uint32_t rt_shard_rows(uint32_t ny, uint32_t shard_band, uint32_t shard_count, uint32_t shard_id);
uint32_t rt_shard_row_to_image_row(uint32_t local_row, uint32_t shard_band, uint32_t shard_count,
                                   uint32_t shard_id);

/* -- render --------------------------------------------------------------------------- */
/*
 * Renders the shard described by `params` and copies the result to host memory.
 * Replaces the pixel loop main.rs:77-108 and the quantisation main.rs:98-105,127.
 *   out_rgb_f32 : [rows_local*nx*3] linear radiance mean (c / spp, BEFORE gamma), local
 *                 row 0 = lowest image row of the shard (reference j order).  May be NULL.
 *   out_rgb8    : [rows_local*nx*3] gamma-2, *255.99 saturating u8, rows in DESCENDING j
 *                 (i.e. flipped like main.rs:127).  May be NULL.
 */
int rt_render(RtCtx* ctx, const RtCamera* cam, const RtParams* params, float* out_rgb_f32,
              uint8_t* out_rgb8, RtStats* stats);

/* Pinned (page-locked) host memory for the two output images.  rt_render() writes a destination allocated here — or any
 * memory the caller has registered with the HIP runtime — by asynchronous copies at PCIe rate behind the last kernel; a
 * pageable destination (a plain Vec / malloc) works too and costs a staged copy (~3 ms instead of ~0.6 ms for a
 * 1920 x 1080 frame).  This is what a host that replaces main.rs:109-128 (the receive loop that fills the ImageBuffer)
 * would allocate its frame in.  NULL on failure. */
void* rt_host_alloc(size_t bytes);
void rt_host_free(void* p);

/*
 * Same, but the f32 framebuffer stays in HBM: `d_out_rgb_f32` is a DEVICE pointer to
 * rows_local*nx*3 floats (e.g. the storage of a tensor that RCCL will gather).  `stream`
 * is a hipStream_t (NULL = the context's own stream); the call returns after the work has
 * been enqueued and `stats` (if not NULL) forces a synchronisation to read the counters.
 * A sphere-only scene synchronises `stream` once more, 0.1 ms into the frame: the number of
 * pixels whose primary-ray candidate list overflowed decides on the host whether depth 0 needs
 * a closest-hit launch (none does in any headline configuration).  The first frame on a stream
 * is preceded by ~0.3 ms of spin kernels that check that the library's second stream runs
 * beside it (two streams on one hardware queue would run the two halves of a slice in turn).
 */
int rt_render_device(RtCtx* ctx, const RtCamera* cam, const RtParams* params,
                     void* d_out_rgb_f32, void* stream, RtStats* stats);

/*
 * Progressive preview.  Replaces the partial saves of the receive loop, main.rs:114-123 (there: every 10
 * columns of pixels; here: after every slice of samples, the unit in which a wavefront renderer finishes
 * work).  After each slice of a following rt_render() the callback gets the running mean of the samples
 * done so far, quantised and flipped exactly like the final image (main.rs:98-105,127): rows_local*nx*3
 * bytes, valid during the call.  The final image is unchanged by the callback; choose the preview cadence
 * with RtParams.spp_slice.  NULL removes the callback.  rt_render_device() never calls it.
 */
typedef void (*RtProgressFn)(void* user, uint32_t spp_done, uint32_t spp_total, const uint8_t* rgb8, uint32_t nx,
                             uint32_t rows);
int rt_set_progress(RtCtx* ctx, RtProgressFn fn, void* user);

/* Per-depth device times of the first slice of the last render that had RT_FLAG_TIME_DEPTHS set:
 * isect_ms[d] = closest-hit kernel, shade_ms[d] = shading kernel, rays[d] = rays traced at depth d
 * in that slice.
 * Returns the number of depths written (<= max_n), or a negative RT_ERR_*. */
int rt_get_depth_timings(RtCtx* ctx, uint32_t max_n, float* isect_ms, float* shade_ms, uint64_t* rays);

/* -- multi-GPU: one process, the GPUs of one node, the framebuffer gather inside the library ---------------------
 * SURVEY.md 8(b)/(e).  The reference's only parallelism is the per-column fan-out over a thread pool with the
 * world shared read-only (main.rs:72-108); here the scene is replicated on every device, device r renders the image
 * rows of the row-interleaved bands (j / band) % n == r (RNG keyed by pixel and sample: the frame does not depend on
 * n), ONE ncclAllGather over RCCL/xGMI brings the equal-sized band buffers together and the first device restores
 * row order.  librccl is opened at rt_multi_create (dlopen); the single-GPU entry points do not depend on it. */
typedef struct RtMulti RtMulti;
/* One RtCtx per listed HIP device + ncclCommInitAll over them.  Replaces main.rs:72-73 for a node.
 * N > 1 over RCCL is UNVERIFIED ON HARDWARE until an 8-GPU run of `bench.py --in-library` has been recorded (no box with
 * more than one GPU has been available to the build); everything around the collective — one host thread per context,
 * padded band buffers, the de-interleave, the statistics — runs for n = 2, 3 on one GPU through rt_multi_create_ex. */
int rt_multi_create(const int* device_ids, int n_devices, RtMulti** out);
/* Same with flags.  RT_MULTI_COPY_GATHER: the all_gather is replaced by device-to-device copies into the same gathered
 * layout and librccl is not opened at all; a device id may then be listed several times (several contexts rendering
 * side by side on one GPU).  The test hook that reaches rt_multi_render's n > 1 code on a one-GPU box; frames are
 * bit-identical to rt_render either way. */
#define RT_MULTI_COPY_GATHER 1u
int rt_multi_create_ex(const int* device_ids, int n_devices, uint32_t flags, RtMulti** out);
void rt_multi_destroy(RtMulti* m);
int rt_multi_device_count(const RtMulti* m);
/* Last error text of `m` (or of the calling thread's last failed rt_multi_create if NULL). */
const char* rt_multi_last_error(const RtMulti* m);
/* rt_scene_upload on every device. */
int rt_multi_scene_upload(RtMulti* m, const RtFlatScene* scene);
/* Renders the WHOLE frame of `params` (nx x ny, spp; shard_count / shard_id are ignored, shard_band = rows per band,
 * 0 = 8) split over the devices and gathers it: out_rgb_f32 [ny*nx*3] row 0 = bottom, out_rgb8 [ny*nx*3] flipped,
 * either may be NULL.  `stats` sums the counters of the devices and takes the maximum of their times. */
int rt_multi_render(RtMulti* m, const RtCamera* cam, const RtParams* params, float* out_rgb_f32, uint8_t* out_rgb8,
                    RtStats* stats);
/* The de-interleave step on its own: `d_gathered` is a DEVICE buffer of n_shards band buffers, each
 * max_r rt_shard_rows(ny, band, n_shards, r) rows of nx*3 floats (what the all_gather delivers); writes the frame in
 * image row order to d_out_rgb_f32 [ny*nx*3] and / or the quantised, flipped image to d_out_rgb8 (device pointers,
 * either may be NULL).  `stream` as in rt_render_device. */
int rt_deinterleave_bands(RtCtx* ctx, const void* d_gathered, uint32_t nx, uint32_t ny, uint32_t band, uint32_t n_shards,
                          void* d_out_rgb_f32, void* d_out_rgb8, void* stream);

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
    RT_OPT__COUNT = 13
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
#endif /* RTOW_MI355X_H */
