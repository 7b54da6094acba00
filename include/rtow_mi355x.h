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

#define RT_ABI_VERSION 11u /* 11: RtFlatScene::med_xform (wrappers around a medium); wrapper chains of any depth, up to RT_MAX_MEDIA = 65 535 media.
                             * 10: rt_prepare; the test hooks moved to rtow_mi355x_debug.h */

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
/* (wrappers nest to any depth, as the trait objects do: chains of up to 4 are walked in registers, longer ones through a list
 * that rt_scene_upload builds) */

#define RT_NO_MEDIUM 0xFFFFFFFFu
#define RT_MAX_MEDIA 65535u /* a depth's block of free-path counters holds this many; the 32nd and later media of a scene are
                             * tested together by a ray that reaches any of them, so scenes with many media are slower */

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
    const uint32_t* med_xform;        /* [n_media] the innermost wrapper AROUND medium i — `Translate { ptr: ConstantMedium }`,
                                       * hitable.rs:409-416: the medium's hit() then sees the moved ray (its length, its
                                       * rec.p = r.at(t)) and the wrapper fixes the record — or RT_NO_XFORM; NULL = none.
                                       * The chains of the boundary's primitives continue through this wrapper (they end
                                       * at the world like any other); wrappers INSIDE the boundary need no entry here. */

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
/* Flag bits 4u and 8u are diagnostic hooks, declared in rtow_mi355x_debug.h (RT_FLAG_TIME_DEPTHS, RT_FLAG_PRODUCTION_KERNELS). */

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

/*
 * Optional: tells the context which frame it will be asked for — nx, ny, spp, spp_slice and the shard fields of `params` are
 * read, no scene is needed — so that the device memory for its work buffers (100 B per ray of a slice: 53 GB for 1920 x 1080 x
 * 256 spp in one slice) is requested NOW, by a helper thread, while the host builds its scene.  The reference knows its frame
 * before it builds the world (main.rs:64-67 against main.rs:74), and renders one frame per process: without the hint the first
 * rt_render starts the same request itself and renders its first slices in what has arrived so far.  Returns at once.
 */
int rt_prepare(RtCtx* ctx, const RtParams* params);

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

/* -- multi-GPU: one process, the GPUs of one node, the framebuffer gather inside the library ---------------------
 * SURVEY.md 8(b)/(e).  The reference's only parallelism is the per-column fan-out over a thread pool with the
 * world shared read-only (main.rs:72-108); here the scene is replicated on every device, device r renders the image
 * rows of the row-interleaved bands (j / band) % n == r (RNG keyed by pixel and sample: the frame does not depend on
 * n), ONE gather over RCCL/xGMI — a group of ncclSend / ncclRecv to the first device — brings the equal-sized band buffers
 * together and the first device restores row order.  librccl is opened at rt_multi_create (dlopen); the single-GPU entry points do not depend on it. */
typedef struct RtMulti RtMulti;
/* One RtCtx per listed HIP device + ncclCommInitAll over them.  Replaces main.rs:72-73 for a node.
 * N > 1 over RCCL is UNVERIFIED ON HARDWARE until an 8-GPU run of `bench.py --in-library` has been recorded (no box with
 * more than one GPU has been available to the build); everything around the collective — one host thread per context,
 * padded band buffers, the de-interleave, the statistics — runs for n = 2, 3 on one GPU through rt_multi_create_ex. */
int rt_multi_create(const int* device_ids, int n_devices, RtMulti** out);
/* Same with flags.  RT_MULTI_COPY_GATHER: the RCCL gather is replaced by device-to-device copies into the same gathered
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
 * max_r rt_shard_rows(ny, band, n_shards, r) rows of nx*3 floats (what the gather delivers); writes the frame in
 * image row order to d_out_rgb_f32 [ny*nx*3] and / or the quantised, flipped image to d_out_rgb8 (device pointers,
 * either may be NULL).  `stream` as in rt_render_device. */
int rt_deinterleave_bands(RtCtx* ctx, const void* d_gathered, uint32_t nx, uint32_t ny, uint32_t band, uint32_t n_shards,
                          void* d_out_rgb_f32, void* d_out_rgb8, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RTOW_MI355X_H */
