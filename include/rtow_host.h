/*
 * rtow_host.h — C handle API over the host-side C++ mirror of the reference's construction
 * surface (ray_tracing_in_one_weekend_amd/host/rtow.hpp).  It exists so that non-C++ hosts
 * (the Python/ctypes test and bench harness here; a Rust `extern "C"` block in the reference)
 * can build scenes with the reference's constructors and obtain the RtFlatScene/RtCamera that
 * rt_scene_upload()/rt_render() take (include/rtow_mi355x.h).  No GPU code in this library.
 *
 * Reference interfaces mirrored (paths relative to /root/reference/src):
 *   rth_scene_build     demo_scene.rs:37,229  scene fns `fn(aspect_ratio) -> (HitableList, Camera)`
 *   rth_tex_*           texture.rs:15-17,32-37,156-161,175-181  ConstantTex / CheckerTex::new /
 *                       PerlinTex::new / ImageTex::new
 *   rth_material        material.rs + pbr.rs public-field structs (tag = RtMatType)
 *   rth_sphere          hitable.rs:57-62  Sphere { c, r, mat, name }
 *   rth_set_sky         lib.rs:11 SKY_COLOR.set(..), demo_scene.rs:19 ENV_TEX
 *   rth_set_camera      camera.rs:14-20  Camera::new(lookfrom, lookat, vup, vfov, aspect_ratio)
 *   rth_rng_reseed      lib.rs:8  thread-local SmallRng::seed_from_u64(1995)
 * All functions return 0 / a valid handle on success; on failure they return a negative code /
 * RTH_INVALID and rth_last_error() holds the text (no exception crosses the boundary).
 */
#ifndef RTOW_HOST_H
#define RTOW_HOST_H
#include "rtow_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RTH_INVALID 0xFFFFFFFFu

typedef struct RthScene RthScene;

const char* rth_last_error(void);

/* Registers decoded pixels (Rgb<f32> = u8/255, row 0 = top) under the path the scene code
 * passes to ImageTex::new, e.g. "res/earthmap.jpg" (texture.rs:175-181). */
int rth_register_image(const char* path, uint32_t w, uint32_t h, const float* rgb);

/* Re-seeds the thread-local RNG (lib.rs:8; default 1995) that PerlinTex::new draws from. */
void rth_rng_reseed(uint64_t seed);

/* Runs a named scene function: "sphere_scene", "test_sphere", "simple_light_scene",
 * "cornell_box", "final_scene", "earth_env_scene", "pbr_sweep_scene"; flattens world + sky + camera.  The thread RNG is reset to its
 * fresh-process state (seed 1995) first, so repeated builds give identical Perlin tables. */
int rth_scene_build(const char* name, float aspect_ratio, RthScene** out);

/* Piecewise construction with the reference's constructors. */
int rth_scene_new(RthScene** out);
uint32_t rth_tex_constant(RthScene* s, const float col[3]);
uint32_t rth_tex_checker(RthScene* s, const float odd[3], const float even[3]);
uint32_t rth_tex_perlin(RthScene* s, float scale);
uint32_t rth_tex_image(RthScene* s, const char* path);
/* type = RtMatType; tex0/tex1 = handles from rth_tex_* or RTH_INVALID; color = Metal albedo;
 * p = the scalar fields in the order documented at RtMatType. */
uint32_t rth_material(RthScene* s, uint32_t type, uint32_t tex0, uint32_t tex1, const float color[3], const float p[4]);
uint32_t rth_sphere(RthScene* s, const float c[3], float r, uint32_t material, const char* name);
/* hitable.rs:244-362 XYRect/XZRect/YZRect { min, max, mat } (axis = RtRectAxis) and hitable.rs:364-383 GBox::new */
uint32_t rth_rect(RthScene* s, uint32_t axis, const float mn[3], const float mx[3], uint32_t material);
uint32_t rth_gbox(RthScene* s, const float mn[3], const float mx[3], uint32_t material);
/* hitable.rs:404-520: wrap the world entry `hitable` in Translate { offset, ptr } / RotateY::new(ptr, angle);
 * the handle stays valid and now denotes the wrapper (wrappers nest, outermost applied last) */
uint32_t rth_translate(RthScene* s, uint32_t hitable, const float offset[3]);
uint32_t rth_rotate_y(RthScene* s, uint32_t hitable, float angle_degrees);
/* hitable.rs:529-533: turn the world entry `hitable` into ConstantMedium::new(hitable, density, phase_tex) */
uint32_t rth_constant_medium(RthScene* s, uint32_t hitable, float density, uint32_t phase_tex);
/* Hitable::bbox (hitable.rs:52) of a world entry, wrappers included: out = {min.xyz, max.xyz}; returns the trait
 * method's bool (1/0), < 0 on a bad handle.  BvhNode::new sorts by it (hitable.rs:163-174). */
int rth_hitable_bbox(RthScene* s, uint32_t hitable, float out[6]);
/* sky = RtSkyType; env_path only for RT_SKY_ENV */
int rth_set_sky(RthScene* s, uint32_t sky, const char* env_path);
int rth_set_camera(RthScene* s, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                   float aspect_ratio);
/* Wraps everything added so far in a BvhNode like build_bvh (demo_scene.rs:223-227) and flattens. */
int rth_scene_finish(RthScene* s, int use_bvh);

const RtFlatScene* rth_scene_flat(const RthScene* s);
int rth_scene_camera(const RthScene* s, RtCamera* out);
const char* rth_scene_sphere_name(const RthScene* s, uint32_t index);
void rth_scene_free(RthScene* s);

/* ---- driver conveniences (main.rs:109-128) -------------------------------------------------------------- */

/* Saves rows*nx RGB8 pixels (row 0 = top, i.e. the layout rt_render's out_rgb8 already has after the flip of
 * main.rs:127) as an 8-bit RGB PNG, what `img.save(file_name)` writes at main.rs:121,128.  The file appears
 * atomically (written to <path>.part, then renamed), so it can be polled as a preview. */
int rth_png_write(const char* path, const uint8_t* rgb8, uint32_t nx, uint32_t ny);

/* The reference's output name (main.rs:110-112): local time as RFC 3339 with ':' replaced by '-', cut before the
 * fractional seconds, + ".png", e.g. "2024-05-17T21-03-44.png".  unix_seconds < 0 = now.  buf_len >= 24. */
int rth_output_file_name(int64_t unix_seconds, char* buf, uint32_t buf_len);

#ifdef __cplusplus
}
#endif
#endif
