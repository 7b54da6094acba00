"""ctypes declarations of the two C ABIs (include/rtow_mi355x.h + its test hooks in include/rtow_mi355x_debug.h, include/rtow_host.h).

Plumbing only: struct layouts, library loading and argument types.  The GPU library is
mandatory for rendering; `load_gpu_library()` raises if it is missing — there is no CPU
fallback in the product path.
"""
import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RTOW_GPU_LIB selects an alternative build of the SAME sources (A/B experiments, scripts/ only)
GPU_LIB_PATH = os.environ.get("RTOW_GPU_LIB") or os.path.join(_PKG_DIR, "librtow_mi355x.so")
HOST_LIB_PATH = os.path.join(_PKG_DIR, "librtow_host.so")

RT_NO_TEX = 0xFFFFFFFF
FLAG_BRUTE_FORCE = 1
FLAG_RUSSIAN_ROULETTE = 2
FLAG_TIME_DEPTHS = 4
FLAG_PRODUCTION_KERNELS = 8  # rt_debug_bounce through the ray queue and the kernels rt_render launches
RTH_INVALID = 0xFFFFFFFF

# enum RtMatType
(MAT_EMISSION, MAT_DIFFUSE, MAT_LAMBERT, MAT_METAL, MAT_DIELECTRIC, MAT_ISOTROPIC, MAT_OREN_NAYAR,
 MAT_BURLEY_DIFFUSE, MAT_ROUGH_PLASTIC, MAT_DISNEY_DIFFUSE, MAT_DISNEY_METAL, MAT_DISNEY_SHEEN,
 MAT_DISNEY_CLEARCOAT) = range(13)
# enum RtTexType
TEX_CONSTANT, TEX_CHECKER, TEX_PERLIN, TEX_IMAGE = range(4)
# enum RtRectAxis
RECT_YZ, RECT_XZ, RECT_XY = range(3)
# enum RtXformType
XF_TRANSLATE, XF_ROTATE_Y = range(2)
NO_XFORM = 0xFFFFFFFF
# enum RtSkyType
SKY_GRADIENT, SKY_BLACK, SKY_ENV = range(3)

_f = C.POINTER(C.c_float)
_u8 = C.POINTER(C.c_uint8)
_u16 = C.POINTER(C.c_uint16)
_u32 = C.POINTER(C.c_uint32)
_u64 = C.POINTER(C.c_uint64)


class RtFlatScene(C.Structure):
    _fields_ = [
        ("n_spheres", C.c_uint32),
        ("sph_cx", _f), ("sph_cy", _f), ("sph_cz", _f), ("sph_r", _f), ("sph_mat", _u32),
        ("n_rects", C.c_uint32),
        ("rect_axis", _u8), ("rect_min", _f), ("rect_max", _f), ("rect_mat", _u32),
        ("n_xforms", C.c_uint32),
        ("xf_type", _u8), ("xf_param", _f), ("xf_parent", _u32), ("sph_xform", _u32), ("rect_xform", _u32),
        ("n_media", C.c_uint32),
        ("med_neg_inv_density", _f), ("med_mat", _u32), ("sph_medium", _u32), ("rect_medium", _u32), ("med_xform", _u32),
        ("n_materials", C.c_uint32),
        ("mat_type", _u8), ("mat_color", _f), ("mat_p0", _f), ("mat_p1", _f), ("mat_p2", _f), ("mat_p3", _f),
        ("mat_tex0", _u32), ("mat_tex1", _u32),
        ("n_textures", C.c_uint32),
        ("tex_type", _u8), ("tex_color0", _f), ("tex_color1", _f), ("tex_scale", _f), ("tex_aux", _u32),
        ("n_perlin", C.c_uint32),
        ("perlin_vec", _f), ("perlin_perm", _u16),
        ("n_images", C.c_uint32),
        ("img_w", _u32), ("img_h", _u32), ("img_offset", _u64), ("texels", _f), ("n_texel_floats", C.c_uint64),
        ("sky_type", C.c_uint32), ("sky_image", C.c_uint32),
    ]


class RtCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("lower_left_corner", C.c_float * 3)]


class RtParams(C.Structure):
    _fields_ = [("nx", C.c_uint32), ("ny", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_int32),
                ("seed", C.c_uint64), ("shard_band", C.c_uint32), ("shard_count", C.c_uint32),
                ("shard_id", C.c_uint32), ("spp_slice", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class RtStats(C.Structure):
    _fields_ = [("n_paths", C.c_uint64), ("n_rays", C.c_uint64), ("n_rays_secondary", C.c_uint64),
                ("n_texture_fetches", C.c_uint64), ("n_bad_dir", C.c_uint64), ("seconds_total", C.c_double),
                ("seconds_trace", C.c_double), ("seconds_device", C.c_double), ("bytes_algorithmic", C.c_uint64),
                ("bytes_trace_algorithmic", C.c_uint64), ("n_trace_launches", C.c_uint32), ("n_slices", C.c_uint32),
                ("rays_per_depth", C.c_uint64 * 64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "rays_per_depth"}
        d["rays_per_depth"] = list(self.rays_per_depth)
        return d


class RtBounceIO(C.Structure):
    _fields_ = [("n", C.c_uint32), ("depth", C.c_uint32), ("in_o", _f), ("in_d", _f), ("in_key", _u32),
                ("out_hit", C.POINTER(C.c_int32)), ("out_t", _f), ("out_radiance", _f), ("out_attenuation", _f),
                ("out_o", _f), ("out_d", _f), ("out_alive", _u8), ("flags", C.c_uint32)]


# void (*RtProgressFn)(void* user, uint32_t spp_done, uint32_t spp_total, const uint8_t* rgb8, uint32_t nx, uint32_t rows)
RtProgressFn = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32)

MULTI_COPY_GATHER = 1  # RT_MULTI_COPY_GATHER (rt_multi_create_ex)
# enum RtDebugOption (rt_debug_set_option): per context, every setting renders the same bits
(OPT_TREE_PLACEMENT, OPT_PRIMARY_LISTS, OPT_PIXEL_ORDER, OPT_TEXEL_POOL, OPT_GRID, OPT_GRID_CELL, OPT_CHAINS,
 OPT_GENERAL_KERNELS, OPT_GENERAL_LDS, OPT_QUEUE_SHARDS, OPT_ISECT_WORKGROUPS, OPT_MATERIALISE_PRIMARIES, OPT_MEDIUM_SEARCH, OPT_POOL_CHUNK_DELAY_US) = range(14)
OPT_NAMES = {"tree_placement": OPT_TREE_PLACEMENT, "primary_lists": OPT_PRIMARY_LISTS, "pixel_order": OPT_PIXEL_ORDER,
             "texel_pool": OPT_TEXEL_POOL, "grid": OPT_GRID, "grid_cell": OPT_GRID_CELL, "chains": OPT_CHAINS,
             "general_kernels": OPT_GENERAL_KERNELS, "general_lds": OPT_GENERAL_LDS, "queue_shards": OPT_QUEUE_SHARDS,
             "isect_workgroups": OPT_ISECT_WORKGROUPS, "materialise_primaries": OPT_MATERIALISE_PRIMARIES,
             "medium_search": OPT_MEDIUM_SEARCH, "pool_chunk_delay_us": OPT_POOL_CHUNK_DELAY_US}


class RtSceneInfo(C.Structure):
    _fields_ = [("n_entries", C.c_uint32), ("n_tree_nodes", C.c_uint32), ("tree_depth", C.c_uint32), ("tree_in_lds", C.c_uint32),
                ("general_kernels", C.c_uint32), ("closest_hit_lds_bytes", C.c_uint32), ("grid", C.c_uint32),
                ("grid_cells", C.c_uint32 * 3), ("grid_refs", C.c_uint32), ("grid_always", C.c_uint32),
                ("grid_lds_bytes", C.c_uint32), ("grid_cell_size", C.c_float * 3), ("general_tables_in_lds", C.c_uint32), ("nest", C.c_uint32)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("grid_cells", "grid_cell_size")}
        d["grid_cells"], d["grid_cell_size"] = list(self.grid_cells), [float(x) for x in self.grid_cell_size]
        return d


EXPECTED_ABI = 11  # RT_ABI_VERSION the struct layouts and prototypes below were written for
GPU_SYMBOLS = ["rt_abi_version", "rt_build_id", "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_scene_upload",
               "rt_shard_rows", "rt_shard_row_to_image_row", "rt_prepare", "rt_render", "rt_render_device", "rt_debug_bounce", "rt_debug_arithmetic",
               "rt_get_depth_timings", "rt_set_progress", "rt_host_alloc", "rt_host_free", "rt_debug_set_option", "rt_debug_get_option",
               "rt_debug_scene_info", "rt_debug_grid_build", "rt_debug_render_parts", "rt_multi_create", "rt_multi_create_ex", "rt_multi_destroy", "rt_multi_device_count",
               "rt_multi_last_error", "rt_multi_scene_upload", "rt_multi_render", "rt_deinterleave_bands"]
HOST_SYMBOLS = ["rth_last_error", "rth_register_image", "rth_rng_reseed", "rth_scene_build", "rth_scene_new",
                "rth_tex_constant", "rth_tex_checker", "rth_tex_perlin", "rth_tex_image", "rth_material",
                "rth_sphere", "rth_rect", "rth_gbox", "rth_translate", "rth_rotate_y", "rth_constant_medium", "rth_hitable_bbox", "rth_set_sky", "rth_set_camera", "rth_scene_finish", "rth_scene_flat",
                "rth_scene_camera", "rth_scene_sphere_name", "rth_scene_free", "rth_png_write", "rth_output_file_name"]

_gpu_lib = None
_host_lib = None


class GpuLibraryMissing(RuntimeError):
    pass


def load_gpu_library():
    """Loads librtow_mi355x.so (hand-written HIP kernels + C-ABI).  Raises if it is not built."""
    global _gpu_lib
    if _gpu_lib is not None:
        return _gpu_lib
    if not os.path.exists(GPU_LIB_PATH):
        raise GpuLibraryMissing(
            f"{GPU_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the render path.")
    lib = C.CDLL(GPU_LIB_PATH)
    vp = C.c_void_p
    lib.rt_abi_version.restype = C.c_uint32
    if lib.rt_abi_version() != EXPECTED_ABI:  # a stale .so would be read with the wrong struct layouts
        raise GpuLibraryMissing(f"{GPU_LIB_PATH} has ABI version {lib.rt_abi_version()}, this package expects {EXPECTED_ABI}: "
                                "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    lib.rt_build_id.restype = C.c_char_p
    lib.rt_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rt_ctx_create.restype = C.c_int
    lib.rt_ctx_destroy.argtypes = [vp]
    lib.rt_ctx_destroy.restype = None
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_scene_upload.argtypes = [vp, C.POINTER(RtFlatScene)]
    lib.rt_scene_upload.restype = C.c_int
    lib.rt_shard_rows.argtypes = [C.c_uint32] * 4
    lib.rt_shard_rows.restype = C.c_uint32
    lib.rt_shard_row_to_image_row.argtypes = [C.c_uint32] * 4
    lib.rt_shard_row_to_image_row.restype = C.c_uint32
    lib.rt_prepare.argtypes = [vp, C.POINTER(RtParams)]
    lib.rt_prepare.restype = C.c_int
    lib.rt_render.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtParams), _f, _u8, C.POINTER(RtStats)]
    lib.rt_render.restype = C.c_int
    lib.rt_render_device.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtParams), vp, vp, C.POINTER(RtStats)]
    lib.rt_render_device.restype = C.c_int
    lib.rt_get_depth_timings.argtypes = [vp, C.c_uint32, _f, _f, _u64]
    lib.rt_get_depth_timings.restype = C.c_int
    lib.rt_debug_bounce.argtypes = [vp, C.POINTER(RtBounceIO)]
    lib.rt_debug_bounce.restype = C.c_int
    lib.rt_debug_arithmetic.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.rt_debug_arithmetic.restype = C.c_int
    lib.rt_set_progress.argtypes = [vp, RtProgressFn, vp]
    lib.rt_set_progress.restype = C.c_int
    lib.rt_host_alloc.argtypes = [C.c_size_t]
    lib.rt_host_alloc.restype = vp
    lib.rt_host_free.argtypes = [vp]
    lib.rt_host_free.restype = None
    lib.rt_debug_set_option.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.rt_debug_set_option.restype = C.c_int
    lib.rt_debug_get_option.argtypes = [vp, C.c_uint32, _u32]
    lib.rt_debug_get_option.restype = C.c_int
    lib.rt_debug_scene_info.argtypes = [vp, C.POINTER(RtSceneInfo)]
    lib.rt_debug_scene_info.restype = C.c_int
    lib.rt_debug_render_parts.argtypes = [vp, C.c_char_p, C.c_uint32]
    lib.rt_debug_render_parts.restype = C.c_int
    lib.rt_debug_grid_build.argtypes = [C.POINTER(RtFlatScene), C.c_uint32, C.c_uint32, C.c_float * 8, C.c_uint32 * 3, _u32, _u32, _u16, _u32,
                                        C.c_uint32 * 4, _u32]
    lib.rt_debug_grid_build.restype = C.c_int
    lib.rt_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.rt_multi_create.restype = C.c_int
    lib.rt_multi_create_ex.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_uint32, C.POINTER(vp)]
    lib.rt_multi_create_ex.restype = C.c_int
    lib.rt_multi_destroy.argtypes = [vp]
    lib.rt_multi_destroy.restype = None
    lib.rt_multi_device_count.argtypes = [vp]
    lib.rt_multi_device_count.restype = C.c_int
    lib.rt_multi_last_error.argtypes = [vp]
    lib.rt_multi_last_error.restype = C.c_char_p
    lib.rt_multi_scene_upload.argtypes = [vp, C.POINTER(RtFlatScene)]
    lib.rt_multi_scene_upload.restype = C.c_int
    lib.rt_multi_render.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtParams), _f, _u8, C.POINTER(RtStats)]
    lib.rt_multi_render.restype = C.c_int
    lib.rt_deinterleave_bands.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    lib.rt_deinterleave_bands.restype = C.c_int
    _gpu_lib = lib
    return lib


def load_host_library():
    """Loads librtow_host.so (C++ mirror of the reference's construction API; no GPU code)."""
    global _host_lib
    if _host_lib is not None:
        return _host_lib
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
    lib = C.CDLL(HOST_LIB_PATH)
    vp = C.c_void_p
    f3 = C.c_float * 3
    lib.rth_last_error.restype = C.c_char_p
    lib.rth_register_image.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _f]
    lib.rth_register_image.restype = C.c_int
    lib.rth_rng_reseed.argtypes = [C.c_uint64]
    lib.rth_rng_reseed.restype = None
    lib.rth_scene_build.argtypes = [C.c_char_p, C.c_float, C.POINTER(vp)]
    lib.rth_scene_build.restype = C.c_int
    lib.rth_scene_new.argtypes = [C.POINTER(vp)]
    lib.rth_scene_new.restype = C.c_int
    lib.rth_tex_constant.argtypes = [vp, f3]
    lib.rth_tex_constant.restype = C.c_uint32
    lib.rth_tex_checker.argtypes = [vp, f3, f3]
    lib.rth_tex_checker.restype = C.c_uint32
    lib.rth_tex_perlin.argtypes = [vp, C.c_float]
    lib.rth_tex_perlin.restype = C.c_uint32
    lib.rth_tex_image.argtypes = [vp, C.c_char_p]
    lib.rth_tex_image.restype = C.c_uint32
    lib.rth_material.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, f3, C.c_float * 4]
    lib.rth_material.restype = C.c_uint32
    lib.rth_sphere.argtypes = [vp, f3, C.c_float, C.c_uint32, C.c_char_p]
    lib.rth_sphere.restype = C.c_uint32
    lib.rth_rect.argtypes = [vp, C.c_uint32, f3, f3, C.c_uint32]
    lib.rth_rect.restype = C.c_uint32
    lib.rth_gbox.argtypes = [vp, f3, f3, C.c_uint32]
    lib.rth_gbox.restype = C.c_uint32
    lib.rth_translate.argtypes = [vp, C.c_uint32, f3]
    lib.rth_translate.restype = C.c_uint32
    lib.rth_rotate_y.argtypes = [vp, C.c_uint32, C.c_float]
    lib.rth_rotate_y.restype = C.c_uint32
    lib.rth_constant_medium.argtypes = [vp, C.c_uint32, C.c_float, C.c_uint32]
    lib.rth_constant_medium.restype = C.c_uint32
    lib.rth_hitable_bbox.argtypes = [vp, C.c_uint32, C.c_float * 6]
    lib.rth_hitable_bbox.restype = C.c_int
    lib.rth_set_sky.argtypes = [vp, C.c_uint32, C.c_char_p]
    lib.rth_set_sky.restype = C.c_int
    lib.rth_set_camera.argtypes = [vp, f3, f3, f3, C.c_float, C.c_float]
    lib.rth_set_camera.restype = C.c_int
    lib.rth_scene_finish.argtypes = [vp, C.c_int]
    lib.rth_scene_finish.restype = C.c_int
    lib.rth_scene_flat.argtypes = [vp]
    lib.rth_scene_flat.restype = C.POINTER(RtFlatScene)
    lib.rth_scene_camera.argtypes = [vp, C.POINTER(RtCamera)]
    lib.rth_scene_camera.restype = C.c_int
    lib.rth_scene_sphere_name.argtypes = [vp, C.c_uint32]
    lib.rth_scene_sphere_name.restype = C.c_char_p
    lib.rth_scene_free.argtypes = [vp]
    lib.rth_png_write.argtypes = [C.c_char_p, _u8, C.c_uint32, C.c_uint32]
    lib.rth_png_write.restype = C.c_int
    lib.rth_output_file_name.argtypes = [C.c_int64, C.c_char_p, C.c_uint32]
    lib.rth_output_file_name.restype = C.c_int
    lib.rth_scene_free.restype = None
    _host_lib = lib
    return lib
