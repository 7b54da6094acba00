"""Host-side Python surface over the two C ABIs.

`Scene` wraps the C++ mirror of the reference's constructors (librtow_host.so);
`Renderer` wraps the GPU library (librtow_mi355x.so).  Nothing here computes pixels: every
render call goes through the HIP kernels, and a missing GPU library raises.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import RtBounceIO, RtCamera, RtFlatScene, RtParams, RtStats


class RtError(RuntimeError):
    pass


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Scene:
    """A flattened scene + camera built with the reference's constructors
    (demo_scene.rs scene fns, or piecewise: textures -> materials -> spheres -> camera)."""

    def __init__(self, handle):
        self._lib = _ffi.load_host_library()
        self._h = handle

    # -- named scene functions (demo_scene.rs:37,229 + the two build-authored ones) --------
    @classmethod
    def build(cls, name, aspect_ratio):
        lib = _ffi.load_host_library()
        h = C.c_void_p()
        rc = lib.rth_scene_build(name.encode(), C.c_float(aspect_ratio), C.byref(h))
        if rc != 0:
            raise RtError(lib.rth_last_error().decode())
        return cls(h)

    # -- piecewise construction ----------------------------------------------------------------
    @classmethod
    def new(cls):
        lib = _ffi.load_host_library()
        h = C.c_void_p()
        if lib.rth_scene_new(C.byref(h)) != 0:
            raise RtError(lib.rth_last_error().decode())
        return cls(h)

    def _check(self, handle):
        if handle == _ffi.RTH_INVALID:
            raise RtError(self._lib.rth_last_error().decode())
        return handle

    def constant_tex(self, col):
        return self._check(self._lib.rth_tex_constant(self._h, _f3(col)))

    def checker_tex(self, odd, even):
        return self._check(self._lib.rth_tex_checker(self._h, _f3(odd), _f3(even)))

    def perlin_tex(self, scale):
        return self._check(self._lib.rth_tex_perlin(self._h, C.c_float(scale)))

    def image_tex(self, path):
        return self._check(self._lib.rth_tex_image(self._h, path.encode()))

    def material(self, mat_type, tex0=_ffi.RTH_INVALID, tex1=_ffi.RTH_INVALID, color=(0, 0, 0), p=(0, 0, 0, 0)):
        p = list(p) + [0.0] * (4 - len(p))
        return self._check(self._lib.rth_material(self._h, mat_type, tex0, tex1, _f3(color),
                                                  (C.c_float * 4)(*[float(x) for x in p])))

    def sphere(self, c, r, material, name=""):
        return self._check(self._lib.rth_sphere(self._h, _f3(c), C.c_float(r), material, name.encode()))

    def rect(self, axis, mn, mx, material):
        return self._check(self._lib.rth_rect(self._h, axis, _f3(mn), _f3(mx), material))

    def gbox(self, mn, mx, material):
        return self._check(self._lib.rth_gbox(self._h, _f3(mn), _f3(mx), material))

    def translate(self, hitable, offset):
        return self._check(self._lib.rth_translate(self._h, hitable, _f3(offset)))

    def rotate_y(self, hitable, angle_degrees):
        return self._check(self._lib.rth_rotate_y(self._h, hitable, C.c_float(angle_degrees)))

    def constant_medium(self, hitable, density, phase_tex):
        return self._check(self._lib.rth_constant_medium(self._h, hitable, C.c_float(density), phase_tex))

    def bbox(self, hitable):
        """Hitable::bbox of a world entry (hitable.rs:52): (has_box, [min.xyz, max.xyz])."""
        out = (C.c_float * 6)()
        rc = self._lib.rth_hitable_bbox(self._h, hitable, out)
        if rc < 0:
            raise RtError(self._lib.rth_last_error().decode())
        return bool(rc), np.array(out, dtype=np.float32)

    def set_sky(self, sky, env_path=None):
        if self._lib.rth_set_sky(self._h, sky, env_path.encode() if env_path else None) != 0:
            raise RtError(self._lib.rth_last_error().decode())

    def set_camera(self, lookfrom, lookat, vup, vfov, aspect_ratio):
        if self._lib.rth_set_camera(self._h, _f3(lookfrom), _f3(lookat), _f3(vup), C.c_float(vfov),
                                    C.c_float(aspect_ratio)) != 0:
            raise RtError(self._lib.rth_last_error().decode())

    def finish(self, use_bvh=True):
        if self._lib.rth_scene_finish(self._h, 1 if use_bvh else 0) != 0:
            raise RtError(self._lib.rth_last_error().decode())
        return self

    # -- accessors ----------------------------------------------------------------------------------
    @property
    def flat(self):
        p = self._lib.rth_scene_flat(self._h)
        if not p:
            raise RtError("scene not finished")
        return p.contents

    @property
    def flat_ptr(self):
        p = self._lib.rth_scene_flat(self._h)
        if not p:
            raise RtError("scene not finished")
        return p

    @property
    def camera(self):
        cam = RtCamera()
        if self._lib.rth_scene_camera(self._h, C.byref(cam)) != 0:
            raise RtError("scene not finished")
        return cam

    def sphere_name(self, i):
        return self._lib.rth_scene_sphere_name(self._h, i).decode()

    def arrays(self):
        """numpy copies of the flat arrays (for tests and inspection)."""
        fs = self.flat

        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)
        ns, nm, nt = fs.n_spheres, fs.n_materials, fs.n_textures
        return {
            "sph_cx": arr(fs.sph_cx, ns, np.float32), "sph_cy": arr(fs.sph_cy, ns, np.float32),
            "sph_cz": arr(fs.sph_cz, ns, np.float32), "sph_r": arr(fs.sph_r, ns, np.float32),
            "sph_mat": arr(fs.sph_mat, ns, np.uint32),
            "rect_axis": arr(fs.rect_axis, fs.n_rects, np.uint8), "rect_min": arr(fs.rect_min, 3 * fs.n_rects, np.float32),
            "rect_max": arr(fs.rect_max, 3 * fs.n_rects, np.float32), "rect_mat": arr(fs.rect_mat, fs.n_rects, np.uint32),
            "xf_type": arr(fs.xf_type, fs.n_xforms, np.uint8), "xf_param": arr(fs.xf_param, 4 * fs.n_xforms, np.float32),
            "xf_parent": arr(fs.xf_parent, fs.n_xforms, np.uint32), "sph_xform": arr(fs.sph_xform, ns, np.uint32),
            "rect_xform": arr(fs.rect_xform, fs.n_rects, np.uint32),
            "med_neg_inv_density": arr(fs.med_neg_inv_density, fs.n_media, np.float32), "med_mat": arr(fs.med_mat, fs.n_media, np.uint32),
            "sph_medium": arr(fs.sph_medium, ns, np.uint32), "rect_medium": arr(fs.rect_medium, fs.n_rects, np.uint32),
            "med_xform": arr(fs.med_xform, fs.n_media, np.uint32),
            "mat_type": arr(fs.mat_type, nm, np.uint8), "mat_color": arr(fs.mat_color, 3 * nm, np.float32),
            "mat_p0": arr(fs.mat_p0, nm, np.float32), "mat_p1": arr(fs.mat_p1, nm, np.float32),
            "mat_p2": arr(fs.mat_p2, nm, np.float32), "mat_p3": arr(fs.mat_p3, nm, np.float32),
            "mat_tex0": arr(fs.mat_tex0, nm, np.uint32), "mat_tex1": arr(fs.mat_tex1, nm, np.uint32),
            "tex_type": arr(fs.tex_type, nt, np.uint8), "tex_color0": arr(fs.tex_color0, 3 * nt, np.float32),
            "tex_color1": arr(fs.tex_color1, 3 * nt, np.float32), "tex_scale": arr(fs.tex_scale, nt, np.float32),
            "tex_aux": arr(fs.tex_aux, nt, np.uint32),
            "perlin_vec": arr(fs.perlin_vec, fs.n_perlin * 768, np.float32),
            "perlin_perm": arr(fs.perlin_perm, fs.n_perlin * 768, np.uint16),
            "img_w": arr(fs.img_w, fs.n_images, np.uint32), "img_h": arr(fs.img_h, fs.n_images, np.uint32),
            "sky_type": fs.sky_type, "sky_image": fs.sky_image,
        }

    def close(self):
        if self._h:
            self._lib.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def grid_build(scene, cell_per_mille=0, lds_budget=0):
    """rt_debug_grid_build (host code of the GPU library, no GPU): the uniform grid rt_scene_upload would build over the spheres
    of `scene`, or None when the scene gets none.  -> dict(origin[3], cell[3], pad, max_coord, dims[3], cells u32 [nz, ny, nx]
    (offset << 12 | count), refs u16, large: list)."""
    lib = _ffi.load_gpu_library()
    ptr = scene.flat_ptr if isinstance(scene, Scene) else C.pointer(scene)
    grid, dims, large = (C.c_float * 8)(), (C.c_uint32 * 3)(), (C.c_uint32 * 4)()
    nc, nr, nl = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    rc = lib.rt_debug_grid_build(ptr, cell_per_mille, lds_budget, grid, dims, None, C.byref(nc), None, C.byref(nr), large, C.byref(nl))
    if rc == -4:  # RT_ERR_UNSUPPORTED: no grid for this scene
        return None
    cells = np.zeros(max(nc.value, 1), np.uint32)
    refs = np.zeros(max(nr.value, 1), np.uint16)
    rc = lib.rt_debug_grid_build(ptr, cell_per_mille, lds_budget, grid, dims, cells.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(nc),
                                 refs.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(nr), large, C.byref(nl))
    if rc != 0:
        raise RtError(f"rt_debug_grid_build failed ({rc})")
    d = [int(x) for x in dims]
    return {"origin": np.array(grid[0:3], np.float32), "cell": np.array(grid[3:6], np.float32), "pad": float(grid[6]), "max_coord": float(grid[7]),
            "dims": d, "cells": cells[:nc.value].reshape(d[2], d[1], d[0]), "refs": refs[:nr.value], "large": [int(large[k]) for k in range(nl.value)]}


def make_params(nx, ny, spp, max_depth=50, seed=95, shard_band=0, shard_count=1, shard_id=0, spp_slice=0, flags=0):
    p = RtParams()
    p.flags = flags
    p.nx, p.ny, p.spp, p.max_depth, p.seed = nx, ny, spp, max_depth, seed
    p.shard_band, p.shard_count, p.shard_id, p.spp_slice = shard_band, shard_count, shard_id, spp_slice
    return p


class Renderer:
    """One GPU context (rt_ctx_create).  Replaces the render half of main.rs:62-129."""

    def __init__(self, device=0):
        self._lib = _ffi.load_gpu_library()  # raises GpuLibraryMissing: no fallback
        self._ctx = C.c_void_p()
        rc = self._lib.rt_ctx_create(device, C.byref(self._ctx))
        if rc != 0:
            raise RtError(f"rt_ctx_create({device}) failed ({rc}): {self._lib.rt_last_error(None).decode()}")
        self.device = device

    def _raise(self, what, rc):
        raise RtError(f"{what} failed ({rc}): {self._lib.rt_last_error(self._ctx).decode()}")

    @property
    def build_id(self):
        """rt_build_id(): hash of the device sources + compiler flags the loaded library was built from."""
        return self._lib.rt_build_id().decode()

    def upload(self, scene):
        ptr = scene.flat_ptr if isinstance(scene, Scene) else C.pointer(scene)
        rc = self._lib.rt_scene_upload(self._ctx, ptr)
        if rc != 0:
            self._raise("rt_scene_upload", rc)

    def prepare(self, params):
        """rt_prepare: start requesting the work buffers of the frame `params` describes (a helper thread inside the library);
        call it before building the scene, as a host that knows its frame size up front would (main.rs:64-67)."""
        rc = self._lib.rt_prepare(self._ctx, C.byref(params))
        if rc != 0:
            self._raise("rt_prepare", rc)

    def set_option(self, option, value):
        """rt_debug_set_option: `option` an _ffi.OPT_* number or its lower-case name.  Per context; every setting renders the
        same bits (they select between equivalent search structures / placements / orders).  Upload-time options
        (tree_placement, texel_pool, grid_cell, general_lds) take effect at the next upload()."""
        opt = _ffi.OPT_NAMES[option] if isinstance(option, str) else int(option)
        rc = self._lib.rt_debug_set_option(self._ctx, opt, int(value))
        if rc != 0:
            self._raise("rt_debug_set_option", rc)

    def get_option(self, option):
        opt = _ffi.OPT_NAMES[option] if isinstance(option, str) else int(option)
        v = C.c_uint32()
        rc = self._lib.rt_debug_get_option(self._ctx, opt, C.byref(v))
        if rc != 0:
            self._raise("rt_debug_get_option", rc)
        return v.value

    def scene_info(self):
        """rt_debug_scene_info as a dict: what upload() built for the closest-hit search (tree placement, grid or not)."""
        info = _ffi.RtSceneInfo()
        rc = self._lib.rt_debug_scene_info(self._ctx, C.byref(info))
        if rc != 0:
            self._raise("rt_debug_scene_info", rc)
        return info.as_dict()

    def shard_rows(self, params):
        return self._lib.rt_shard_rows(params.ny, params.shard_band or 1, params.shard_count, params.shard_id)

    def _pinned(self, shape, dtype):
        """A numpy array over rt_host_alloc'ed (page-locked) memory, kept and reused by this Renderer: rt_render writes it
        by asynchronous copies at PCIe rate (what a host replacing main.rs:109-128 would allocate its frame in)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        key = np.dtype(dtype).char
        if not hasattr(self, "_pin"):
            self._pin, self._pin_retired = {}, []
        ptr, size = self._pin.get(key, (None, 0))
        if size < n:
            if ptr:  # arrays handed out earlier are views of it: an outgrown buffer lives until close()
                self._pin_retired.append(ptr)
            ptr = self._lib.rt_host_alloc(max(n, 1))
            if not ptr:
                raise RtError("rt_host_alloc failed")
            self._pin[key] = (ptr, n)
        buf = (C.c_char * max(n, 1)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def render(self, camera, params, want_rgb8=False, pinned=False):
        """Returns (f32 image [rows, nx, 3] (row 0 = bottom), rgb8 or None, RtStats).  pinned=True: the arrays are views of
        this Renderer's page-locked staging buffers: overwritten by the next pinned render, and their memory is FREED by close() —
        copy what must outlive the Renderer."""
        rows = self.shard_rows(params)
        if pinned:
            img = self._pinned((rows, params.nx, 3), np.float32)
            rgb8 = self._pinned((rows, params.nx, 3), np.uint8) if want_rgb8 else None
        else:
            img = np.zeros((rows, params.nx, 3), dtype=np.float32)
            rgb8 = np.zeros((rows, params.nx, 3), dtype=np.uint8) if want_rgb8 else None
        stats = RtStats()
        rc = self._lib.rt_render(self._ctx, C.byref(camera), C.byref(params),
                                 img.ctypes.data_as(C.POINTER(C.c_float)),
                                 rgb8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_rgb8 else None, C.byref(stats))
        if rc != 0:
            self._raise("rt_render", rc)
        return img, rgb8, stats

    def set_progress(self, fn):
        """fn(spp_done, spp_total, rgb8[rows, nx, 3]) after every slice but the last of a following render()
        (main.rs:114-123, the partial saves); None removes it.  Pick the cadence with make_params(spp_slice=...)."""
        if fn is None:
            self._progress = None
            rc = self._lib.rt_set_progress(self._ctx, _ffi.RtProgressFn(), None)
        else:
            def tramp(_user, done, total, ptr, nx, rows):
                fn(done, total, np.ctypeslib.as_array(ptr, shape=(rows, nx, 3)).copy())
            self._progress = _ffi.RtProgressFn(tramp)  # keep the thunk alive as long as it is registered
            rc = self._lib.rt_set_progress(self._ctx, self._progress, None)
        if rc != 0:
            self._raise("rt_set_progress", rc)

    def render_device(self, camera, params, device_ptr, stream=None, want_stats=True):
        """Renders into HBM at `device_ptr` (e.g. torch_tensor.data_ptr()); nothing crosses PCIe."""
        stats = RtStats()
        rc = self._lib.rt_render_device(self._ctx, C.byref(camera), C.byref(params), C.c_void_p(device_ptr),
                                        C.c_void_p(stream) if stream else None,
                                        C.byref(stats) if want_stats else None)
        if rc != 0:
            self._raise("rt_render_device", rc)
        return stats

    def render_parts(self):
        """rt_debug_render_parts: host-side timeline of the last render as {label: ms} in call order (allocations one by one, the
        hardware-queue probe = first kernel launch, candidate lists + the in-frame synchronisation, enqueue, wait)."""
        import json
        buf = C.create_string_buffer(4096)
        n = self._lib.rt_debug_render_parts(self._ctx, buf, len(buf))
        if n < 0:
            self._raise("rt_debug_render_parts", n)
        return json.loads(buf.value.decode())

    def depth_timings(self, max_n=128):
        """(isect_ms, shade_ms, rays) per depth of the first slice of the last render with FLAG_TIME_DEPTHS."""
        a = np.zeros(max_n, np.float32)
        b = np.zeros(max_n, np.float32)
        r = np.zeros(max_n, np.uint64)
        n = self._lib.rt_get_depth_timings(self._ctx, max_n, a.ctypes.data_as(C.POINTER(C.c_float)),
                                           b.ctypes.data_as(C.POINTER(C.c_float)), r.ctypes.data_as(C.POINTER(C.c_uint64)))
        if n < 0:
            self._raise("rt_get_depth_timings", n)
        return a[:n], b[:n], r[:n]

    def debug_bounce(self, origins, dirs, keys, depth=0, flags=0):
        """One closest-hit + shade step for caller-given rays (rt_debug_bounce)."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        k = np.ascontiguousarray(keys, dtype=np.uint32).reshape(-1, 2)
        n = o.shape[0]
        out = {"hit": np.zeros(n, np.int32), "t": np.zeros(n, np.float32), "radiance": np.zeros((n, 3), np.float32),
               "attenuation": np.zeros((n, 3), np.float32), "o": np.zeros((n, 3), np.float32),
               "d": np.zeros((n, 3), np.float32), "alive": np.zeros(n, np.uint8)}
        io = RtBounceIO()
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        io.n, io.depth, io.flags = n, depth, flags
        io.in_o, io.in_d, io.in_key = fp(o), fp(d), k.ctypes.data_as(C.POINTER(C.c_uint32))
        io.out_hit = out["hit"].ctypes.data_as(C.POINTER(C.c_int32))
        io.out_t, io.out_radiance, io.out_attenuation = fp(out["t"]), fp(out["radiance"]), fp(out["attenuation"])
        io.out_o, io.out_d = fp(out["o"]), fp(out["d"])
        io.out_alive = out["alive"].ctypes.data_as(C.POINTER(C.c_uint8))
        rc = self._lib.rt_debug_bounce(self._ctx, C.byref(io))
        if rc != 0:
            self._raise("rt_debug_bounce", rc)
        return out

    def debug_shared_division(self, x, a):
        """x / a element by element through the kernels' shared-reciprocal division (rt_debug_arithmetic)."""
        return self._debug_arithmetic(0, x, a)

    def debug_sqrt(self, x):
        """sqrt(x) element by element as the kernels take it of discriminants and squared lengths (rt_debug_arithmetic)."""
        return self._debug_arithmetic(1, x, None)

    def debug_to_i32(self, x):
        """`x as i32` (Rust: toward zero, saturating, NaN -> 0) as the kernels convert (rt_debug_arithmetic)."""
        return self._debug_arithmetic(2, x, None).view(np.int32)

    def debug_to_u32(self, x):
        """`x as u32` as the kernels convert (rt_debug_arithmetic)."""
        return self._debug_arithmetic(3, x, None).view(np.uint32)

    def _debug_arithmetic(self, op, x, a):
        x = np.ascontiguousarray(x, dtype=np.float32).ravel()
        a = x if a is None else np.ascontiguousarray(a, dtype=np.float32).ravel()
        assert x.shape == a.shape
        out = np.zeros_like(x)
        fp = lambda v: v.ctypes.data_as(C.POINTER(C.c_float))
        rc = self._lib.rt_debug_arithmetic(self._ctx, op, len(x), fp(x), fp(a), fp(out))
        if rc != 0:
            self._raise("rt_debug_arithmetic", rc)
        return out

    def deinterleave_bands(self, d_gathered, nx, ny, band, n_shards, d_out_f32=None, d_out_u8=None, stream=None):
        """rt_deinterleave_bands on device pointers (ints): gathered band buffers -> frame in image row order."""
        rc = self._lib.rt_deinterleave_bands(self._ctx, C.c_void_p(d_gathered), nx, ny, band, n_shards,
                                             C.c_void_p(d_out_f32) if d_out_f32 else None, C.c_void_p(d_out_u8) if d_out_u8 else None,
                                             C.c_void_p(stream) if stream else None)
        if rc != 0:
            self._raise("rt_deinterleave_bands", rc)

    def close(self):
        if self._ctx:
            self._lib.rt_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()
        for ptr in [q for q, _ in getattr(self, "_pin", {}).values()] + getattr(self, "_pin_retired", []):
            if ptr:
                self._lib.rt_host_free(ptr)  # (invalidates every array a pinned render() returned)
        self._pin, self._pin_retired = {}, []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_png(path, rgb8):
    """img.save(file_name) of main.rs:121,128: rgb8 is [ny, nx, 3] uint8 with row 0 at the top, as render(..., want_rgb8=True)
    returns it.  Written by the host library (rth_png_write), atomically."""
    lib = _ffi.load_host_library()
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("save_png: expected [ny, nx, 3] uint8")
    rc = lib.rth_png_write(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0])
    if rc != 0:
        raise RtError(f"rth_png_write({path}) failed ({rc})")


def output_file_name(unix_seconds=-1):
    """main.rs:110-112: local RFC 3339 time with ':' -> '-', cut before the fraction, + '.png'."""
    lib = _ffi.load_host_library()
    buf = C.create_string_buffer(64)
    if lib.rth_output_file_name(int(unix_seconds), buf, 64) != 0:
        raise RtError("rth_output_file_name failed")
    return buf.value.decode()


class MultiRenderer:
    """The GPUs of one node behind one handle (rt_multi_create): every device renders its row-interleaved bands, RCCL
    gathers the frame inside the library.  Replaces the thread-pool fan-out of main.rs:72-108 for a node.
    `copy_gather=True` (rt_multi_create_ex, RT_MULTI_COPY_GATHER): device-to-device copies instead of RCCL; `devices` may
    then repeat an id — several contexts side by side on one GPU, the test hook for the n > 1 code on a one-GPU box."""

    def __init__(self, devices, copy_gather=False):
        self._lib = _ffi.load_gpu_library()
        self._m = C.c_void_p()
        ids = (C.c_int * len(devices))(*devices)
        rc = self._lib.rt_multi_create_ex(ids, len(devices), _ffi.MULTI_COPY_GATHER if copy_gather else 0, C.byref(self._m))
        if rc != 0:
            raise RtError(f"rt_multi_create({list(devices)}) failed ({rc}): {self._lib.rt_multi_last_error(None).decode()}")

    def _raise(self, what, rc):
        raise RtError(f"{what} failed ({rc}): {self._lib.rt_multi_last_error(self._m).decode()}")

    @property
    def n_devices(self):
        return self._lib.rt_multi_device_count(self._m)

    def upload(self, scene):
        ptr = scene.flat_ptr if isinstance(scene, Scene) else C.pointer(scene)
        rc = self._lib.rt_multi_scene_upload(self._m, ptr)
        if rc != 0:
            self._raise("rt_multi_scene_upload", rc)

    def render(self, camera, params, want_rgb8=False):
        """Returns (f32 image [ny, nx, 3] (row 0 = bottom), rgb8 or None, RtStats summed over the devices)."""
        img = np.zeros((params.ny, params.nx, 3), dtype=np.float32)
        rgb8 = np.zeros((params.ny, params.nx, 3), dtype=np.uint8) if want_rgb8 else None
        st = RtStats()
        rc = self._lib.rt_multi_render(self._m, C.byref(camera), C.byref(params), img.ctypes.data_as(C.POINTER(C.c_float)),
                                       rgb8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_rgb8 else None, C.byref(st))
        if rc != 0:
            self._raise("rt_multi_render", rc)
        return img, rgb8, st

    def close(self):
        if self._m:
            self._lib.rt_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
