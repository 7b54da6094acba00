// rt_multi.h — the multi-GPU entry points of include/rtow_mi355x.h (included at the end of rt_api.hip).
//
// one process, n devices of one node, the framebuffer gather behind the C-ABI so that a Rust or C
// host needs no torch and no RCCL code of its own.  Pixels are independent (RNG keyed by pixel and sample), so the
// path shards with no exchange until the end: device r renders the image rows of the row-interleaved bands
// (j / band) % n == r into an equal-sized band buffer, ONE gather to the first device — a group of ncclSend / ncclRecv
// (RCCL; over xGMI every peer pair has its own link, so the n - 1 transfers of 12.4 MB at 4K run side by side; only the
// first device receives, as in the one-process-per-GPU path's dist.gather(dst=0)) — brings the buffers together and a
// kernel on the first device restores row order, producing the f32 frame and the flipped RGB8 image of main.rs:98-105,127.
//
// librccl is opened with dlopen at rt_multi_create: the single-GPU entry points carry no RCCL dependency, and the
// library still loads on a machine without RCCL.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <thread>

namespace rt {

// gathered[r][lj][i] -> image row j = ((lj / band) * n + r) * band + lj % band; out_u8 is gamma-2, *255.99, flipped
__global__ __launch_bounds__(256) void k_bands_to_image(const float* __restrict__ gathered, float* __restrict__ out_f32,
                                                        uint8_t* __restrict__ out_u8, uint32_t nx, uint32_t ny, uint32_t band,
                                                        uint32_t n_shards, uint32_t rows_pad) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= nx * ny) return;
    const uint32_t j = p / nx, i = p - j * nx;
    uint32_t r = 0, lj = j;
    if (n_shards > 1u) {
        const uint32_t b = j / band;
        r = b % n_shards;
        lj = (b / n_shards) * band + j % band;
    }
    const float* src = gathered + (((size_t)r * rows_pad + lj) * nx + i) * 3u;
    const float c[3] = {src[0], src[1], src[2]};
    if (out_f32) out_f32[3 * (size_t)p] = c[0], out_f32[3 * (size_t)p + 1] = c[1], out_f32[3 * (size_t)p + 2] = c[2];
    if (out_u8) {
        const size_t dst = ((size_t)(ny - 1u - j) * nx + i) * 3u;
        for (int k = 0; k < 3; ++k) {
            const float g = sqrtf(c[k]) * 255.99f;
            const uint32_t u = (g == g && g > 0.0f) ? (g >= 255.0f ? 255u : (uint32_t)g) : 0u;
            out_u8[dst + k] = (uint8_t)u;
        }
    }
}

} // namespace rt

struct RtMulti {
    std::vector<RtCtx*> ctx;
    std::vector<int> devices;
    std::string err;
    // RCCL, resolved at run time
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<ncclComm_t> comms;
    bool copy_gather = false; // RT_MULTI_COPY_GATHER: device-to-device copies instead of ncclSend / ncclRecv, no RCCL at all
    std::vector<hipEvent_t> gather_done; // copy_gather: context i's copy has been enqueued up to this event
    // per device: band buffer; first device: the gathered band buffers and the full frame
    std::vector<DevBuf> local;
    DevBuf gathered, full_f32, full_u8;
};

namespace {
thread_local std::string g_multi_create_error;

int multi_fail(RtMulti* m, int code, const std::string& msg) {
    if (m) m->err = msg;
    else g_multi_create_error = msg;
    return code;
}
} // namespace

extern "C" {

const char* rt_multi_last_error(const RtMulti* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

void rt_multi_destroy(RtMulti* m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        (void)hipSetDevice(m->devices[i]);
        if (i < m->local.size()) free_buf(m->local[i]);
    }
    if (!m->ctx.empty()) {
        (void)hipSetDevice(m->devices[0]);
        free_buf(m->gathered), free_buf(m->full_f32), free_buf(m->full_u8);
    }
    for (ncclComm_t c : m->comms)
        if (c && m->CommDestroy) (void)m->CommDestroy(c);
    for (size_t i = 0; i < m->gather_done.size(); ++i) {
        (void)hipSetDevice(m->devices[i]);
        if (m->gather_done[i]) (void)hipEventDestroy(m->gather_done[i]);
    }
    for (RtCtx* c : m->ctx) rt_ctx_destroy(c);
    if (m->lib) dlclose(m->lib);
    delete m;
}

int rt_multi_create(const int* device_ids, int n_devices, RtMulti** out) { return rt_multi_create_ex(device_ids, n_devices, 0u, out); }

int rt_multi_create_ex(const int* device_ids, int n_devices, uint32_t flags, RtMulti** out) {
    if (!out) return multi_fail(nullptr, RT_ERR_INVALID, "rt_multi_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices <= 0) return multi_fail(nullptr, RT_ERR_INVALID, "rt_multi_create: need at least one device id");
    if (flags & ~RT_MULTI_COPY_GATHER) return multi_fail(nullptr, RT_ERR_INVALID, "rt_multi_create_ex: unknown flag");
    const bool copy_gather = (flags & RT_MULTI_COPY_GATHER) != 0u;
    for (int i = 0; i < n_devices && !copy_gather; ++i) // (an RCCL communicator cannot hold one device twice)
        for (int k = 0; k < i; ++k)
            if (device_ids[i] == device_ids[k]) return multi_fail(nullptr, RT_ERR_INVALID, "rt_multi_create: a device id is listed twice");
    RtMulti* m = new (std::nothrow) RtMulti();
    if (!m) return multi_fail(nullptr, RT_ERR_NOMEM, "rt_multi_create: out of host memory");
    auto bail = [&](int code, const std::string& msg) {
        rt_multi_destroy(m);
        return multi_fail(nullptr, code, msg);
    };
    for (int i = 0; i < n_devices; ++i) {
        RtCtx* c = nullptr;
        const int rc = rt_ctx_create(device_ids[i], &c);
        if (rc) return bail(rc, std::string("rt_multi_create: ") + rt_last_error(nullptr));
        m->ctx.push_back(c);
        m->devices.push_back(device_ids[i]);
    }
    m->local.resize(n_devices);
    if (copy_gather) { // no RCCL: the gather is n device-to-device copies ordered by one event per context
        m->copy_gather = true;
        m->gather_done.assign(n_devices, nullptr);
        for (int i = 0; i < n_devices; ++i) {
            hipError_t e = hipSetDevice(device_ids[i]);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&m->gather_done[i], hipEventDisableTiming);
            if (e != hipSuccess) return bail(RT_ERR_DEVICE, std::string("rt_multi_create_ex: hipEventCreate: ") + hipGetErrorString(e));
        }
        *out = m;
        return RT_OK;
    }
    {   // the RCCL that belongs to the HIP runtime this library is bound to: a process may hold two ROCm stacks (PyTorch
        // wheels bundle their own), and an RCCL from the other one brings up a second HSA runtime that sees no device
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1"), names.push_back(dir + "librccl.so");
            }
        }
        names.push_back("librccl.so.1"), names.push_back("librccl.so");
        for (const std::string& name : names)
            if (!m->lib) m->lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
    }
    if (!m->lib) return bail(RT_ERR_UNSUPPORTED, std::string("rt_multi_create: librccl not found (") + dlerror() + ")");
#define RT_RCCL_SYM(field, sym)                                                                    \
    m->field = reinterpret_cast<decltype(m->field)>(dlsym(m->lib, sym));                           \
    if (!m->field) return bail(RT_ERR_UNSUPPORTED, std::string("rt_multi_create: librccl has no ") + sym)
    RT_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RT_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RT_RCCL_SYM(Send, "ncclSend");
    RT_RCCL_SYM(Recv, "ncclRecv");
    RT_RCCL_SYM(GroupStart, "ncclGroupStart");
    RT_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RT_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RT_RCCL_SYM
    m->comms.assign(n_devices, nullptr);
    const ncclResult_t nr = m->CommInitAll(m->comms.data(), n_devices, device_ids); // one communicator per device, one process
    if (nr != ncclSuccess) return bail(RT_ERR_DEVICE, std::string("rt_multi_create: ncclCommInitAll: ") + m->GetErrorString(nr));
    *out = m;
    return RT_OK;
}

int rt_multi_device_count(const RtMulti* m) { return m ? (int)m->ctx.size() : 0; }

int rt_multi_scene_upload(RtMulti* m, const RtFlatScene* scene) {
    if (!m) return RT_ERR_INVALID;
    for (size_t i = 0; i < m->ctx.size(); ++i) { // the scene is replicated: <= ~25 MB per device
        const int rc = rt_scene_upload(m->ctx[i], scene);
        if (rc) return multi_fail(m, rc, std::string("device ") + std::to_string(m->devices[i]) + ": " + rt_last_error(m->ctx[i]));
    }
    return RT_OK;
}

int rt_deinterleave_bands(RtCtx* ctx, const void* d_gathered, uint32_t nx, uint32_t ny, uint32_t band, uint32_t n_shards,
                          void* d_out_rgb_f32, void* d_out_rgb8, void* stream) {
    if (!ctx) return RT_ERR_INVALID;
    if (!d_gathered || (!d_out_rgb_f32 && !d_out_rgb8) || nx == 0 || ny == 0 || n_shards == 0)
        return fail(ctx, RT_ERR_INVALID, "rt_deinterleave_bands: bad argument");
    if (band == 0) band = 1;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t rows_pad = 0;
    for (uint32_t r = 0; r < n_shards; ++r) rows_pad = std::max(rows_pad, rt_shard_rows(ny, band, n_shards, r));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    hipLaunchKernelGGL(k_bands_to_image, dim3((unsigned)(((size_t)nx * ny + 255u) / 256u)), dim3(256), 0, st, (const float*)d_gathered,
                       (float*)d_out_rgb_f32, (uint8_t*)d_out_rgb8, nx, ny, band, n_shards, rows_pad);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

int rt_multi_render(RtMulti* m, const RtCamera* cam, const RtParams* params, float* out_rgb_f32, uint8_t* out_rgb8, RtStats* stats) {
    if (!m) return RT_ERR_INVALID;
    if (!cam || !params) return multi_fail(m, RT_ERR_INVALID, "rt_multi_render: camera/params NULL");
    const uint32_t n = (uint32_t)m->ctx.size();
    const uint32_t nx = params->nx, ny = params->ny, band = params->shard_band ? params->shard_band : 8u;
    if (nx == 0 || ny == 0) return multi_fail(m, RT_ERR_INVALID, "rt_multi_render: nx and ny must be > 0");
    uint32_t rows_pad = 0;
    for (uint32_t r = 0; r < n; ++r) rows_pad = std::max(rows_pad, rt_shard_rows(ny, band, n, r));
    const size_t band_floats = (size_t)rows_pad * nx * 3u;
    // ---- every device renders its bands; one host thread per device, so that no device waits for another's enqueue
    std::vector<int> rcs(n, RT_OK);
    std::vector<RtStats> sts(n);
    auto work = [&](uint32_t i) {
        RtCtx* c = m->ctx[i];
        if (hipSetDevice(m->devices[i]) != hipSuccess) {
            rcs[i] = fail(c, RT_ERR_DEVICE, "hipSetDevice failed");
            return;
        }
        // (plain hipMalloc: these are the buffers RCCL and the other devices' copy engines address)
        if ((rcs[i] = ensure(c, m->local[i], band_floats * sizeof(float), true))) return;
        if (i == 0 && (rcs[i] = ensure(c, m->gathered, band_floats * sizeof(float) * n, true))) return;
        if (hipMemsetAsync(m->local[i].p, 0, band_floats * sizeof(float), c->stream) != hipSuccess) { // the padding rows
            rcs[i] = fail(c, RT_ERR_DEVICE, "hipMemsetAsync failed");
            return;
        }
        RtParams p = *params;
        p.shard_band = band, p.shard_count = n, p.shard_id = i;
        rcs[i] = rt_render_device(c, cam, &p, m->local[i].p, nullptr, &sts[i]);
    };
    if (n == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (uint32_t i = 0; i < n; ++i) th.emplace_back(work, i);
        for (auto& t : th) t.join();
    }
    for (uint32_t i = 0; i < n; ++i)
        if (rcs[i]) return multi_fail(m, rcs[i], std::string("device ") + std::to_string(m->devices[i]) + ": " + rt_last_error(m->ctx[i]));
    // ---- the one exchange step: every band buffer to slot i of the first device's gathered buffer ---------------
    float* const slots = (float*)m->gathered.p;
    if (m->copy_gather) {
        // the same data movement as plain device-to-device copies on context i's stream; the first context's stream then waits
        // for all of them
        hipError_t e = hipSuccess;
        for (uint32_t i = 0; i < n && e == hipSuccess; ++i) {
            e = hipSetDevice(m->devices[i]);
            if (e == hipSuccess)
                e = hipMemcpyAsync(slots + (size_t)i * band_floats, m->local[i].p, band_floats * sizeof(float), hipMemcpyDeviceToDevice, m->ctx[i]->stream);
            if (e == hipSuccess) e = hipEventRecord(m->gather_done[i], m->ctx[i]->stream);
        }
        if (e == hipSuccess) e = hipSetDevice(m->devices[0]);
        for (uint32_t k = 1; k < n && e == hipSuccess; ++k) e = hipStreamWaitEvent(m->ctx[0]->stream, m->gather_done[k], 0);
        if (e != hipSuccess) return multi_fail(m, RT_ERR_DEVICE, std::string("rt_multi_render: copy gather: ") + hipGetErrorString(e));
    } else {
        // device 0's own band is a local copy; devices 1 .. n-1 send, device 0 posts the matching receives — one group, so that
        // RCCL sees every pair before it starts any ("grouped ncclSend / ncclRecv to rank 0")
        if (hipSetDevice(m->devices[0]) != hipSuccess ||
            hipMemcpyAsync(slots, m->local[0].p, band_floats * sizeof(float), hipMemcpyDeviceToDevice, m->ctx[0]->stream) != hipSuccess)
            return multi_fail(m, RT_ERR_DEVICE, "rt_multi_render: copy of the first device's own band failed");
        ncclResult_t nr = m->GroupStart();
        for (uint32_t i = 1; i < n && nr == ncclSuccess; ++i) {
            nr = m->Send(m->local[i].p, band_floats, ncclFloat, 0, m->comms[i], m->ctx[i]->stream);
            if (nr == ncclSuccess) nr = m->Recv(slots + (size_t)i * band_floats, band_floats, ncclFloat, (int)i, m->comms[0], m->ctx[0]->stream);
        }
        const ncclResult_t ne = m->GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return multi_fail(m, RT_ERR_DEVICE, std::string("rt_multi_render: ncclSend / ncclRecv: ") + m->GetErrorString(nr));
    }
    // ---- row order + quantisation on the first device, then to the host --------------------------------------
    RtCtx* c0 = m->ctx[0];
    if (hipSetDevice(m->devices[0]) != hipSuccess) return multi_fail(m, RT_ERR_DEVICE, "hipSetDevice failed");
    const size_t n_px = (size_t)nx * ny;
    int rc;
    if ((rc = ensure(c0, m->full_f32, n_px * 3u * sizeof(float))) || (out_rgb8 && (rc = ensure(c0, m->full_u8, n_px * 3u))))
        return multi_fail(m, rc, rt_last_error(c0));
    if ((rc = rt_deinterleave_bands(c0, m->gathered.p, nx, ny, band, n, m->full_f32.p, out_rgb8 ? m->full_u8.p : nullptr, c0->stream)))
        return multi_fail(m, rc, rt_last_error(c0));
    hipError_t e = hipSuccess;
    if (out_rgb_f32) e = hipMemcpyAsync(out_rgb_f32, m->full_f32.p, n_px * 3u * sizeof(float), hipMemcpyDeviceToHost, c0->stream);
    if (e == hipSuccess && out_rgb8) e = hipMemcpyAsync(out_rgb8, m->full_u8.p, n_px * 3u, hipMemcpyDeviceToHost, c0->stream);
    for (uint32_t i = 0; i < n && e == hipSuccess; ++i) { // every device's part of the collective has finished
        if ((e = hipSetDevice(m->devices[i])) == hipSuccess) e = hipStreamSynchronize(m->ctx[i]->stream);
    }
    if (e != hipSuccess) return multi_fail(m, RT_ERR_DEVICE, std::string("rt_multi_render: ") + hipGetErrorString(e));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (uint32_t i = 0; i < n; ++i) {
            const RtStats& s = sts[i];
            stats->n_paths += s.n_paths, stats->n_rays += s.n_rays, stats->n_rays_secondary += s.n_rays_secondary;
            stats->n_texture_fetches += s.n_texture_fetches, stats->n_bad_dir += s.n_bad_dir;
            stats->bytes_algorithmic += s.bytes_algorithmic, stats->bytes_trace_algorithmic += s.bytes_trace_algorithmic;
            stats->n_trace_launches += s.n_trace_launches;
            stats->n_slices = std::max(stats->n_slices, s.n_slices);
            stats->seconds_total = std::max(stats->seconds_total, s.seconds_total); // devices run side by side
            stats->seconds_trace = std::max(stats->seconds_trace, s.seconds_trace);
            stats->seconds_device = std::max(stats->seconds_device, s.seconds_device);
            for (int d = 0; d < 64; ++d) stats->rays_per_depth[d] += s.rays_per_depth[d];
        }
    }
    return RT_OK;
}

} // extern "C"
