// Microbenchmark: what device-memory allocation costs on an MI355X box, and when (round 6, the first-frame question).
// The first rt_render of a process allocates tens of GB; it took 8 ms on a quiet device and 0.5-3.9 s right after other processes
// had freed as much (profiles/round5/first_frame_five_fresh_processes.txt).  This program times the pieces in a chosen state:
//   alloc_probe probe <label>          hipMalloc of 64 MB, 1, 4, 16, 32 GB in a row, the first kernel touching each, hipFree
//   alloc_probe hold <GB> <seconds>    allocate + touch, sleep, free (a tenant beside the measured process)
//   alloc_probe churn <GB> <times>     allocate + touch + free, <times> times (memory the driver has to take back)
//   alloc_probe vmm <label> <chunkMB> <GB>   hipMemAddressReserve once, then hipMemCreate + hipMemMap + hipMemSetAccess per chunk
//   alloc_probe concurrent <label> <GB>      a stream of streaming kernels on the main thread while a helper thread hipMallocs <GB>
//   alloc_probe small <label> <n> <MB>       n hipMallocs of <MB> each (does a small request wait as long as a large one?)
// One JSON line per measurement on stdout.  hipcc --offload-arch=gfx950 -O2 -o alloc_probe scripts/micro/alloc_probe.hip -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));                   \
            std::fflush(stdout);                                                                   \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

__global__ void k_touch(uint4* p, size_t n16) { // one 16 B store per 4 KB page is enough to fault / validate it; here: every 16 B (streaming fill)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ void k_stream(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = a[i];
        v.x += 1u;
        b[i] = v;
    }
}

static double touch(void* p, size_t bytes, hipStream_t st) {
    const auto t0 = clk::now();
    hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, st, (uint4*)p, bytes / 16);
    CK(hipStreamSynchronize(st));
    return ms_since(t0);
}

static size_t free_now() {
    size_t f = 0, t = 0;
    CK(hipMemGetInfo(&f, &t));
    return f;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string mode = argv[1];
    const auto t_start = clk::now();
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    { // the runtime's own first-launch work out of the way
        void* w = nullptr;
        CK(hipMalloc(&w, 1 << 20));
        touch(w, 1 << 20, st);
        CK(hipFree(w));
    }
    const double init_ms = ms_since(t_start);
    const size_t GB = 1ull << 30, MB = 1ull << 20;
    if (mode == "probe") {
        const char* label = argc > 2 ? argv[2] : "";
        std::printf("{\"mode\": \"probe\", \"label\": \"%s\", \"runtime_init_ms\": %.1f, \"free_gb\": %.1f}\n", label, init_ms, free_now() / 1e9);
        const size_t sizes[] = {64 * MB, 1 * GB, 4 * GB, 16 * GB, 32 * GB};
        std::vector<void*> ps;
        for (size_t s : sizes) {
            void* p = nullptr;
            const auto t0 = clk::now();
            CK(hipMalloc(&p, s));
            const double a = ms_since(t0);
            const double t1 = touch(p, s, st), t2 = touch(p, s, st);
            std::printf("{\"mode\": \"probe\", \"label\": \"%s\", \"gb\": %.3f, \"hipMalloc_ms\": %.2f, \"first_touch_ms\": %.2f, \"second_touch_ms\": %.2f}\n", label,
                        s / 1e9, a, t1, t2);
            std::fflush(stdout);
            ps.push_back(p);
        }
        for (size_t k = 0; k < ps.size(); ++k) {
            const auto t0 = clk::now();
            CK(hipFree(ps[k]));
            std::printf("{\"mode\": \"probe\", \"label\": \"%s\", \"gb\": %.3f, \"hipFree_ms\": %.2f}\n", label, sizes[k] / 1e9, ms_since(t0));
        }
    } else if (mode == "small") {
        const char* label = argc > 2 ? argv[2] : "";
        const int n = argc > 3 ? std::atoi(argv[3]) : 8;
        const size_t s = (argc > 4 ? std::atoi(argv[4]) : 64) * MB;
        std::vector<void*> ps;
        for (int k = 0; k < n; ++k) {
            void* p = nullptr;
            const auto t0 = clk::now();
            CK(hipMalloc(&p, s));
            const double a = ms_since(t0);
            const double t1 = touch(p, s, st);
            std::printf("{\"mode\": \"small\", \"label\": \"%s\", \"k\": %d, \"mb\": %zu, \"hipMalloc_ms\": %.2f, \"first_touch_ms\": %.2f, \"since_start_ms\": %.1f}\n", label, k,
                        s / MB, a, t1, ms_since(t_start));
            std::fflush(stdout);
            ps.push_back(p);
        }
        for (void* p : ps) CK(hipFree(p));
    } else if (mode == "hold") {
        const size_t s = (size_t)std::atoi(argv[2]) * GB;
        const int secs = std::atoi(argv[3]);
        void* p = nullptr;
        CK(hipMalloc(&p, s));
        touch(p, s, st);
        std::printf("{\"mode\": \"hold\", \"gb\": %.1f, \"holding\": true}\n", s / 1e9);
        std::fflush(stdout);
        std::this_thread::sleep_for(std::chrono::seconds(secs));
        CK(hipFree(p));
    } else if (mode == "churn") {
        const size_t s = (size_t)std::atoi(argv[2]) * GB;
        const int times = std::atoi(argv[3]);
        for (int k = 0; k < times; ++k) {
            void* p = nullptr;
            const auto t0 = clk::now();
            CK(hipMalloc(&p, s));
            const double a = ms_since(t0);
            const double t1 = touch(p, s, st);
            const auto t2 = clk::now();
            CK(hipFree(p));
            std::printf("{\"mode\": \"churn\", \"gb\": %.1f, \"k\": %d, \"hipMalloc_ms\": %.2f, \"touch_ms\": %.2f, \"hipFree_ms\": %.2f}\n", s / 1e9, k, a, t1, ms_since(t2));
            std::fflush(stdout);
        }
    } else if (mode == "vmm") {
        const char* label = argc > 2 ? argv[2] : "";
        const size_t chunk = (size_t)std::atoi(argv[3]) * MB;
        const size_t total = (size_t)std::atoi(argv[4]) * GB;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        void* base = nullptr;
        auto t0 = clk::now();
        CK(hipMemAddressReserve(&base, total, gran, nullptr, 0));
        std::printf("{\"mode\": \"vmm\", \"label\": \"%s\", \"granularity\": %zu, \"reserve_gb\": %.1f, \"reserve_ms\": %.3f}\n", label, gran, total / 1e9, ms_since(t0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        const auto t_all = clk::now();
        for (size_t off = 0; off < total; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            t0 = clk::now();
            CK(hipMemCreate(&h, chunk, &prop, 0));
            const double c = ms_since(t0);
            t0 = clk::now();
            CK(hipMemMap((char*)base + off, chunk, 0, h, 0));
            const double m = ms_since(t0);
            t0 = clk::now();
            CK(hipMemSetAccess((char*)base + off, chunk, &acc, 1));
            const double a = ms_since(t0);
            hs.push_back(h);
            if (off / chunk < 6 || (off / chunk) % 8 == 0)
                std::printf("{\"mode\": \"vmm\", \"label\": \"%s\", \"chunk\": %zu, \"create_ms\": %.3f, \"map_ms\": %.3f, \"access_ms\": %.3f, \"since_ms\": %.1f}\n", label, off / chunk,
                            c, m, a, ms_since(t_all));
            std::fflush(stdout);
        }
        const double all = ms_since(t_all);
        const double t1 = touch(base, total, st), t2 = touch(base, total, st);
        // a bandwidth check on mapped memory against hipMalloc'ed memory of the same size (is the mapping as good?)
        void* plain = nullptr;
        t0 = clk::now();
        CK(hipMalloc(&plain, total));
        const double pm = ms_since(t0);
        const double p1 = touch(plain, total, st), p2 = touch(plain, total, st);
        std::printf("{\"mode\": \"vmm\", \"label\": \"%s\", \"total_gb\": %.1f, \"chunk_mb\": %zu, \"map_all_ms\": %.1f, \"first_touch_ms\": %.2f, \"second_touch_ms\": %.2f, "
                    "\"plain_hipMalloc_ms\": %.2f, \"plain_first_touch_ms\": %.2f, \"plain_second_touch_ms\": %.2f}\n",
                    label, total / 1e9, chunk / MB, all, t1, t2, pm, p1, p2);
        CK(hipFree(plain));
        t0 = clk::now();
        for (size_t k = 0; k < hs.size(); ++k) {
            CK(hipMemUnmap((char*)base + k * chunk, chunk));
            CK(hipMemRelease(hs[k]));
        }
        CK(hipMemAddressFree(base, total));
        std::printf("{\"mode\": \"vmm\", \"label\": \"%s\", \"unmap_release_all_ms\": %.1f}\n", label, ms_since(t0));
    } else if (mode == "vmmmix") { // chunks of different sizes side by side, mapped by different threads: what does hipMemSetAccess accept?
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        void* base = nullptr;
        CK(hipMemAddressReserve(&base, 8 * GB, 2 * MB, nullptr, 0));
        auto one = [&](const char* what, size_t at, size_t bytes) {
            hipMemGenericAllocationHandle_t h;
            hipError_t e1 = hipMemCreate(&h, bytes, &prop, 0);
            hipError_t e2 = e1 == hipSuccess ? hipMemMap((char*)base + at, bytes, 0, h, 0) : hipErrorUnknown;
            hipError_t e3 = e2 == hipSuccess ? hipMemSetAccess((char*)base + at, bytes, &acc, 1) : hipErrorUnknown;
            (void)hipGetLastError();
            std::printf("{\"mode\": \"vmmmix\", \"what\": \"%s\", \"at_mb\": %zu, \"mb\": %zu, \"create\": \"%s\", \"map\": \"%s\", \"access\": \"%s\"}\n", what, at / MB,
                        bytes / MB, hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
            std::fflush(stdout);
        };
        one("64 MB at 0, main thread", 0, 64 * MB);
        one("64 MB behind it, main thread", 64 * MB, 64 * MB);
        one("512 MB behind them (offset 128 MB), main thread", 128 * MB, 512 * MB);
        std::thread([&] { CK(hipSetDevice(0)); one("512 MB at 1 GB, another thread", 1 * GB, 512 * MB); }).join();
        std::thread([&] { CK(hipSetDevice(0)); one("64 MB at 1.5 GB, another thread", 1536 * MB, 64 * MB); }).join();
        std::thread([&] { CK(hipSetDevice(0)); one("512 MB at 1.5 GB + 64 MB, another thread", 1600 * MB, 512 * MB); }).join();
        one("whole-range access over everything mapped from 0 (640 MB)", 4 * GB, 64 * MB);
        hipError_t e = hipMemSetAccess(base, 640 * MB, &acc, 1);
        std::printf("{\"mode\": \"vmmmix\", \"what\": \"hipMemSetAccess(base, 640 MB) over three mappings\", \"access\": \"%s\"}\n", hipGetErrorString(e));
    } else if (mode == "stallprobe") { // what else waits while one allocation of this process is stuck inside the driver?
        const char* label = argc > 2 ? argv[2] : "";
        const size_t total = (size_t)std::atoi(argv[3]) * GB, chunk = 128 * MB;
        const bool with_malloc = argc > 4 && std::atoi(argv[4]) != 0; // (a blocked hipMalloc would hide what the other calls do meanwhile)
        void *dbuf = nullptr, *hpin = nullptr;
        CK(hipMalloc(&dbuf, 64 * MB));
        CK(hipHostMalloc(&hpin, 1 * MB, hipHostMallocDefault));
        std::vector<char> hpage(1 * MB, 1);
        touch(dbuf, 64 * MB, st);
        std::atomic<int> state{0};
        std::atomic<char*> vmm0{nullptr}; // the first mapped chunk: a destination inside the growing range for the main thread
        double slowest_chunk = 0, grow_ms = 0;
        int slowest_at = -1, n_chunks = 0;
        std::thread helper([&] {
            CK(hipSetDevice(0));
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            void* base = nullptr;
            CK(hipMemAddressReserve(&base, total, 2 * MB, nullptr, 0));
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
            state = 1;
            const auto tg = clk::now();
            std::vector<hipMemGenericAllocationHandle_t> hs;
            for (size_t off = 0; off < total; off += chunk) {
                const auto t0 = clk::now();
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, chunk, &prop, 0));
                CK(hipMemMap((char*)base + off, chunk, 0, h, 0));
                CK(hipMemSetAccess((char*)base + off, chunk, &acc, 1));
                const double ms = ms_since(t0);
                if (ms > slowest_chunk) slowest_chunk = ms, slowest_at = (int)(off / chunk);
                hs.push_back(h);
                ++n_chunks;
                if (off == 0) vmm0 = (char*)base;
            }
            grow_ms = ms_since(tg);
            state = 2;
            for (size_t k = 0; k < hs.size(); ++k) {
                CK(hipMemUnmap((char*)base + k * chunk, chunk));
                CK(hipMemRelease(hs[k]));
            }
            CK(hipMemAddressFree(base, total));
        });
        double mx_launch = 0, mx_h2d_page = 0, mx_h2d_pin = 0, mx_malloc = 0, mx_event = 0, mx_vmm_kernel = 0, mx_vmm_h2d = 0, mx_vmm_memset = 0, mx_vmm_d2h = 0;
        int iters = 0;
        hipEvent_t ev;
        CK(hipEventCreate(&ev));
        while (state.load() < 2) {
            if (state.load() == 0) continue;
            auto t0 = clk::now();
            hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, st, (uint4*)dbuf, (1 * MB) / 16);
            CK(hipStreamSynchronize(st));
            mx_launch = std::max(mx_launch, ms_since(t0));
            t0 = clk::now();
            CK(hipMemcpyAsync(dbuf, hpage.data(), 1 * MB, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            mx_h2d_page = std::max(mx_h2d_page, ms_since(t0));
            t0 = clk::now();
            CK(hipMemcpyAsync(dbuf, hpin, 1 * MB, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            mx_h2d_pin = std::max(mx_h2d_pin, ms_since(t0));
            t0 = clk::now();
            CK(hipEventRecord(ev, st));
            CK(hipEventSynchronize(ev));
            mx_event = std::max(mx_event, ms_since(t0));
            if (char* v = vmm0.load()) { // the same operations with the already mapped first chunk of the GROWING range as the target
                t0 = clk::now();
                hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, st, (uint4*)v, (1 * MB) / 16);
                CK(hipStreamSynchronize(st));
                mx_vmm_kernel = std::max(mx_vmm_kernel, ms_since(t0));
                t0 = clk::now();
                CK(hipMemcpyAsync(v, hpin, 1 * MB, hipMemcpyHostToDevice, st));
                CK(hipStreamSynchronize(st));
                mx_vmm_h2d = std::max(mx_vmm_h2d, ms_since(t0));
                t0 = clk::now();
                CK(hipMemsetAsync(v, 0, 1 * MB, st));
                CK(hipStreamSynchronize(st));
                mx_vmm_memset = std::max(mx_vmm_memset, ms_since(t0));
                t0 = clk::now();
                CK(hipMemcpyAsync(hpin, v, 1 * MB, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                mx_vmm_d2h = std::max(mx_vmm_d2h, ms_since(t0));
            }
            if (with_malloc) {
                t0 = clk::now();
                void* q = nullptr;
                CK(hipMalloc(&q, 1 * MB));
                mx_malloc = std::max(mx_malloc, ms_since(t0));
                CK(hipFree(q));
            }
            ++iters;
        }
        helper.join();
        std::printf("{\"mode\": \"stallprobe\", \"label\": \"%s\", \"grow_gb\": %.1f, \"chunks\": %d, \"grow_ms\": %.1f, \"slowest_chunk_ms\": %.1f, \"slowest_chunk\": %d, "
                    "\"main_thread_iterations\": %d, \"slowest_ms\": {\"kernel_launch_and_sync\": %.2f, \"h2d_1mb_pageable\": %.2f, \"h2d_1mb_pinned\": %.2f, "
                    "\"event_record_and_sync\": %.2f, \"hipMalloc_1mb\": %.2f, \"kernel_on_the_growing_range\": %.2f, \"h2d_pinned_into_the_growing_range\": %.2f, "
                    "\"memset_in_the_growing_range\": %.2f, \"d2h_from_the_growing_range\": %.2f}}\n",
                    label, total / 1e9, n_chunks, grow_ms, slowest_chunk, slowest_at, iters, mx_launch, mx_h2d_page, mx_h2d_pin, mx_event, mx_malloc, mx_vmm_kernel,
                    mx_vmm_h2d, mx_vmm_memset, mx_vmm_d2h);
    } else if (mode == "concurrent") {
        const char* label = argc > 2 ? argv[2] : "";
        const size_t s = (size_t)std::atoi(argv[3]) * GB;
        const size_t wb = 2 * GB;
        void *a = nullptr, *b = nullptr;
        CK(hipMalloc(&a, wb));
        CK(hipMalloc(&b, wb));
        touch(a, wb, st), touch(b, wb, st);
        std::atomic<int> state{0}; // 0 idle, 1 allocating, 2 done
        double alloc_ms = 0, launch_gap_max_ms = 0;
        void* big = nullptr;
        std::thread helper([&] {
            CK(hipSetDevice(0));
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            state = 1;
            const auto t0 = clk::now();
            hipError_t e = hipMalloc(&big, s);
            alloc_ms = ms_since(t0);
            if (e != hipSuccess) big = nullptr;
            state = 2;
        });
        // the main thread keeps a queue of streaming kernels going and times each one with events
        std::vector<double> before, during, after;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        int after_n = 0;
        while (after_n < 10) {
            const int s0 = state.load();
            const auto tl = clk::now();
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, st, (const uint4*)a, (uint4*)b, wb / 16);
            CK(hipEventRecord(e1, st));
            const double launch_ms = ms_since(tl);
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const int s1 = state.load();
            if (s0 == 1 || s1 == 1) launch_gap_max_ms = std::max(launch_gap_max_ms, launch_ms);
            (s0 == 0 && s1 == 0 ? before : (s0 == 2 ? after : during)).push_back(ms);
            if (s0 == 2) ++after_n;
        }
        helper.join();
        auto stat = [](const std::vector<double>& v, double& mean, double& mx) {
            mean = 0, mx = 0;
            for (double x : v) mean += x, mx = std::max(mx, x);
            if (!v.empty()) mean /= v.size();
        };
        double m0, x0, m1, x1, m2, x2;
        stat(before, m0, x0), stat(during, m1, x1), stat(after, m2, x2);
        const double first_touch = big ? touch(big, s, st) : -1.0;
        std::printf("{\"mode\": \"concurrent\", \"label\": \"%s\", \"alloc_gb\": %.1f, \"helper_hipMalloc_ms\": %.2f, \"kernel_ms_before\": [%.3f, %.3f, %zu], "
                    "\"kernel_ms_during\": [%.3f, %.3f, %zu], \"kernel_ms_after\": [%.3f, %.3f, %zu], \"slowest_launch_call_during_ms\": %.3f, \"first_touch_ms\": %.2f, "
                    "\"note\": \"[mean, max, n] of a 2 GB read + 2 GB write kernel\"}\n",
                    label, s / 1e9, alloc_ms, m0, x0, before.size(), m1, x1, during.size(), m2, x2, after.size(), launch_gap_max_ms, first_touch);
        if (big) CK(hipFree(big));
        CK(hipFree(a));
        CK(hipFree(b));
    }
    std::fflush(stdout);
    return 0;
}
