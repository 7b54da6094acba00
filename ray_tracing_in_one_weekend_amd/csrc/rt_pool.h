// rt_pool.h — the work buffers of a context (two ray queues, hit records, radiance slots: 100 B per ray of a slice, 53 GB for
// config 2 in one slice) as ONE virtual range that a helper thread backs with physical memory chunk by chunk.
//
// Why (round 6; scripts/micro/alloc_probe.hip, profiles/round6/first_frame_states.txt).  The reference renders one frame per
// process (main.rs:62-129), so the first rt_render is the render, and its allocations are on the clock.  On a quiet device
// hipMalloc returns in 0.3 ms whatever the size.  On a device whose memory other processes — or this one — have freed in the last
// seconds, ONE request out of a few waits 3-6 s inside the driver (a 1 GB request as readily as a 100 GB one: the driver is
// taking freed memory back, and the request that reaches into it waits for that), and nothing a process does shortens the wait.
// What a renderer can do is not stand in it:
//   * the range is reserved once (hipMemAddressReserve: 10 us) and grown by a thread of its own in 128 MB chunks
//     (hipMemCreate + hipMemMap + hipMemSetAccess: 15 us a chunk, 67 GB in ~10 ms on a quiet device; kernels run at the same
//     bandwidth on mapped memory as on hipMalloc'ed memory, measured).  ONE chunk size: on ROCm 7.2 hipMemSetAccess returns
//     "invalid argument" for a mapping whose size differs from that of the first mapping of the reservation
//     (alloc_probe vmmmix), so a smaller fallback size for a device that is nearly full is not an option;
//   * rt_prepare() starts the growth for a frame size before the scene exists, so that it overlaps the host's scene build;
//   * render_impl sizes every slice by what is mapped when the slice is enqueued: while the pool is still growing the frame
//     starts in small slices instead of waiting (slices are independent and frames bit-identical for any slicing), a stalled
//     chunk delays the growth and not the frame, and a device that has little memory to give yields more slices, not an error.
// The pool only grows; it is kept across frames and scene uploads and released by rt_ctx_destroy.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace rt {

struct WorkPool {
#ifndef RT_POOL_CHUNK_MB
#define RT_POOL_CHUNK_MB 128
#endif
    static constexpr size_t CHUNK = (size_t)RT_POOL_CHUNK_MB << 20;
    int device = 0;
    char* base = nullptr;
    size_t reserved = 0;
    struct Chunk {
        hipMemGenericAllocationHandle_t h;
        size_t at, bytes;
    };
    std::vector<Chunk> chunks; // grower thread; the owner after joining it
    std::atomic<size_t> mapped{0};                       // bytes of [base, base + mapped) that kernels may touch
    std::mutex mu;
    std::condition_variable cv;
    // guarded by mu
    size_t target = 0;
    bool stop = false, running = false, exhausted = false;
    std::string err;
    std::thread th;
    // statistics of the growth (rt_debug_render_parts)
    std::atomic<unsigned> chunk_delay_us{0}; // RT_OPT_POOL_CHUNK_DELAY_US (test hook: a device that hands out memory slowly)
    std::atomic<unsigned> n_grown{0};
    std::atomic<double> slowest_chunk_ms{0.0};
};

inline void pool_grower(WorkPool* p) {
    (void)hipSetDevice(p->device);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = p->device;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t chunk = WorkPool::CHUNK;
    for (;;) {
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (p->stop || p->mapped.load() >= p->target) {
                p->running = false;
                p->cv.notify_all();
                return;
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        const size_t at = p->mapped.load();
        if (const unsigned us = p->chunk_delay_us.load()) std::this_thread::sleep_for(std::chrono::microseconds(us));
        hipMemGenericAllocationHandle_t h{};
        hipError_t e = hipMemCreate(&h, chunk, &prop, 0);
        const char* what = "hipMemCreate";
        if (e == hipSuccess) {
            e = hipMemMap(p->base + at, chunk, 0, h, 0);
            what = "hipMemMap";
            if (e == hipSuccess) {
                e = hipMemSetAccess(p->base + at, chunk, &acc, 1);
                what = "hipMemSetAccess";
                if (e != hipSuccess) (void)hipMemUnmap(p->base + at, chunk);
            }
            if (e != hipSuccess) (void)hipMemRelease(h);
        }
        if (e != hipSuccess) { // out of device memory (or of mappings): the pool stays at what it has
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lk(p->mu);
            p->exhausted = true;
            p->err = std::string(what) + ": " + hipGetErrorString(e);
            p->running = false;
            p->cv.notify_all();
            return;
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > p->slowest_chunk_ms.load()) p->slowest_chunk_ms.store(ms);
        p->n_grown.fetch_add(1u);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->chunks.push_back(WorkPool::Chunk{h, at, chunk});
            p->mapped.store(at + chunk);
        }
        p->cv.notify_all();
    }
}

// Asks for a pool of at least `bytes` (rounded up to chunks, capped by the reservation) and returns at once; the grower thread is
// started if it is not running.  A pool that stopped at the device's limit earlier tries again: memory may have come back.
// Returns the error text of a failed reservation, or nullptr.
inline const char* pool_request(WorkPool& p, size_t bytes) {
    constexpr size_t RESERVE = 160ull << 30; // RT_MAX_SLICE_RAYS x 100 B + the small buffers + rounding: the largest slice render_impl ever asks for
    if (!p.base) {
        void* va = nullptr;
        if (hipMemAddressReserve(&va, RESERVE, 2ull << 20, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            return "hipMemAddressReserve failed";
        }
        p.base = (char*)va, p.reserved = RESERVE;
    }
    bytes = std::min(((bytes + WorkPool::CHUNK - 1) / WorkPool::CHUNK) * WorkPool::CHUNK, p.reserved);
    std::unique_lock<std::mutex> lk(p.mu);
    if (bytes <= p.target && !(p.exhausted && p.mapped.load() < bytes)) return nullptr;
    p.target = std::max(p.target, bytes);
    p.exhausted = false;
    if (!p.running) {
        if (p.th.joinable()) {
            lk.unlock();
            p.th.join();
            lk.lock();
        }
        p.running = true;
        p.th = std::thread(pool_grower, &p);
    }
    return nullptr;
}

// Blocks until `bytes` are mapped or the grower has given up; true when they are.
inline bool pool_wait(WorkPool& p, size_t bytes) {
    std::unique_lock<std::mutex> lk(p.mu);
    p.cv.wait(lk, [&] { return p.mapped.load() >= bytes || !p.running; });
    return p.mapped.load() >= bytes;
}

// Waits for the mapped size to change, the grower to end, or `us` microseconds.
inline void pool_wait_progress(WorkPool& p, size_t seen, unsigned us) {
    std::unique_lock<std::mutex> lk(p.mu);
    p.cv.wait_for(lk, std::chrono::microseconds(us), [&] { return p.mapped.load() != seen || !p.running; });
}

inline bool pool_growing(WorkPool& p) {
    std::lock_guard<std::mutex> lk(p.mu);
    return p.running;
}

inline std::string pool_error(WorkPool& p) {
    std::lock_guard<std::mutex> lk(p.mu);
    return p.err;
}

// Stops the grower (a chunk request that is waiting inside the driver is waited for: there is no way to withdraw it), unmaps and
// releases everything.  The caller has synchronised the streams that used the memory.
inline void pool_destroy(WorkPool& p) {
    {
        std::lock_guard<std::mutex> lk(p.mu);
        p.stop = true;
    }
    if (p.th.joinable()) p.th.join();
    for (const WorkPool::Chunk& c : p.chunks) {
        (void)hipMemUnmap(p.base + c.at, c.bytes);
        (void)hipMemRelease(c.h);
    }
    p.chunks.clear();
    p.mapped.store(0);
    if (p.base) (void)hipMemAddressFree(p.base, p.reserved);
    p.base = nullptr, p.reserved = 0, p.target = 0, p.stop = false, p.exhausted = false;
}

} // namespace rt
