// Microbenchmark: what rocprofv3's FETCH_SIZE / WRITE_SIZE (and the raw TCC_EA0_* request counters behind them) report on
// gfx950 for the access patterns of the trace kernels, against a KNOWN number of requested bytes and touched 128 B lines.
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (it reports 1/2); k_shade's class-ordered
// gathers take 16 / 32 / 8 B pieces out of scattered lines and scripts/collect_traffic.py applied the same x2 to them.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/fetch_gather scripts/micro/fetch_gather.hip
//   scripts/fetch_calibration.py runs it under rocprofv3 (separate --pmc passes) and writes profiles/round3/fetch_calibration.json
// Every kernel touches each 128 B line of a 4 GiB buffer (16x the 256 MiB Infinity Cache) exactly ONCE; gathers visit the
// lines in a pseudo-random order (line = i * odd mod 2^25, a bijection), so consecutive lanes are ~GBs apart.
//   rd_stream16 / 8 / 4   coalesced, 16 / 8 / 4 B per lane (the whole buffer; k_resolve, hit records, class table)
//   rd_stream32           coalesced 32 B records, two 16 B loads per lane (k_intersect reading the ray queue)
//   rd_gather16           one 16 B piece at offset 0 of every line
//   rd_gather32           32 B record at offset 0 (two 16 B loads by the same lane: k_shade's ray gather)
//   rd_gather16x2_far     16 B at offset 0 and at offset 64 of the same line, same lane, back to back: tells whether a miss
//                         fetches the 128 B line (same request count as rd_gather16) or a 64 B sector (twice as many)
//   rd_gather16x2_mid     16 B at offsets 0 and 32
//   rd_gather8 / 4        8 B (qc, hit record by position) / 4 B (RGBA8 texel)
//   wr_stream16 / 8       coalesced stores
//   wr_scatter16 / 12 / 32 / 8   one 16 B / 12 B (radiance slot) / 32 B (survivor record) / 8 B store per line at offset 0
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

constexpr uint32_t LINE_BITS = 25;             // 2^25 lines x 128 B = 4 GiB
constexpr uint32_t N_LINES = 1u << LINE_BITS;
constexpr uint32_t MULT = 0x9E3779B1u;         // odd: i -> i * MULT mod 2^25 is a bijection

__device__ __forceinline__ uint32_t line_of(uint32_t i) { return (i * MULT) & (N_LINES - 1u); }

template <class T>
__global__ __launch_bounds__(256) void rd_stream(const T* __restrict__ buf, size_t n, float* sink) {
    float acc = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = buf[i];
        acc += *reinterpret_cast<const float*>(&v);
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void rd_stream32(const float4* __restrict__ buf, size_t n_rec, float* sink) {
    float acc = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_rec; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = buf[2 * i], b = buf[2 * i + 1];
        acc += a.x + b.w;
    }
    if (acc == 12345.678f) *sink = acc;
}
// PIECES 16-byte loads at byte offsets OFF0 and OFF1 of each line (PIECES = 1: OFF0 only)
template <int PIECES, int OFF0, int OFF1>
__global__ __launch_bounds__(256) void rd_gather16(const char* __restrict__ buf, float* sink) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N_LINES; i += gridDim.x * blockDim.x) {
        const char* p = buf + (size_t)line_of(i) * 128u;
        const float4 a = *reinterpret_cast<const float4*>(p + OFF0);
        acc += a.x;
        if (PIECES == 2) {
            const float4 b = *reinterpret_cast<const float4*>(p + OFF1);
            acc += b.w;
        }
    }
    if (acc == 12345.678f) *sink = acc;
}
template <class T>
__global__ __launch_bounds__(256) void rd_gather_small(const char* __restrict__ buf, float* sink) {
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N_LINES; i += gridDim.x * blockDim.x) {
        const T v = *reinterpret_cast<const T*>(buf + (size_t)line_of(i) * 128u);
        acc += *reinterpret_cast<const float*>(&v);
    }
    if (acc == 12345.678f) *sink = acc;
}
template <class T>
__global__ __launch_bounds__(256) void wr_stream(T* __restrict__ buf, size_t n) {
    T v{};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = v;
}
// BYTES per line at offset 0: 8 (float2), 12 (float3: global_store_dwordx3), 16 (float4), 32 (two float4)
template <int BYTES>
__global__ __launch_bounds__(256) void wr_scatter(char* __restrict__ buf) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N_LINES; i += gridDim.x * blockDim.x) {
        char* p = buf + (size_t)line_of(i) * 128u;
        if (BYTES == 8) *reinterpret_cast<float2*>(p) = make_float2(1.f, 2.f);
        if (BYTES == 12) *reinterpret_cast<float3*>(p) = make_float3(1.f, 2.f, 3.f);
        if (BYTES >= 16) *reinterpret_cast<float4*>(p) = make_float4(1.f, 2.f, 3.f, 4.f);
        if (BYTES == 32) *reinterpret_cast<float4*>(p + 16) = make_float4(5.f, 6.f, 7.f, 8.f);
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)N_LINES * 128u;
    char* buf = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 16));
    CHECK(hipMemset(buf, 0, bytes));
    const int reps = argc > 1 ? atoi(argv[1]) : 2;
    const dim3 grid(256 * 16), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"buffer_bytes\": %zu, \"lines\": %u, \"variants\": [\n", bytes, N_LINES);
    bool first = true;
#define RUN(NAME, REQ_BYTES, ...)                                                                                 \
    for (int r = 0; r < reps; ++r) {                                                                              \
        CHECK(hipEventRecord(e0));                                                                                \
        __VA_ARGS__;                                                                                              \
        CHECK(hipEventRecord(e1));                                                                                \
        CHECK(hipEventSynchronize(e1));                                                                           \
        float ms = 0.f;                                                                                           \
        CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                  \
        if (r == reps - 1) {                                                                                      \
            printf("%s {\"name\": \"%s\", \"requested_bytes\": %.0f, \"ms\": %.4f, \"requested_GBps\": %.1f}\n",   \
                   first ? " " : ",", NAME, (double)(REQ_BYTES), ms, (double)(REQ_BYTES) / ms / 1e6);              \
            first = false;                                                                                        \
        }                                                                                                         \
    }
    RUN("rd_stream16", bytes, hipLaunchKernelGGL(rd_stream<float4>, grid, block, 0, 0, (const float4*)buf, bytes / 16, sink))
    RUN("rd_stream8", bytes, hipLaunchKernelGGL(rd_stream<float2>, grid, block, 0, 0, (const float2*)buf, bytes / 8, sink))
    RUN("rd_stream4", bytes, hipLaunchKernelGGL(rd_stream<float>, grid, block, 0, 0, (const float*)buf, bytes / 4, sink))
    RUN("rd_stream32", bytes, hipLaunchKernelGGL(rd_stream32, grid, block, 0, 0, (const float4*)buf, bytes / 32, sink))
    RUN("rd_gather16", (double)N_LINES * 16, hipLaunchKernelGGL((rd_gather16<1, 0, 0>), grid, block, 0, 0, buf, sink))
    RUN("rd_gather32", (double)N_LINES * 32, hipLaunchKernelGGL((rd_gather16<2, 0, 16>), grid, block, 0, 0, buf, sink))
    RUN("rd_gather16x2_mid", (double)N_LINES * 32, hipLaunchKernelGGL((rd_gather16<2, 0, 32>), grid, block, 0, 0, buf, sink))
    RUN("rd_gather16x2_far", (double)N_LINES * 32, hipLaunchKernelGGL((rd_gather16<2, 0, 64>), grid, block, 0, 0, buf, sink))
    RUN("rd_gather8", (double)N_LINES * 8, hipLaunchKernelGGL(rd_gather_small<float2>, grid, block, 0, 0, buf, sink))
    RUN("rd_gather4", (double)N_LINES * 4, hipLaunchKernelGGL(rd_gather_small<float>, grid, block, 0, 0, buf, sink))
    RUN("wr_stream16", bytes, hipLaunchKernelGGL(wr_stream<float4>, grid, block, 0, 0, (float4*)buf, bytes / 16))
    RUN("wr_stream8", bytes, hipLaunchKernelGGL(wr_stream<float2>, grid, block, 0, 0, (float2*)buf, bytes / 8))
    RUN("wr_scatter16", (double)N_LINES * 16, hipLaunchKernelGGL(wr_scatter<16>, grid, block, 0, 0, buf))
    RUN("wr_scatter12", (double)N_LINES * 12, hipLaunchKernelGGL(wr_scatter<12>, grid, block, 0, 0, buf))
    RUN("wr_scatter32", (double)N_LINES * 32, hipLaunchKernelGGL(wr_scatter<32>, grid, block, 0, 0, buf))
    RUN("wr_scatter8", (double)N_LINES * 8, hipLaunchKernelGGL(wr_scatter<8>, grid, block, 0, 0, buf))
    printf("]}\n");
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
