// Microbenchmark: latencies of DEPENDENT operations on gfx950, one wave alone on a CU, in shader-clock cycles.
// The trace kernels' waves spend their time in chains of such operations (k_shade has no bottleneck
// that throughput counters show); these numbers price a chain.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o latency scripts/micro/latency.hip && ./latency [out.json]
// Every row is cycles per operation of a chain in which each operation needs the result of the one before:
//   valu_add / valu_mul_lo : v_add_f32, v_mul_lo_u32 (the counter RNG's multiply)
//   lds_read               : idx = lds[idx], each lane its own chain (ds_read_b32)
//   bpermute               : x = ds_bpermute(f(x), x): the cross-lane exchange of __shfl and of the cooperative sampler
//   lds_atomic_same/_own   : ds_add_rtn_u32 of all 64 lanes on ONE address (the class counters of k_shade's sort) / one address per lane
//   gather_<footprint>     : idx = buf[idx] in global memory, 64 lanes on 64 different 128 B lines, the chain confined to
//                            16 KB (vector L1), 2 MB (L2), 64 MB (MALL) or 2 GB (HBM)
//   single_<footprint>     : the same with ONE lane active (what the other 63 lines of a gather cost)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

constexpr int UNROLL = 32;

__global__ void k_fill(uint32_t* buf, uint32_t n, uint32_t stride) { // buf[i] = (i + stride) mod n, n a power of two
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        buf[i] = (uint32_t)((i + stride) & (n - 1u));
}

template <int OP>
__global__ __launch_bounds__(64) void k_lat(const uint32_t* __restrict__ buf, uint32_t n, int iters, int lanes,
                                            unsigned long long* out, uint32_t* sink) {
    __shared__ uint32_t lds[1024];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024u; i += 64u) lds[i] = (i * 37u + 64u) & 1023u; // a permutation with long cycles
    __syncthreads();
    uint32_t x = lane * 17u + 3u;
    float f = (float)lane;
    const float fk = 1e-3f * (float)(iters & 7);
    uint32_t idx = (uint32_t)(((size_t)lane * (n / 64u)) & (n - 1u));
    const bool on = (int)lane < lanes;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (OP == 0) f = f + fk;
            if (OP == 1) x = x * (x | 1u);
            if (OP == 2) x = lds[x & 1023u];
            if (OP == 3) x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((x + 1u) & 63u) << 2), (int)(x + 1u));
            if (OP == 4) x = atomicAdd(&lds[x >> 31], 1u);          // every lane on lds[0] (x stays small)
            if (OP == 5) x = atomicAdd(&lds[lane + (x >> 31)], 1u); // one counter per lane
            if (OP == 6) {
                if (on) idx = buf[idx];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = t1 - t0;
    sink[blockIdx.x * 64u + lane] = x ^ idx ^ __float_as_uint(f);
}

template <int OP>
double run(const uint32_t* buf, uint32_t n, int iters, int lanes, unsigned long long* d_out, uint32_t* d_sink) {
    hipLaunchKernelGGL(k_lat<OP>, dim3(1), dim3(64), 0, 0, buf, n, 2, lanes, d_out, d_sink); // warm-up (code, TLB)
    hipLaunchKernelGGL(k_lat<OP>, dim3(1), dim3(64), 0, 0, buf, n, iters, lanes, d_out, d_sink);
    (void)hipDeviceSynchronize();
    unsigned long long c = 0;
    (void)hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
    return (double)c / ((double)iters * UNROLL);
}

int main(int argc, char** argv) {
    const size_t max_elems = (size_t)1 << 29; // 2 GB of uint32
    uint32_t *buf, *sink;
    unsigned long long* d_out;
    if (hipMalloc(&buf, max_elems * 4) != hipSuccess || hipMalloc(&sink, 64 * 4) != hipSuccess || hipMalloc(&d_out, 8) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    std::string js = "{";
    auto emit = [&](const char* name, double cyc) {
        printf("%-22s %9.1f cycles\n", name, cyc);
        char b[128];
        snprintf(b, sizeof b, "%s\"%s\": %.2f", js.size() > 1 ? ", " : "", name, cyc);
        js += b;
    };
    emit("valu_add", run<0>(buf, 64, 2000, 64, d_out, sink));
    emit("valu_mul_lo", run<1>(buf, 64, 2000, 64, d_out, sink));
    emit("lds_read", run<2>(buf, 64, 500, 64, d_out, sink));
    emit("bpermute", run<3>(buf, 64, 500, 64, d_out, sink));
    emit("lds_atomic_same", run<4>(buf, 64, 200, 64, d_out, sink));
    emit("lds_atomic_own", run<5>(buf, 64, 200, 64, d_out, sink));
    struct {
        const char* name;
        size_t elems;
        int iters;
    } fp[] = {{"16KB", (size_t)1 << 12, 200}, {"2MB", (size_t)1 << 19, 200}, {"64MB", (size_t)1 << 24, 100}, {"2GB", max_elems, 50}};
    for (auto& f : fp) {
        // stride: an odd multiple of 32 elements (one 128 B line) that walks the whole footprint before it repeats
        const uint32_t stride = (uint32_t)(((f.elems / 64u / 7u) | 1u) * 32u + 32u) & (uint32_t)(f.elems - 1u);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, (uint32_t)f.elems, stride ? stride : 32u);
        (void)hipDeviceSynchronize();
        char nm[32];
        snprintf(nm, sizeof nm, "gather_%s", f.name);
        emit(nm, run<6>(buf, (uint32_t)f.elems, f.iters, 64, d_out, sink));
        snprintf(nm, sizeof nm, "single_%s", f.name);
        emit(nm, run<6>(buf, (uint32_t)f.elems, f.iters, 1, d_out, sink));
    }
    js += "}";
    if (argc > 1) {
        FILE* fo = fopen(argv[1], "w");
        if (fo) {
            fprintf(fo, "%s\n", js.c_str());
            fclose(fo);
        }
    }
    return 0;
}
