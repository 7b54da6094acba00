// Does a wave64 VALU instruction with an all-zero EXEC half issue faster on gfx950 (SIMD-32, two passes of 32 lanes)?
// No: 17.73 ms with all 64 lanes, 17.36 / 17.39 / 17.34 ms with lanes 0-31 / the even lanes / lanes 0-15 only (MI355X,
// round 2).  Packing the live rays of a half-empty wave into one half would buy nothing: k_intersect's lane utilisation
// (0.52) can only be raised by keeping more lanes busy, not by arranging the idle ones.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o half_wave_rate scripts/micro/half_wave_rate.hip && ./half_wave_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float m, float c, int iters) {
    const int lane = threadIdx.x & 63;
    float a = threadIdx.x, b = a + 1.0f, d = a + 2.0f, e = a + 3.0f;
    const bool on = MODE == 0 ? true : (MODE == 1 ? lane < 32 : (MODE == 2 ? (lane & 1) == 0 : lane < 16));
    if (on) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { a = a * m; b = b + c; d = d * m; e = e + c; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + d + e;
}
template <int MODE> float run(float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, 0.999f, 0.001f, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(2048), dim3(256), 0, 0, d, 0.999f, 0.001f, 20000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; (void)hipMalloc(&d, 2048 * 256 * 4);
    printf("all 64 lanes      %.3f ms\n", run<0>(d));
    printf("lanes 0-31 only   %.3f ms\n", run<1>(d));
    printf("even lanes only   %.3f ms\n", run<2>(d));
    printf("lanes 0-15 only   %.3f ms\n", run<3>(d));
    return 0;
}
