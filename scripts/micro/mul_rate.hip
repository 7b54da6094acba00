// Microbenchmark: issue rate of v_mul_lo_u32 against v_add_u32 / v_fma_f32 / v_mul_u32_u24, and of the transcendental
// v_rcp_f32 / v_sqrt_f32 (each paired with one full-rate add), on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o mul_rate scripts/micro/mul_rate.hip && ./mul_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x9E3779B9u, d = b + 77u;
    float fa = (float)a, fb = 1.0001f, fc = 0.5f, fd = 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { a = a * b; b = b * c; c = c * d; d = d * a; } // variable operands: no constant folding
            if (OP == 1) { a = a + 0x85EBCA6Bu ^ b; b = b + 0xC2B2AE35u ^ c; c = c + 0x9E3779B9u ^ d; d = d + 0x85EBCA6Bu ^ a; }
            if (OP == 2) { fa = fa * fb + fc; fb = fb * fc + fd; fc = fc * fd + fa; fd = fd * fa + fb; }
            if (OP == 4) { fa = __builtin_amdgcn_rcpf(fa) + fb; fb = __builtin_amdgcn_rcpf(fb) + fc; fc = __builtin_amdgcn_rcpf(fc) + fd; fd = __builtin_amdgcn_rcpf(fd) + fa; } // rcp + add
            if (OP == 5) { fa = __builtin_amdgcn_sqrtf(fa) + fb; fb = __builtin_amdgcn_sqrtf(fb) + fc; fc = __builtin_amdgcn_sqrtf(fc) + fd; fd = __builtin_amdgcn_sqrtf(fd) + fa; } // sqrt + add
            if (OP == 3) { a = __umul24(a, b) + c; b = __umul24(b, c) + d; c = __umul24(c, d) + a; d = __umul24(d, a) + b; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ __float_as_uint(fa + fb + fc + fd);
}
template <int OP>
float run(uint32_t* d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 32), dim3(256), 0, 0, d, 1u, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 32), dim3(256), 0, 0, d, 1u, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    uint32_t* d;
    (void)hipMalloc(&d, 256 * 32 * 256 * 4);
    const int iters = 2000;
    const double ops = 256.0 * 32 * 256 * iters * 16 * 4;
    const char* names[6] = {"v_mul_lo_u32", "v_add+xor (2 ops)", "v_fma_f32", "v_mad_u32_u24", "v_rcp_f32 + v_add", "v_sqrt_f32 + v_add"};
    float ms[6] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters)};
    for (int i = 0; i < 6; ++i) printf("%-20s %8.3f ms  %8.2f Tops/s (lane-ops)\n", names[i], ms[i], ops / ms[i] / 1e9);
    return 0;
}
