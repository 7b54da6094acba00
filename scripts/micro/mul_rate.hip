// Microbenchmark: measured VALU issue rates on gfx950, the denominator of the VALU companion roofline
// (scripts/collect_valu.py, bench.py roofline.valu).  MI355X_MICROARCH.md: SIMD-32, a wave64 v_fma_f32 issues in
// 2 cycles once >= 2 waves share the SIMD (4 for one wave alone), i.e. 1024 SIMDs x 32 lanes x 2.4 GHz = 78.6 T
// lane-ops/s = 157.3 TFLOP/s.  This program measures that rate at 1, 2, 4 and 8 waves per SIMD, and the rates of the
// other instruction classes the trace kernels are made of (integer multiply of the counter RNG, min/max/compare/select
// of the slab test, the quarter-rate v_rcp_f32 / v_sqrt_f32 inside IEEE division and square root).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o mul_rate scripts/micro/mul_rate.hip && ./mul_rate [out.json]
// (-fno-slp-vectorize: otherwise the four scalar chains are packed into v_pk_* pairs and the row measures those)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <string>
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); // shader clock / 100 MHz
    uint32_t a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x9E3779B9u, d = b + 77u;
    float fa = (float)a, fb = 1.0001f, fc = 0.5f, fd = 0.25f;
    const float fm = 1.0f - 1e-7f * (float)(seed & 3u), fk = 1e-3f * (float)seed; // wave-uniform, not compile-time constants
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa = {fa, fb}, pb = {fc, fd};
    const f2 pm = {fm, fm}, pk = {fk, fk};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { a = a * b; b = b * c; c = c * d; d = d * a; } // variable operands: no constant folding
            if (OP == 1) { a = a + 0x85EBCA6Bu ^ b; b = b + 0xC2B2AE35u ^ c; c = c + 0x9E3779B9u ^ d; d = d + 0x85EBCA6Bu ^ a; }
            if (OP == 2) { fa = __builtin_fmaf(fa, fm, fk); fb = __builtin_fmaf(fb, fm, fk); fc = __builtin_fmaf(fc, fm, fk); fd = __builtin_fmaf(fd, fm, fk); } // 4 independent chains
            if (OP == 8) { fa = fa * fm; fb = fb + fk; fc = fc * fm; fd = fd + fk; }                       // v_mul_f32 / v_add_f32, the -ffp-contract=off mix
            if (OP == 9) { pa = __builtin_elementwise_fma(pa, pm, pk); pb = __builtin_elementwise_fma(pb, pm, pk); } // v_pk_fma_f32: 2 lanes-ops per instruction and lane
            if (OP == 10) { pa = pa * pm; pb = pb + pk; }                                                   // v_pk_mul_f32 / v_pk_add_f32
            if (OP == 4) { fa = __builtin_amdgcn_rcpf(fa) + fb; fb = __builtin_amdgcn_rcpf(fb) + fc; fc = __builtin_amdgcn_rcpf(fc) + fd; fd = __builtin_amdgcn_rcpf(fd) + fa; } // rcp + add
            if (OP == 5) { fa = __builtin_amdgcn_sqrtf(fa) + fb; fb = __builtin_amdgcn_sqrtf(fb) + fc; fc = __builtin_amdgcn_sqrtf(fc) + fd; fd = __builtin_amdgcn_sqrtf(fd) + fa; } // sqrt + add
            if (OP == 3) { a = __umul24(a, b) + c; b = __umul24(b, c) + d; c = __umul24(c, d) + a; d = __umul24(d, a) + b; }
            if (OP == 6) { fa = fminf(fa, fb) + fc; fb = fmaxf(fb, fc) + fd; fc = fminf(fc, fd) + fa; fd = fmaxf(fd, fa) + fb; } // min/max + add (2 ops)
            if (OP == 7) { fa = fa < fb ? fc : fd; fb = fb < fc ? fd : fa; fc = fc < fd ? fa : fb; fd = fd < fa ? fb : fc; } // v_cmp + v_cndmask (2 ops)
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = __builtin_amdgcn_s_memtime() - t0, clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ __float_as_uint(fa + fb + fc + fd + pa.x + pa.y + pb.x + pb.y);
}
// `waves` waves per SIMD: 256 CUs x waves workgroups of 256 threads (one wave per SIMD each), all resident at once
template <int OP>
double run(uint32_t* d, int iters, int waves, int ops_per_stmt, double* ghz) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int grid = 256 * waves;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1u, 10, (unsigned long long*)(d + 256 * 8 * 256));
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1u, iters, (unsigned long long*)(d + 256 * 8 * 256));
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2] = {0, 1};
    (void)hipMemcpy(c, d + 256 * 8 * 256, 16, hipMemcpyDeviceToHost);
    *ghz = (double)c[0] / (double)(c[1] ? c[1] : 1) * 0.1; // in-kernel clock: d(s_memtime) / d(s_memrealtime) x 100 MHz
    return (double)grid * 256 * iters * 16 * 4 * ops_per_stmt / ms / 1e9; // T lane-ops/s
}
int main(int argc, char** argv) {
    uint32_t* d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4 + 64);
    const int iters = 20000;
    std::string js = "{\"unit\": \"T lane-ops/s\", \"model_peak\": 78.64, \"model\": \"1024 SIMDs x 32 lanes x 2.4 GHz (one wave64 instruction per 2 cycles per SIMD)\", \"rates\": {";
    const int W[4] = {1, 2, 4, 8};
    char buf[512];
    printf("%-26s %8s %8s %8s %8s   (T lane-ops/s at 1/2/4/8 waves per SIMD; a v_pk_* instruction is 2 lane-ops)\n", "instruction", "1", "2", "4", "8");
#define ROW(OP, NAME, N)                                                                      \
    {                                                                                         \
        double r[4], g[4];                                                                    \
        for (int w = 0; w < 4; ++w) r[w] = run<OP>(d, iters, W[w], N, &g[w]);                 \
        printf("%-26s %8.2f %8.2f %8.2f %8.2f   clock at 8 waves %.2f GHz -> %.2f cycles per wave64 instruction\n", NAME, r[0], r[1], r[2], r[3], g[3], \
               1024.0 * 64.0 * g[3] * 1e9 / (r[3] * 1e12));                                   \
        snprintf(buf, sizeof buf, "%s\"%s\": {\"rate\": [%.3f, %.3f, %.3f, %.3f], \"clock_ghz\": [%.3f, %.3f, %.3f, %.3f]}", js.back() == '{' ? "" : ", ", NAME, r[0], r[1], r[2], r[3], g[0], g[1], g[2], g[3]); \
        js += buf;                                                                            \
    }
    ROW(2, "v_fma_f32", 1)
    ROW(8, "v_mul_f32/v_add_f32", 1)
    ROW(9, "v_pk_fma_f32", 1)
    ROW(10, "v_pk_mul_f32/v_pk_add_f32", 1)
    ROW(1, "v_add_u32+v_xor_b32", 2)
    ROW(0, "v_mul_lo_u32", 1)
    ROW(3, "v_mad_u32_u24", 1)
    ROW(6, "v_min/max_f32+v_add_f32", 2)
    ROW(7, "v_cmp_f32+v_cndmask", 2)
    ROW(4, "v_rcp_f32+v_add_f32", 2)
    ROW(5, "v_sqrt_f32+v_add_f32", 2)
    js += "}, \"waves_per_simd\": [1, 2, 4, 8]}";
    printf("%s\n", js.c_str());
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (f) fprintf(f, "%s\n", js.c_str()), fclose(f);
    }
    return 0;
}
