// rt_device.h — gfx950 device functions of the wavefront path tracer: fp32 vector math in the
// reference's (glam 0.21) operation order, the counter-based RNG, sphere intersection,
// textures, sky models and all 13 materials.  Citations are to /root/reference/src.
//
// Build with -ffp-contract=off: the reference never fuses a*b+c, and with fusion disabled
// every +,-,*,/ and sqrt below is IEEE-exact, so ray geometry (hit points, scatter
// directions, branch decisions) is bit-identical to a CPU evaluation of the same formulas;
// only libm-class functions (sin, cos, acos, atan2, log) may differ in the last ulp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 splat(float s) { return V3{s, s, s}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ V3 operator+(V3 a, float s) { return V3{a.x + s, a.y + s, a.z + s}; }
__device__ __forceinline__ V3 operator-(float s, V3 a) { return V3{s - a.x, s - a.y, s - a.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// glam dot3: (x*x' + y*y') + z*z'
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float length_squared(V3 a) { return dot(a, a); }
// ---------------------------------------------------------------------------------------------
// x / a for many x and ONE a, correctly rounded.  What the compiler emits for an fp32 division on gfx9 is
//     y = rcp(a'); y += y * (1 - a' y);  q = x' y;  q += (x' - a' q) y;  v_div_fmas(x' - a' q, y, q);  v_div_fixup
// with a', x' = v_div_scale of the operands — powers of two that are 1 unless an exponent is extreme (a denormal or beyond 2^126,
// a quotient denormal or beyond 2^96, a numerator below 2^-103).  Outside those cases the same fused multiply-adds on the unscaled
// operands return the same bits, and the refined reciprocal `y` (a quarter-rate v_rcp and two fmas) depends on `a` alone: a ray's
// |d|^2 divides the roots of every sphere it is tested against, a sphere's radius the three components of its normal.  Five
// instructions and the fixup per quotient instead of eleven with a v_rcp.  v_div_fixup is the compiler's own last step: zeros keep
// their sign, infinities and NaNs come out as IEEE wants them.  The extreme cases cannot carry a result here: a root below 2^-103 fails
// `root < t_min` whatever its last bit, unit directions have a = 1 +- 1e-6, radii are ordinary numbers (rt_scene_upload rejects
// non-finite geometry; a direction that is not unit is dropped by k_shade as main.rs:39 would panic).  Held by every bit-exact
// comparison of t, o and d against the oracle's IEEE divisions, and against the list walk, which keeps the `/` operator.
// ---------------------------------------------------------------------------------------------
struct SharedRcp {
    float a, y;
};
__device__ __forceinline__ SharedRcp shared_rcp(float a) {
    const float y0 = __builtin_amdgcn_rcpf(a);
    const float e = __builtin_fmaf(-a, y0, 1.0f);
    return SharedRcp{a, __builtin_fmaf(e, y0, y0)};
}
__device__ __forceinline__ float div_shared(float x, const SharedRcp& r) {
    float q = x * r.y;
    float e = __builtin_fmaf(-r.a, q, x);
    q = __builtin_fmaf(e, r.y, q);
    e = __builtin_fmaf(-r.a, q, x);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(e, r.y, q), r.a, x);
}

// sqrt(x), correctly rounded, as the compiler emits it for fp32 — v_sqrt_f32 (one ulp), then the neighbours s -+ 1 ulp tried with
// two exact residuals x - s' s, then zeros and +inf passed through — without the 2^32 scaling it wraps around that for arguments
// below 2^-96 (where the residuals would be denormal): 11 instructions instead of 16, two compare + select pairs (4 cycles each) fewer.
// The same bits for every argument from 2^-96 up, for 0, inf, NaN and negative numbers.  Used where the argument cannot lie in
// (0, 2^-96): a discriminant hb^2 - a c is a difference of two rounded numbers, zero or at least an ulp of them; the squared length
// of a direction, of a rejection sample (coordinates k 2^-23) or of a sum / difference of unit vectors is zero or at least 2^-50.
// Cross products (a tangent at a sphere's pole) keep sqrtf.  Held bit for bit against numpy by rt_debug_arithmetic's test, and by
// every comparison of t, o and d with the oracle.
__device__ __forceinline__ float sqrt_ns(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = 0.0f >= r_dn ? s_dn : s;
    r = 0.0f < r_up ? s_up : r;
    return __builtin_amdgcn_classf(x, 0x260) ? x : r; // -0, +0, +inf
}
__device__ __forceinline__ float length(V3 a) { return sqrt_ns(dot(a, a)); }
// glam normalize: v * (1 / length)
__device__ __forceinline__ V3 normalize(V3 a) {
    float inv = 1.0f / sqrt_ns(dot(a, a)); // (the unscaled division sequence for this single quotient: no gain, ab12 of round 5)
    return a * inv;
}
__device__ __forceinline__ V3 normalize_any(V3 a) { // (a vector whose squared length may be anything: sqrtf)
    float inv = 1.0f / sqrtf(dot(a, a));
    return a * inv;
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 lerp3(V3 a, V3 b, float s) { return a + ((b - a) * s); }

#define RT_PI 3.14159265358979323846f
#define RT_FRAC_1_PI 0.318309886183790671537767526745028724f
#define RT_FLT_MAX 3.402823466e+38f

__device__ __forceinline__ float powi2(float x) { return x * x; }
__device__ __forceinline__ float powi5(float x) { // LLVM powi expansion: x * ((x*x)*(x*x))
    float x2 = x * x;
    float x4 = x2 * x2;
    return x * x4;
}
// Rust f32::clamp keeps NaN
__device__ __forceinline__ float clampf(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
// Rust's `f as u32` / `f as i32`: toward zero, saturating, NaN -> 0 — which is what v_cvt_u32_f32 / v_cvt_i32_f32 do by themselves (CDNA ISA:
// "out-of-range values saturate, NaN is converted to 0").  A C cast promises nothing outside the range, so written with compares it
// compiled to three nested exec-mask regions per conversion (nine in offset_hit_point); the instruction is named instead.
// tests/test_gpu_parity.py holds both against the Rust rule over every kind of argument (rt_debug_arithmetic).
__device__ __forceinline__ uint32_t sat_u32(float f) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
__device__ __forceinline__ int32_t sat_i32(float f) {
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// Lane statistics (diagnostic builds only, -DRT_PROFILE_LANES; scripts/gpu_lane_stats.py): for each counted site,
// [i] += 64 per trip of a wave and [i + 1] += the lanes that were active in it.
//   k_intersect: 0/1 main-loop trips / lanes holding a ray   2/3 node steps   4/5 trips of the leaf loop   6/7 refill blocks
//   shading:     8/9 trips of the rejection loop of random_in_unit_sphere   10/11 calls of it (lanes entering)
//   depth 0:     12/13 trips of the candidate-list test of k_shade<GEN>   14/15 waves at depth 0 / lanes that have a list
//   textures:    16/17 waves entering the Perlin turbulence (texture.rs:115-124) / lanes evaluating it
//   shading:     18/19 64-ray segments shaded by k_shade / lanes holding a ray in them
//   materials:   20/21 + 2 m: waves / lanes entering the branch of material tag m of shade() (Lambert and the pbr.rs tags: their common part)
#ifdef RT_PROFILE_LANES
#define RT_LANE_STAT_N 48
__device__ unsigned long long g_lane_stats[RT_LANE_STAT_N];
#define RT_LANE_STAT(I, PRED)                                                                      \
    do {                                                                                           \
        const unsigned long long act_ = __ballot(true), m_ = __ballot(PRED);                       \
        if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)act_) - 1u) {                      \
            atomicAdd(&g_lane_stats[I], 64ull);                                                    \
            atomicAdd(&g_lane_stats[(I) + 1], (unsigned long long)__popcll(m_));                   \
        }                                                                                          \
    } while (0)
#else
#define RT_LANE_STAT(I, PRED)
#endif

// ---------------------------------------------------------------------------------------------
// Counter-based RNG (DESIGN.md "RNG"); replaces the thread-local SmallRng of lib.rs:7-9.
// draw(k0,k1,ctr) = mix32((k0 ^ ctr*0x9E3779B9) + k1); f32 = (u32 >> 8) * 2^-24 exactly
// as rand's Standard distribution maps a u32 (main.rs:89-90, math.rs:19-21).
// A SplitMix-style generator per path: a keyed Weyl sequence (ctr * golden ratio, scrambled by the path's two 32-bit keys,
// themselves three fmix32 rounds over seed / pixel / sample) through ONE 32-bit finaliser with full avalanche.  Rounds 1-2
// used two (fmix32(fmix32(..) + k1)): the second costs 7.5 % of k_shade (profiles/round3/whatif_shade_costs.txt: the draws
// are quarter-rate integer multiplies) and buys nothing measurable — uniformity, 3-D equidistribution of the ball sampler's
// triples, serial and neighbour-pixel correlation and counter-bit avalanche are the same (tests/test_oracle_kat.py).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
// the per-draw finaliser: "lowbias32" (C. Wellons, hash-prospector, public domain): the two-multiply xorshift-multiply
// construction of fmix32 with the constants of lowest measured avalanche bias
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ void path_key(uint64_t seed, uint32_t pix, uint32_t samp, uint32_t& k0, uint32_t& k1) {
    uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
    uint32_t a = fmix32(pix ^ s_lo);
    k0 = fmix32(a + samp * 0x9E3779B9u + s_hi);
    k1 = fmix32((a ^ 0xA511E9B3u) + samp * 0xC2B2AE3Du);
}
// `w` is the Weyl term ctr * 0x9E3779B9 of the NEXT draw, advanced by an addition: the same values as the product, and one
// quarter-rate integer multiply less per draw (two are left, in mix32).
struct Rng {
    uint32_t k0, k1, w;
    __device__ __forceinline__ Rng() {}
    __device__ __forceinline__ Rng(uint32_t k0_, uint32_t k1_, uint32_t ctr) : k0(k0_), k1(k1_), w(ctr * 0x9E3779B9u) {}
    __device__ __forceinline__ float next() {
        uint32_t r = mix32((k0 ^ w) + k1);
        w += 0x9E3779B9u;
        return (float)(r >> 8) * (1.0f / 16777216.0f);
    }
    // 2 u - 1 for the next draw u, i.e. `u * (1.0 - -1.0) + -1.0` of math.rs:30-35, in one fused multiply-add: with k = r >> 8,
    // u = k 2^-24 and 2 u are exact scalings and k 2^-23 - 1 = (k - 2^23) 2^-23 has a 24-bit numerator, so the reference's sum
    // is exact as well and fma(k, 2^-23, -1) — one rounding of that same exact value — returns it bit for bit (2 instructions
    // per coordinate instead of 4 in the rejection loop that every diffuse bounce runs 1.9 times).
    __device__ __forceinline__ float next_pm1() {
        uint32_t r = mix32((k0 ^ w) + k1);
        w += 0x9E3779B9u;
        return __builtin_fmaf((float)(r >> 8), 1.0f / 8388608.0f, -1.0f);
    }
};
__device__ __forceinline__ uint32_t depth_counter_base(int depth) { return (uint32_t)(depth + 1) * 256u; }

// ---------------------------------------------------------------------------------------------
// math.rs sampling and geometry helpers
// ---------------------------------------------------------------------------------------------
// (Measured with the counters above, profiles/round2/lane_stats_shading.txt: 6.0 trips per wave for 1.9 per lane, at 21 %
// lane utilisation — a third of k_shade's vector instructions.  Two remedies were built and neither pays)
__device__ __forceinline__ V3 random_in_unit_sphere(Rng& rng) { // math.rs:17-37
    RT_LANE_STAT(10, true);
    for (;;) {
        RT_LANE_STAT(8, true);
        const float x = rng.next_pm1(); // vec3a_random() * (1.0 - -1.0) + -1.0, x then y then z (math.rs:19-21, 30-35)
        const float y = rng.next_pm1();
        const float z = rng.next_pm1();
        const V3 v = v3(x, y, z);
        if (length_squared(v) < 1.0f) return v;
    }
}
__device__ __forceinline__ V3 random_on_hemisphere(Rng& rng, V3 n) { // math.rs:43-53
    V3 v = random_in_unit_sphere(rng);
    if (!(dot(v, n) > 0.0f)) v = -v;
    return normalize(v);
}
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - 2.0f * dot(v, n) * n; } // math.rs:68-70
__device__ __forceinline__ V3 refract(V3 uv, V3 n, float etai_over_etat) {             // math.rs:72-77
    float cos_theta = -fminf(dot(uv, n), 1.0f);
    V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V3 r_out_parallel = -sqrtf(fabsf(1.0f - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}
__device__ __forceinline__ float schlick_fresnel(float u) { return powi5(1.0f - u); } // math.rs:79-81
__device__ __forceinline__ float reflectance(float cosine, float ref_idx) {           // math.rs:84-88
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * schlick_fresnel(cosine);
}
__device__ __forceinline__ float lerpf(float from, float to, float s) { return from + (to - from) * s; } // math.rs:154-156
__device__ __forceinline__ float offset_axis(float p, float n) { // one lane of math.rs:137-152
    const float ORIGIN = 1.0f / 32.0f;
    const float INT_SCALE = 256.0f;
    const float FLOAT_SCALE = 1.0f / 65536.0f;
    int32_t of_i = sat_i32(n * INT_SCALE);
    uint32_t bits = __float_as_uint(p) + (uint32_t)(p < 0.0f ? -of_i : of_i);
    float p_i = __uint_as_float(bits);
    return fabsf(p) < ORIGIN ? p + n * FLOAT_SCALE : p_i;
}
__device__ __forceinline__ V3 offset_hit_point(V3 p, V3 n) { // math.rs:137-152
    return v3(offset_axis(p.x, n.x), offset_axis(p.y, n.y), offset_axis(p.z, n.z));
}
// math.rs:13-15: (v.length() - 1.0).abs() < 1e-6.  The square root is correctly rounded and monotone, s - 1 is exact near 1, so the
// squared lengths that pass are an interval of floats: exactly the 50 from 0x3f7fffe0 (0.9999981) to 0x3f800011 (1.000002), found
// by running the expression over every float (tests/test_device_constants.py does it again).  Two comparisons per ray instead of
// a square root; NaN fails both ways.
#define RT_NEAR_ONE_LO 0x3f7fffe0u
#define RT_NEAR_ONE_HI 0x3f800011u
__device__ __forceinline__ bool near_one(V3 d) {
    const float l2 = dot(d, d);
    return l2 >= __uint_as_float(RT_NEAR_ONE_LO) && l2 <= __uint_as_float(RT_NEAR_ONE_HI);
}

// hitable.rs:65-71 Sphere::get_uv
__device__ __forceinline__ V2 sphere_get_uv(V3 n) {
    float theta = acosf(-n.y);
    float phi = atan2f(-n.z, n.x) + RT_PI;
    return V2{phi / (2.0f * RT_PI), theta / RT_PI};
}

// ---------------------------------------------------------------------------------------------
// Packed device records (built by rt_scene_upload from RtFlatScene)
// ---------------------------------------------------------------------------------------------
struct MatRec { // 48 B
    uint32_t type, tex0, tex1, pad0;
    float cr, cg, cb, p0;
    float p1, p2, p3, pad1;
};
struct TexRec { // 48 B
    uint32_t type, aux;
    float scale;
    uint32_t pad0;
    float c0r, c0g, c0b, pad1;
    float c1r, c1g, c1b, pad2;
};
struct ImgRec { // 16 B
    uint32_t w, h;
    uint32_t offset_lo, offset_hi; // offset in texels into the float4 texel pool
};
struct DevScene {
    // instance wrappers (hitable.rs:404-520): xf_param = (offset xyz | sin, cos, angle), xf_meta = (type, parent)
    uint32_t n_xforms;
    const float4* xf_param;
    const uint2* xf_meta;
    const uint32_t* prim_xform; // [n_prims] innermost wrapper of the primitive or RT_NO_XFORM_DEV
    // homogeneous media (hitable.rs:523-588): medium m is world entry n_prims + m; its boundary is the list of
    // primitives med_prims[med_range[m].x .. +.y), which are not hit directly (prim_medium != RT_NO_XFORM_DEV)
    uint32_t n_media;
    const float* med_neg_inv_density;
    const uint2* med_range;
    const uint32_t* med_prims;
    uint32_t n_med_prims;      // total length of med_prims
    uint32_t n_xf_listed;      // entries behind xf_meta[n_xforms): the chains of more than RT_MAX_CHAIN wrappers, outermost first
    // world entries (BVH leaves): bounding sphere (xyz, r) and leaf id, for the primary-ray candidate lists
    uint32_t n_entries;
    const float4* ent_bs;
    const uint32_t* ent_leaf;
    float bvh_exact_eps;       // slab slack above which a ray takes the cancellation-free test (2^-10 of the scene extent)
    const uint2* med_xform;    // [n_media] .x: wrapper chain shared by all boundary primitives, or RT_MED_XF_MIXED; .y: the innermost
                               // wrapper AROUND the medium (RtFlatScene::med_xform) or RT_NO_XFORM_DEV
    const uint32_t* prim_medium; // [n_prims] owning medium or 0xFFFFFFFF
    uint32_t n_rects;   // axis-aligned rectangles; primitive index = n_spheres + rect index
    uint32_t n_prims;   // n_spheres + n_rects
    const float4* rect_geo; // 2 per rect: (k, u0, u1, v0), (v1, axis bits, 0, 0); (u, v) = uv axes of the rect
    const float4* prim_geo; // sph_geo[0..n_spheres) followed by rect_geo (one allocation; the LDS image of k_intersect)
    uint32_t n_spheres;
    uint32_t n_materials;
    uint32_t n_textures;
    uint32_t n_perlin;
    uint32_t n_images;
    uint32_t sky_type;
    uint32_t sky_image;
    uint32_t key_miss; // sort key (sph_class) of "no hit"
    uint32_t pad;
    const float4* sph_geo;   // (cx, cy, cz, r)
    const uint32_t* sph_mat;
    const MatRec* mats;
    const TexRec* texs;
    const float4* perlin_vec;   // [n_perlin*256] xyz_
    const unsigned short* perlin_perm2; // [n_perlin*3*256] pairs perm[i] | perm[(i+1) & 255] << 8 (PerlinTables)
    const ImgRec* imgs;
    const float4* texels;       // rgb_ (used when texels8 is NULL)
    const uint32_t* texels8;    // r | g << 8 | b << 16: every texel of every image is k/255 exactly (any decoded 8-bit image,
                                // texture.rs:176-177 to_rgb32f), stored as k; else NULL
    // LDS-resident 4-wide BVH over the spheres, built at upload (rt_bvh.h HostBvh4): per node one
    // float4 per box plane over the 4 children (min_x, min_y, min_z, max_x, max_y, max_z) + child ids:
    // id >= 0: inner node, id < 0: sphere ~id, INT_MIN: empty slot
    uint32_t n_bvh4_nodes;
    uint32_t bvh4_depth;
    const float4* bvh4_p[6];
    const int4* bvh4_id;
    // sort key of every world entry: the rank of its shading class (1 + material_type*4 + texture_type of tex0; 0 is
    // "miss") among the classes present in the scene, cheap classes first.  k_shade orders the hit records of a block
    // by this key so that a wave runs one material branch.
    const uint8_t* sph_class;
    // Per-primitive shading record, 5 x float4 (one dependent fetch after the hit index instead of
    // primitive -> material -> texture):  [0] sphere: cx cy cz r / rect: k u0 u1 v0
    // [1] type, tex0 type, tex0 aux, tex1 (u32 bits)   [2] colour (ConstantTex / Checker odd / Metal albedo) rgb, p0
    // [3] p1, p2, tex0 scale, tex0 (u32 bits)   [4] rect: v1, axis bits
    const float4* sph_rec;
};

// ---------------------------------------------------------------------------------------------
// texture.rs
// ---------------------------------------------------------------------------------------------
// Perlin table sets: `pvec`/`pperm` point either at the HBM copies in DevScene or at the
// workgroup's LDS copies (k_shade stages them when they fit, see rt_kernels.h).
// The permutation tables are stored as PAIRS, perm2[i] = perm[i] | perm[(i + 1) & 255] << 8: a lattice cell needs
// the entries of i and i + 1 on every axis (texture.rs:103-108), one 16-bit read instead of two byte reads.
struct PerlinTables {
    const float4* vec;   // [n_sets*256] xyz_
    const unsigned short* perm; // [n_sets*768] pairs of perm_x, perm_y, perm_z
};
// Lattice index of a far coordinate.  The reference takes `p.x.floor() as isize` and `(i + di).rem_euclid(256)` (texture.rs:126-138):
// a saturating conversion to 64 bits.  sat_i32 (v_cvt_i32_f32) followed by `& 255` is that for |fx| < 2^31 (two's complement), for
// fx <= -2^31 (saturates to 0x80000000: index 0, as every float there is a multiple of 256 and isize::MIN is one too), for NaN (0)
// and for fx >= 2^63 (0x7FFFFFFF: index 255 and, through the pair table, 0 for i + 1 — isize::MAX and the wrapped isize::MIN of a
// release build; a debug build of the reference panics there).  It is NOT that for 2^31 <= fx < 2^63, multiples of 256 every one:
// index 0.  perlin_turb tells its octaves whether any coordinate can get there (`far`), so the near path pays nothing.
__device__ __forceinline__ int perlin_far_index(float fx, int i) {
    return (fx >= 2147483648.0f && fx < 9223372036854775808.0f) ? 0 : i;
}
__device__ inline float perlin_noise(const PerlinTables& pt, uint32_t set, V3 p, bool far) { // texture.rs:125-146, 93-112
    const float4* rv = pt.vec + (size_t)set * 256;
    const unsigned short* px = pt.perm + (size_t)set * 768;
    const unsigned short* py = px + 256;
    const unsigned short* pz = py + 256;
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    int i = sat_i32(fx), j = sat_i32(fy), k = sat_i32(fz); // `as isize`: saturating, NaN -> 0
    if (far) i = perlin_far_index(fx, i), j = perlin_far_index(fy, j), k = perlin_far_index(fz, k);
    V3 uvw = p - v3(fx, fy, fz);
    V3 uvw2 = uvw * uvw * (3.0f - 2.0f * uvw); // math.rs:133-135 smooth
    float u = uvw2.x, v = uvw2.y, w = uvw2.z;
    const uint32_t ex = px[i & 255], ey = py[j & 255], ez = pz[k & 255]; // rem_euclid(256) of i and i + 1
    uint32_t hx[2] = {ex & 255u, ex >> 8};
    uint32_t hy[2] = {ey & 255u, ey >> 8};
    uint32_t hz[2] = {ez & 255u, ez >> 8};
    float accum = 0.0f;
    // All eight gradient gathers of the cell are issued before the first one is used.  Written one by one the compiler
    // waits for every ds_read before it issues the next (one register triple reused eight times): eight LDS round trips
    // per octave instead of one.  24 more live registers; k_shade past depth 0 is 4 % faster.
    float4 gg[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) gg[c] = rv[hx[c >> 2] ^ hy[(c >> 1) & 1] ^ hz[c & 1]];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const float4 g = gg[a * 4 + b * 2 + d];
                V3 weight = uvw - v3((float)a, (float)b, (float)d);
                accum += dot(v3(g.x, g.y, g.z), weight) * (a == 1 ? u : 1.0f - u) * (b == 1 ? v : 1.0f - v) *
                         (d == 1 ? w : 1.0f - w);
            }
    return accum;
}
__device__ inline float perlin_turb(const PerlinTables& pt, uint32_t set, V3 p) { // texture.rs:115-124
    float accum = 0.0f;
    float w = 1.0f;
    // 2^25: the seventh octave looks up p * 2^6 (a NaN coordinate is index 0 on either path)
    const bool far = fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z)) >= 33554432.0f;
    for (int it = 0; it < 7; ++it) {
        accum += w * perlin_noise(pt, set, p, far);
        p = p * 2.0f;
        w *= 0.5f;
    }
    return fabsf(accum);
}
__device__ inline V3 image_value(const DevScene& sc, uint32_t img, V2 uv, uint32_t& n_fetch) { // texture.rs:183-193
    ImgRec ir = sc.imgs[img];
    float u = clampf(uv.x, 0.0f, 1.0f);
    float v = 1.0f - clampf(uv.y, 0.0f, 1.0f);
    uint32_t i = min(sat_u32(u * (float)ir.w), ir.w - 1u);
    uint32_t j = min(sat_u32(v * (float)ir.h), ir.h - 1u);
    uint64_t off = ((uint64_t)ir.offset_hi << 32) | ir.offset_lo;
    ++n_fetch;
    if (sc.texels8) {
        // 4 B per texel instead of 16: a 128 B line holds 32 texels of the lookup's neighbourhood instead of 8, and the
        // pools of config 4 are 7 MB instead of 28.  (float)k / 255.0f is the correctly rounded quotient, the very value
        // the host's to_rgb32f produced.
        const uint32_t t8 = sc.texels8[off + (uint64_t)j * ir.w + i];
        return v3((float)(t8 & 255u) / 255.0f, (float)((t8 >> 8) & 255u) / 255.0f, (float)((t8 >> 16) & 255u) / 255.0f);
    }
    float4 t = sc.texels[off + (uint64_t)j * ir.w + i];
    return v3(t.x, t.y, t.z);
}
// `on` = outward unit normal of the hit sphere, from which rec.uv is derived lazily (hitable.rs:98)
__device__ inline V3 texture_value(const DevScene& sc, const PerlinTables& pt, uint32_t tex, V3 on, bool is_rect, V2 rect_uv,
                                   V3 p, uint32_t& n_fetch) {
    TexRec tr = sc.texs[tex];
    switch (tr.type) {
    case 0: // ConstantTex texture.rs:19-23
        return v3(tr.c0r, tr.c0g, tr.c0b);
    case 1: { // CheckerTex texture.rs:40-49
        float sines = sinf(p.x * 10.0f) * sinf(p.y * 10.0f) * sinf(p.z * 10.0f);
        return sines < 0.0f ? v3(tr.c0r, tr.c0g, tr.c0b) : v3(tr.c1r, tr.c1g, tr.c1b);
    }
    case 2: { // PerlinTex texture.rs:164-168
        float s = sinf(10.0f * perlin_turb(pt, tr.aux, p) + tr.scale * p.z);
        return (s + 1.0f) * 0.5f * splat(1.0f);
    }
    default: // ImageTex
        return image_value(sc, tr.aux, is_rect ? rect_uv : sphere_get_uv(on), n_fetch);
    }
}

// Texture::value for the material's first texture, described inline by the sphere record
// (type, aux, scale, colour 0); only CheckerTex needs its second colour from the TexRec table.
// `is_rect`: rec.uv is the rectangle's (p - min)/(max - min) instead of Sphere::get_uv(on).
__device__ inline V3 texture_value_inline(const DevScene& sc, const PerlinTables& pt, uint32_t ttype, uint32_t taux,
                                          float scale, V3 c0, uint32_t tex, V3 on, bool is_rect, V2 rect_uv, V3 p,
                                          uint32_t& n_fetch) {
    switch (ttype) {
    case 0: // ConstantTex texture.rs:19-23
        return c0;
    case 1: { // CheckerTex texture.rs:40-49
        float sines = sinf(p.x * 10.0f) * sinf(p.y * 10.0f) * sinf(p.z * 10.0f);
        if (sines < 0.0f) return c0;
        TexRec tr = sc.texs[tex];
        return v3(tr.c1r, tr.c1g, tr.c1b);
    }
    case 2: { // PerlinTex texture.rs:164-168
        RT_LANE_STAT(16, true);
        float s = sinf(10.0f * perlin_turb(pt, taux, p) + scale * p.z);
        return (s + 1.0f) * 0.5f * splat(1.0f);
    }
    default: // ImageTex
        return image_value(sc, taux, is_rect ? rect_uv : sphere_get_uv(on), n_fetch);
    }
}

// demo_scene.rs:22-35
__device__ inline V3 sky_value(const DevScene& sc, V3 d, uint32_t& n_fetch) {
    if (sc.sky_type == 0) { // sky_color demo_scene.rs:28-31
        float t = d.y * 0.5f + 0.5f;
        return lerp3(splat(1.0f), v3(0.5f, 0.7f, 1.0f), t);
    }
    if (sc.sky_type == 2) { // tex_sky_color demo_scene.rs:22-26
        V2 uv = sphere_get_uv(d);
        V3 c = image_value(sc, sc.sky_image, V2{1.0f - uv.x, uv.y}, n_fetch);
        return c * c;
    }
    return splat(0.0f); // black_sky
}

// ---------------------------------------------------------------------------------------------
// pbr.rs helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gtr1(float n_dot_h, float a) { // pbr.rs:72-79
    if (a >= 1.0f) return RT_FRAC_1_PI;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * n_dot_h * n_dot_h;
    return (a2 - 1.0f) / (RT_PI * logf(a2) * t);
}
__device__ __forceinline__ float gtr2(float n_dot_h, float a) { // pbr.rs:81-85
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * n_dot_h * n_dot_h;
    return a2 / (RT_PI * t * t);
}
__device__ __forceinline__ float gtr2_aniso(V3 h, float ax, float ay) { // pbr.rs:87-89
    return 1.0f / (RT_PI * ax * ay * powi2(powi2(h.x / ax) + powi2(h.y / ay) + h.z * h.z));
}
__device__ __forceinline__ float smith_geo_ggx(float n_dot_v, float alpha) { // pbr.rs:92-96
    float a = alpha * alpha;
    float b = n_dot_v * n_dot_v;
    return 1.0f / (n_dot_v + sqrtf(a + b - a * b));
}
__device__ __forceinline__ float smith_geo_ggx_aniso(V3 v, float ax, float ay) { // pbr.rs:98-100
    return 1.0f / (v.z + sqrtf(powi2(v.x * ax) + powi2(v.y * ay) + v.z * v.z));
}
__device__ __forceinline__ float fresnel_dielectric(float n_dot_i, float n_dot_t, float eta) { // pbr.rs:107-113
    float rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
    float rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
    return (rs * rs + rp * rp) / 2.0f;
}
__device__ __forceinline__ float fresnel_dielectric_2(float n_dot_i, float eta) { // pbr.rs:120-129
    float n_dot_t_sq = 1.0f - (1.0f - n_dot_i * n_dot_i) / (eta * eta);
    if (n_dot_t_sq < 0.0f) return 1.0f;
    return fresnel_dielectric(fabsf(n_dot_i), sqrtf(n_dot_t_sq), eta);
}
__device__ __forceinline__ float smith_masking_gtr2_2(V3 v_world, V3 n, float roughness) { // pbr.rs:145-152
    float alpha = roughness * roughness;
    float a2 = alpha * alpha;
    float v2_z_ = dot(v_world, n);
    float v2_z = v2_z_ * v2_z_;
    float lambda = (-1.0f + sqrtf(1.0f + a2 * (1.0f - v2_z) / v2_z)) / 2.0f;
    return 1.0f / (1.0f + lambda);
}
// hitable.rs:37-41 HitRecord::world_to_local_with_rot
__device__ __forceinline__ V3 world_to_local_with_rot(V3 norm, V3 tang0, V3 v, float rot) {
    V3 tang = cosf(rot) * tang0 - sinf(rot) * cross(norm, tang0);
    V3 bitang = cross(norm, tang);
    return v3(dot(v, tang), dot(v, bitang), dot(v, norm));
}

// ---------------------------------------------------------------------------------------------
// Closest hit against one sphere (hitable.rs:75-91).  Returns the accepted root in `t_hit`.
// The caller keeps (t, index) only; the HitRecord fields (hitable.rs:93-99) are derived
// once for the final hit in shade().
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sphere_root(float4 g, V3 o, V3 d, float a, float t_min, float t_max, float& t_hit) {
    V3 oc = o - v3(g.x, g.y, g.z);
    float half_b = dot(oc, d);
    float c = length_squared(oc) - g.w * g.w;
    // A sphere behind an origin outside it cannot be hit, and the reference's arithmetic agrees to the bit (so the square root
    // and the two divisions are skipped without changing any result): with c > 0 and a > 0 the discriminant is
    // fl(fl(hb^2) - fl(a c)) <= fl(hb^2), so sqrtd <= fl(sqrt(fl(hb^2))) = hb (correctly rounded sqrt of a rounded square gives
    // the number back; an overflowing hb^2 makes both roots infinite), both numerators -hb -+ sqrtd are <= 0 and both roots fail
    // `root < t_min` for any t_min > 0.  That is every ray leaving the surface it was scattered from (offset_hit_point puts the
    // origin 256 ulps outside, math.rs:144-156) — the r = 1000 ground for most secondary rays — and about half of the spheres
    // listed around a ray's origin.  (ConstantMedium's boundary searches pass t_min = -inf and take the plain path.)
    if (t_min > 0.0f && half_b > 0.0f && c > 0.0f) return false;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return false;
    float sqrtd = sqrt_ns(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    t_hit = root;
    return true;
}

// The same with the divisions by a = |d|^2 through the ray's shared reciprocal (above); t_min > 0 only.
__device__ __forceinline__ bool sphere_root(float4 g, V3 o, V3 d, const SharedRcp& ra, float t_min, float t_max, float& t_hit) {
    V3 oc = o - v3(g.x, g.y, g.z);
    float half_b = dot(oc, d);
    float c = length_squared(oc) - g.w * g.w;
    if (half_b > 0.0f && c > 0.0f) return false;
    float discriminant = half_b * half_b - ra.a * c;
    if (discriminant < 0.0f) return false;
    float sqrtd = sqrt_ns(discriminant);
    float root = div_shared(-half_b - sqrtd, ra);
    if (root < t_min || t_max < root) {
        root = div_shared(-half_b + sqrtd, ra);
        if (root < t_min || t_max < root) return false;
    }
    t_hit = root;
    return true;
}

// hitable.rs:244-362 XYRect/XZRect/YZRect::hit up to the accepted t.  g0 = (k, u0, u1, v0), g1 = (v1, axis):
// axis 0: x = k, (u,v) = (y,z); axis 1: y = k, (u,v) = (x,z); axis 2: z = k, (u,v) = (x,y).
// A NaN t (ray inside the plane: 0/0) is rejected here; the reference's comparisons all fail on it and
// report a hit with t = NaN (documented deviation, measure zero).
__device__ __forceinline__ bool rect_root(float4 g0, float4 g1, V3 o, V3 d, float t_min, float t_max, float& t_hit) {
    const uint32_t axis = __float_as_uint(g1.y);
    const float oa = axis == 0u ? o.x : (axis == 1u ? o.y : o.z);
    const float da = axis == 0u ? d.x : (axis == 1u ? d.y : d.z);
    const float t = (g0.x - oa) / da;
    if (!(t == t) || t < t_min || t > t_max) return false;
    const V3 p = o + d * t; // Ray::at
    const float pu = axis == 0u ? p.y : p.x;
    const float pv = axis == 2u ? p.y : p.z;
    if (pu < g0.y || pu > g0.z || pv < g0.w || pv > g1.x) return false;
    t_hit = t;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Instance wrappers Translate / RotateY (hitable.rs:404-520) as a per-primitive chain
// ---------------------------------------------------------------------------------------------
// xf_meta[x] = (type, parent).  Chains of up to RT_MAX_CHAIN wrappers (every demo scene: 2) are walked through the parent links into an
// unrolled array.  The reference nests without limit (hitable.rs:404-520 hold an Arc<dyn Hitable>), so a deeper chain is listed once
// more, outermost wrapper first, behind the wrapper table itself — entries xf_meta[p0 + k] = (wrapper, chain length) with p0 = the
// bits of xf_param[x].w, 0 for a short chain (rt_scene_upload) — and walked by a loop: the same xform_ray / unwind steps in the same
// order, so the same bits (tests: chains of up to 18 wrappers against the oracle).  NEST: kernels are compiled twice, and only
// scenes that nest beyond what the registers hold (RtCtx::nest: a chain of more than RT_MAX_CHAIN, more media than the per-lane
// mask has bits, wrappers AROUND a medium) run the instantiation with the loops — the others run the code they always ran.
#define RT_NO_XFORM_DEV 0xFFFFFFFFu
#define RT_MAX_CHAIN 4
struct Chain {
    uint32_t id[RT_MAX_CHAIN]; // id[0] = innermost wrapper ... id[n-1] = outermost
    int n;
};
// (templates over the table holder: DevScene, or the GenTables view of rt_kernels.h whose tables live in LDS)
template <class Tables>
__device__ __forceinline__ Chain load_chain(const Tables& sc, uint32_t xf) {
    Chain c;
    c.n = 0;
#pragma unroll
    for (int k = 0; k < RT_MAX_CHAIN; ++k) {
        c.id[k] = xf;
        if (xf != RT_NO_XFORM_DEV) {
            ++c.n;
            xf = sc.xf_meta[xf].y;
        }
    }
    return c;
}
// The ray a wrapper hands to its child: Translate moves the origin (hitable.rs:411), RotateY rotates origin
// and direction (hitable.rs:483-492).
template <class Tables>
__device__ __forceinline__ void xform_ray(const Tables& sc, uint32_t x, V3& o, V3& d) {
    const float4 q = sc.xf_param[x];
    if (sc.xf_meta[x].x == 0u) {
        o = o - v3(q.x, q.y, q.z);
    } else {
        const float sin_theta = q.x, cos_theta = q.y;
        const float ox = cos_theta * o.x - sin_theta * o.z, oz = sin_theta * o.x + cos_theta * o.z;
        const float dx = cos_theta * d.x - sin_theta * d.z, dz = sin_theta * d.x + cos_theta * d.z;
        o.x = ox, o.z = oz, d.x = dx, d.z = dz;
    }
}
// world ray -> the ray the primitive itself is tested with (outermost wrapper first)
template <class Tables>
__device__ __forceinline__ void chain_to_object(const Tables& sc, const Chain& c, V3& o, V3& d) {
#pragma unroll
    for (int k = RT_MAX_CHAIN - 1; k >= 0; --k)
        if (k < c.n) xform_ray(sc, c.id[k], o, d);
}
// world ray -> object ray below the innermost wrapper `xf`, whatever the depth of its chain
template <bool NEST, class Tables>
__device__ __forceinline__ void ray_to_object(const Tables& sc, uint32_t xf, V3& o, V3& d) {
    const uint32_t p0 = NEST ? __float_as_uint(sc.xf_param[xf].w) : 0u;
    if (NEST && p0 != 0u) {
        const uint32_t depth = sc.xf_meta[p0].y;
        for (uint32_t k = 0; k < depth; ++k) xform_ray(sc, sc.xf_meta[p0 + k].x, o, d);
    } else {
        chain_to_object(sc, load_chain(sc, xf), o, d);
    }
}

// Result of one bounce for one ray.
struct Bounce {
    V3 radiance;    // emitted (hit) or sky (miss) term of this segment, untinted
    V3 attenuation; // material attenuation when alive
    V3 o, d;        // scattered ray when alive
    bool alive;
};

// main.rs:44-58 for one segment whose closest hit is already known.
// hit < 0: miss -> sky.  Otherwise rebuild the HitRecord (hitable.rs:93-99) and run
// emitted + scatter of the material (material.rs, pbr.rs).
// RECTS = false compiles the rectangle branches out (scenes without rectangles: 2 % faster).
// `after_record_loads()` is called once, right after the loads of the primitive record have been issued: k_shade
// requests the NEXT segment's rays there.  Vector-memory results return in issue order (one vmcnt counter), so a
// prefetch issued BEFORE the record loads would have to complete before the record can be used — its HBM latency would
// sit in front of the shading instead of under it.
struct NoPrefetch {
    __device__ __forceinline__ void operator()() const {}
};
template <bool RECTS, class AfterLoads = NoPrefetch, bool NEST = false>
__device__ inline Bounce shade(const DevScene& sc, const PerlinTables& pt, V3 ro, V3 rd, int hit, float t, Rng& rng,
                               uint32_t& n_fetch, AfterLoads after_record_loads = AfterLoads()) {
    Bounce out;
    out.radiance = splat(0.0f);
    out.attenuation = splat(1.0f);
    out.o = splat(0.0f);
    out.d = splat(0.0f);
    out.alive = false;
    // (the record of entry 0 for a miss: every lane of the wave issues the same loads before the prefetch)
    const float4* rec = sc.sph_rec + 5u * (uint32_t)(hit < 0 ? 0 : hit);
    const float4 g = rec[0], r1 = rec[1], r2 = rec[2];
    after_record_loads();
    if (hit < 0) {
        out.radiance = sky_value(sc, rd, n_fetch); // main.rs:58
        return out;
    }
    const bool is_medium = RECTS && (uint32_t)hit >= sc.n_prims;
    const bool is_rect = RECTS && !is_medium && (uint32_t)hit >= sc.n_spheres;
    // RECTS also stands for "general scene": the primitive or medium may sit below Translate / RotateY wrappers.  The
    // HitRecord is then built from the innermost (object-space) ray and fixed on the way out (hitable.rs:412-414,
    // 494-506); `ro`/`rd` stay the world ray that scatter() receives (main.rs:48).
    Chain chain;
    chain.n = 0;
    uint32_t deep_n = 0u, deep_p0 = 0u; // a chain of more than RT_MAX_CHAIN wrappers: its length and its outermost-first list
    V3 wo = ro, wd = rd; // the world ray
    if (RECTS && (NEST || !is_medium)) { // (a medium: the wrappers AROUND it, hitable.rs:409-416 with ptr = a ConstantMedium; those of its boundary are inside its hit())
        const uint32_t xf = NEST && is_medium ? sc.med_xform[(uint32_t)hit - sc.n_prims].y : sc.prim_xform[hit];
        deep_p0 = NEST && xf != RT_NO_XFORM_DEV ? __float_as_uint(sc.xf_param[xf].w) : 0u;
        if (NEST && deep_p0 != 0u) {
            deep_n = sc.xf_meta[deep_p0].y;
            for (uint32_t k = 0; k < deep_n; ++k) xform_ray(sc, sc.xf_meta[deep_p0 + k].x, ro, rd);
        } else {
            chain = load_chain(sc, xf);
            chain_to_object(sc, chain, ro, rd);
        }
    }
    V3 p = ro + rd * t;                                 // math.rs:64 Ray::at
    V3 on;
    V2 rect_uv = V2{0.0f, 0.0f};
    if (is_rect) { // hitable.rs:262-268 (and the XZ / YZ twins): uv = (p - min)/(max - min), normal = +axis
        const float4 g1 = rec[4];
        const uint32_t axis = __float_as_uint(g1.y);
        const float pu = axis == 0u ? p.y : p.x;
        const float pv = axis == 2u ? p.y : p.z;
        rect_uv = V2{(pu - g.y) / (g.z - g.y), (pv - g.w) / (g1.x - g.w)};
        on = v3(axis == 0u ? 1.0f : 0.0f, axis == 1u ? 1.0f : 0.0f, axis == 2u ? 1.0f : 0.0f);
    } else if (is_medium) {
        on = v3(1.0f, 0.0f, 0.0f);                      // hitable.rs:574 rec.norm = Vec3A::X
    } else {
        const SharedRcp rr = shared_rcp(g.w);           // hitable.rs:95 outward_normal = (p - center) / radius, one reciprocal for three quotients
        const V3 pc = p - v3(g.x, g.y, g.z);
        on = v3(div_shared(pc.x, rr), div_shared(pc.y, rr), div_shared(pc.z, rr));
    }
    bool front_face = is_medium ? true : dot(rd, on) < 0.0f; // hitable.rs:26 / hitable.rs:575
    V3 n = front_face ? on : -on;                       // hitable.rs:27-31
    if (RECTS && chain.n > 0) {
        // unwind the wrappers from the inside out
#pragma unroll
        for (int k = 0; k < RT_MAX_CHAIN; ++k) {
            if (k < chain.n) {
                const uint32_t x = chain.id[k];
                const float4 q = sc.xf_param[x];
                if (sc.xf_meta[x].x == 0u) { // Translate: only rec.p moves (hitable.rs:413)
                    p = p + v3(q.x, q.y, q.z);
                } else { // RotateY (hitable.rs:495-505)
                    const float sin_theta = q.x, cos_theta = q.y;
                    const float px = cos_theta * p.x + sin_theta * p.z, pz = -sin_theta * p.x + cos_theta * p.z;
                    const float nx = cos_theta * n.x + sin_theta * n.z, nz = -sin_theta * n.x + cos_theta * n.z;
                    p.x = px, p.z = pz, n.x = nx, n.z = nz;
                    // set_face_normal(&rot_r, n): rot_r.d = the world direction through the wrappers k..n-1
                    V3 dk = wd, ok = wo;
#pragma unroll
                    for (int j = RT_MAX_CHAIN - 1; j >= 0; --j)
                        if (j < chain.n && j >= k) xform_ray(sc, chain.id[j], ok, dk);
                    front_face = dot(dk, n) < 0.0f;
                    n = front_face ? n : -n;
                }
            }
        }
        ro = wo, rd = wd; // materials see the world ray
    }
    if (RECTS && NEST && deep_n) { // the same unwinding for a chain that is only listed (level k from the inside = list entry deep_n - 1 - k)
        for (uint32_t k = 0; k < deep_n; ++k) {
            const uint32_t x = sc.xf_meta[deep_p0 + deep_n - 1u - k].x;
            const float4 q = sc.xf_param[x];
            if (sc.xf_meta[x].x == 0u) {
                p = p + v3(q.x, q.y, q.z);
            } else {
                const float sin_theta = q.x, cos_theta = q.y;
                const float px = cos_theta * p.x + sin_theta * p.z, pz = -sin_theta * p.x + cos_theta * p.z;
                const float nx = cos_theta * n.x + sin_theta * n.z, nz = -sin_theta * n.x + cos_theta * n.z;
                p.x = px, p.z = pz, n.x = nx, n.z = nz;
                V3 dk = wd, ok = wo; // rot_r of this wrapper: the world ray through the wrappers from the outermost one down to it
                for (uint32_t j = 0; j < deep_n - k; ++j) xform_ray(sc, sc.xf_meta[deep_p0 + j].x, ok, dk);
                front_face = dot(dk, n) < 0.0f;
                n = front_face ? n : -n;
            }
        }
        ro = wo, rd = wd;
    }
    // the material fields are fetched where a branch needs them (keeps the live set small)
    struct {
        uint32_t type, tex1;
        const float4* rec;
        float4 r2;
        __device__ __forceinline__ float p0() const { return r2.w; }
        __device__ __forceinline__ float p1() const { return rec[3].x; }
        __device__ __forceinline__ float p2() const { return rec[3].y; }
        __device__ __forceinline__ V3 color() const { return v3(r2.x, r2.y, r2.z); }
    } m{__float_as_uint(r1.x), __float_as_uint(r1.w), rec, r2};
    const uint32_t t0type = __float_as_uint(r1.y), t0aux = __float_as_uint(r1.z);
    // Every texture-bearing material evaluates its first texture exactly once at (uv(on), p), and
    // Texture::value draws no random numbers, so it is evaluated here, at ONE call site (the 7-octave
    // Perlin body is instantiated once and stays inlined; per-branch call sites made the compiler emit
    // real function calls with a scratch stack).
    const uint32_t TEX0_USERS = (1u << 0) | (1u << 1) | (1u << 2) | (1u << 5) | (1u << 6) | (1u << 7) | (1u << 8) | (1u << 9) |
                                (1u << 10) | (1u << 11);
    V3 tex0_value = splat(0.0f);
    if ((TEX0_USERS >> m.type) & 1u)
        tex0_value = texture_value_inline(sc, pt, t0type, t0aux, rec[3].z, m.color(), __float_as_uint(rec[3].w), on, is_rect,
                                          rect_uv, p, n_fetch);
#define RT_TEX0() tex0_value
    switch (m.type) {
    case 0: // Emission material.rs:21-28
        RT_LANE_STAT(20, true);
        out.radiance = RT_TEX0();
        return out;
    case 1: { // Diffuse material.rs:35-46
        RT_LANE_STAT(22, true);
        V3 sd = n + normalize(random_in_unit_sphere(rng));
        const float eps = 1.1920929e-7f;
        if (fabsf(sd.x) < eps && fabsf(sd.y) < eps && fabsf(sd.z) < eps) sd = n; // math.rs:8-11
        out.o = offset_hit_point(p, n);
        out.d = normalize(sd);
        out.attenuation = RT_TEX0();
        out.alive = true;
        return out;
    }
    case 3: { // Metal material.rs:66-73
        RT_LANE_STAT(26, true);
        V3 reflected = reflect(rd, n) + m.p0() * random_in_unit_sphere(rng);
        out.o = p;
        out.d = normalize(reflected);
        out.attenuation = m.color();
        out.alive = dot(reflected, n) > 0.0f;
        return out;
    }
    case 4: { // Dielectric material.rs:79-97
        RT_LANE_STAT(28, true);
        float ref_idx = front_face ? 1.0f / m.p0() : m.p0();
        float cos_theta = -fminf(dot(rd, n), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        bool cannot_refract = sin_theta * ref_idx > 1.0f;
        float rnd_num = rng.next();
        V3 dir = (cannot_refract || reflectance(cos_theta, ref_idx) > rnd_num) ? reflect(rd, n) : refract(rd, n, ref_idx);
        out.o = p;
        out.d = normalize(dir);
        out.alive = true;
        return out;
    }
    case 5: { // Isotropic material.rs:103-113
        out.o = p;
        out.d = normalize(random_in_unit_sphere(rng));
        out.attenuation = RT_TEX0();
        out.alive = true;
        return out;
    }
    default:
        break;
    }
    // Lambert (material.rs:52-59) and the seven pbr.rs materials share: offset origin + a
    // uniform hemisphere direction (pbr.rs:19-20 and equivalents).
    RT_LANE_STAT(24, true);
    V3 po = offset_hit_point(p, n);
    V3 dir_o = random_on_hemisphere(rng, n);
    out.o = po;
    out.d = dir_o;
    out.alive = true;
    float n_dot_i = dot(n, -rd);
    float n_dot_o = dot(n, dir_o);
    switch (m.type) {
    case 2: // Lambert material.rs:56
        out.attenuation = RT_TEX0() * 2.0f * dot(n, dir_o);
        break;
    case 6: { // OrenNayar pbr.rs:21-39
        float cos_i = fabsf(dot(n, rd));
        float cos_o = n_dot_o;
        float sin_i = sqrtf(1.0f - cos_i * cos_i);
        float sin_o = sqrtf(1.0f - cos_o * cos_o);
        float max_cos = fmaxf(cos_i * cos_o + sin_i * sin_o, 0.0f);
        float r2 = m.p0() * m.p0();
        float a = 1.0f - 0.5f * r2 / (r2 + 0.33f);
        float b = 0.45f * r2 / (r2 + 0.09f);
        float sin_alpha, tan_beta;
        if (cos_i > cos_o) {
            sin_alpha = sin_o;
            tan_beta = sin_i / cos_i;
        } else {
            sin_alpha = sin_i;
            tan_beta = sin_o / cos_o;
        }
        float w = a + b * max_cos * sin_alpha * tan_beta;
        out.attenuation = RT_TEX0() * w * 2.0f * cos_o;
        break;
    }
    case 7: { // BurleyDiffuse pbr.rs:54-66
        V3 h = normalize(dir_o - rd);
        float h_dot_o = dot(h, dir_o);
        float fl = schlick_fresnel(n_dot_o);
        float fv = schlick_fresnel(n_dot_i);
        float fd90 = 0.5f + 2.0f * h_dot_o * h_dot_o * m.p0();
        float fd = lerpf(1.0f, fd90, fl) * lerpf(1.0f, fd90, fv);
        out.attenuation = RT_TEX0() * fd * 2.0f * n_dot_o;
        break;
    }
    case 8: { // RoughPlastic pbr.rs:164-186
        V3 h = normalize(dir_o - rd);
        float h_dot_i = dot(h, -rd);
        float h_dot_o = dot(h, dir_o);
        float n_dot_h = dot(n, h);
        V3 kd = texture_value(sc, pt, m.tex1, on, is_rect, rect_uv, p, n_fetch);
        V3 ks = RT_TEX0();
        float roughness = clampf(m.p0(), 0.01f, 1.0f);
        float eta = m.p1();
        float f_o = fresnel_dielectric_2(h_dot_o, eta);
        float dd = gtr2(n_dot_h, roughness);
        float gg = smith_masking_gtr2_2(-rd, n, roughness) * smith_masking_gtr2_2(dir_o, n, roughness);
        V3 spec_contrib = ks * (gg * f_o * dd) / (4.0f * n_dot_i * n_dot_o);
        float f_i = fresnel_dielectric_2(h_dot_i, eta);
        V3 diff_contrib = kd * (1.0f - f_o) * (1.0f - f_i) * RT_FRAC_1_PI;
        out.attenuation = (spec_contrib + diff_contrib) * n_dot_o * 2.0f * RT_PI;
        break;
    }
    case 9: { // DisneyDiffuse pbr.rs:202-220
        V3 h = normalize(dir_o - rd);
        float h_dot_o = dot(h, dir_o);
        float fo = schlick_fresnel(n_dot_o);
        float fi = schlick_fresnel(n_dot_i);
        float fd90 = 0.5f + 2.0f * h_dot_o * h_dot_o * m.p0();
        float fd = lerpf(1.0f, fd90, fo) * lerpf(1.0f, fd90, fi);
        float fss90 = m.p0() * h_dot_o * h_dot_o;
        float fss_wi = lerpf(1.0f, fss90, fi);
        float fss_wo = lerpf(1.0f, fss90, fo);
        float fss = 1.25f * (fss_wi * fss_wo * (1.0f / (n_dot_i + n_dot_o) - 0.5f) + 0.5f);
        out.attenuation = RT_TEX0() * lerpf(fd, fss, m.p1()) * 2.0f * n_dot_o;
        break;
    }
    case 10: { // DisneyMetal pbr.rs:236-275
        V3 h = normalize(dir_o - rd);
        float h_dot_o = dot(h, dir_o);
        float n_dot_h = dot(n, h);
        V3 albedo = RT_TEX0();
        V3 fm = lerp3(albedo, splat(1.0f), schlick_fresnel(h_dot_o));
        const float alpha_min = 0.0001f;
        float dm, gm;
        if (m.p1() > -10.0f) {
            float aspect = sqrtf(1.0f - 0.9f * m.p1());
            float ax = fmaxf(m.p0() * m.p0() / aspect, alpha_min);
            float ay = fmaxf(m.p0() * m.p0() * aspect, alpha_min);
            float rot = m.p2() * 2.0f * RT_PI;
            V3 tang = normalize_any(cross(v3(0.0f, 1.0f, 0.0f), on)); // hitable.rs:96
            V3 h_local = world_to_local_with_rot(n, tang, h, rot);
            dm = gtr2_aniso(h_local, ax, ay);
            V3 i_local = world_to_local_with_rot(n, tang, -rd, rot);
            V3 o_local = world_to_local_with_rot(n, tang, dir_o, rot);
            gm = smith_geo_ggx_aniso(i_local, ax, ay) * smith_geo_ggx_aniso(o_local, ax, ay);
        } else {
            float r2 = fmaxf(m.p0() * m.p0(), alpha_min);
            dm = gtr2(n_dot_h, r2);
            gm = smith_geo_ggx(n_dot_i, r2) * smith_geo_ggx(n_dot_o, r2);
        }
        V3 metal_w = fm * dm * gm;
        out.attenuation = metal_w * n_dot_o * 2.0f * RT_PI;
        break;
    }
    case 11: { // DisneySheen pbr.rs:290-305
        V3 h = normalize(dir_o - rd);
        float h_dot_o = dot(h, dir_o);
        V3 albedo = RT_TEX0();
        float luminance = dot(v3(0.3f, 0.6f, 0.1f), albedo);
        V3 c_tint = luminance > 0.0f ? albedo / luminance : splat(1.0f);
        V3 c_sheen = lerp3(splat(1.0f), c_tint, m.p0());
        V3 f_sheen = c_sheen * schlick_fresnel(h_dot_o);
        out.attenuation = f_sheen * n_dot_o * 2.0f * RT_PI;
        break;
    }
    case 12: { // DisneyClearcoat pbr.rs:318-332
        V3 h = normalize(dir_o - rd);
        float h_dot_o = dot(h, dir_o);
        float n_dot_h = dot(n, h);
        float fc = lerpf(0.4f, 1.0f, schlick_fresnel(h_dot_o));
        float dc = gtr1(n_dot_h, lerpf(0.1f, 0.001f, m.p0()));
        float gc = smith_geo_ggx(n_dot_i, 0.25f) * smith_geo_ggx(n_dot_o, 0.25f);
        float cc = 0.25f * fc * dc * gc;
        out.attenuation = splat(cc) * n_dot_o * 2.0f * RT_PI;
        break;
    }
    default: // unknown tag: absorb
        out.alive = false;
        break;
    }
#undef RT_TEX0
    return out;
}

} // namespace rt
