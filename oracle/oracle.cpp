/*
 * oracle.cpp — CPU restatement of the reference's per-pixel integration loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ray_tracing_in_one_weekend_amd/,
 * include/) may include, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (zhouhang95/ray_tracing_in_one_weekend, Rust) ships no
 * tests, golden vectors or fixtures, cannot be compiled here (no rustc/cargo, crate sources
 * absent), and leans on third-party crates that are not in /root/reference:
 *   rand 0.8.5 / rand_core 0.6.4 (SmallRng = xoshiro256++, seed_from_u64 = rand_core's
 *   PCG32 seed expansion, Standard f32 = (u32 >> 8) * 2^-24, gen_range widening-multiply
 *   rejection, Fisher-Yates shuffle), glam 0.21.3 (Vec3A op order, normalize = v * (1/len)),
 *   Rust std f32 libm.  Those published algorithms are restated below from knowledge of the
 *   crates and are pinned only by (a) the public xoshiro256++ reference vector, (b) analytic
 *   known-answer tests derived from the reference formulas, (c) analytic image
 *   tests and (d) self-consistency between the two RNG modes.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src).  Arithmetic is IEEE fp32 in the reference's operation order; build
 * with -ffp-contract=off (Rust never fuses a*b+c).
 *
 * Two RNG modes:
 *   stream  : reference order — one xoshiro256++ stream per image column seeded 95+i
 *             (main.rs:81-83), jitter for all samples of a pixel drawn before tracing
 *             (main.rs:88-93).  This is the timed CPU baseline.
 *   counter : the counter-based generator that the GPU uses, keyed by (pixel, sample) with a
 *             counter block per depth (DESIGN.md "RNG").  This is the parity oracle.
 */
#include "../include/rtow_mi355x_debug.h" /* RtBounceIO: the oracle mirrors the single-bounce test hook */

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

/* ------------------------------------------------------------------------------------------
 * glam 0.21.3 Vec3A restatement (x86-64 SSE2 path): component ops are lane-wise IEEE fp32;
 * dot = (x*x' + y*y') + z*z'; normalize = v * (1 / sqrt(dot)); lerp = a + (b - a) * s.
 * ---------------------------------------------------------------------------------------- */
struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 splat(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 operator+(V3 a, float s) { return V3{a.x + s, a.y + s, a.z + s}; }
inline V3 operator-(V3 a, float s) { return V3{a.x - s, a.y - s, a.z - s}; }
inline V3 operator-(float s, V3 a) { return V3{s - a.x, s - a.y, s - a.z}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length_squared(V3 a) { return dot(a, a); }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) {
    float inv = 1.0f / std::sqrt(dot(a, a));
    return a * inv;
}
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 lerp3(V3 a, V3 b, float s) { return a + ((b - a) * s); }
inline float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

const float PI_F = 3.14159265358979323846f;
const float FRAC_1_PI_F = 0.318309886183790671537767526745028724f;

/* Rust `as` casts saturate and map NaN to 0. */
inline uint32_t sat_u32(float f) {
    if (!(f == f)) return 0u;
    if (f <= 0.0f) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
inline int32_t sat_i32(float f) {
    if (!(f == f)) return 0;
    if (f <= -2147483648.0f) return INT32_MIN;
    if (f >= 2147483648.0f) return INT32_MAX;
    return (int32_t)f;
}
inline int64_t sat_i64(float f) {
    if (!(f == f)) return 0;
    if (f <= -9223372036854775808.0f) return INT64_MIN;
    if (f >= 9223372036854775808.0f) return INT64_MAX;
    return (int64_t)f;
}
inline uint8_t sat_u8(float f) {
    if (!(f == f)) return 0;
    if (f <= 0.0f) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}
/* Rust f32::clamp: NaN stays NaN. */
inline float clampf(float x, float lo, float hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
inline uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float u2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
/* powi with a constant exponent as LLVM expands it: powi(x,5) = x * ((x*x)*(x*x)). */
inline float powi2(float x) { return x * x; }
inline float powi5(float x) {
    float x2 = x * x;
    float x4 = x2 * x2;
    return x * x4;
}

/* ------------------------------------------------------------------------------------------
 * rand 0.8.5 restatement.  SmallRng on 64-bit = Xoshiro256PlusPlus.
 * ---------------------------------------------------------------------------------------- */
struct Xoshiro256pp {
    uint64_t s[4];
    static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next_u64() {
        uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    /* xoshiro256plusplus.rs: next_u32 takes the upper half */
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
    /* rand::distributions::Standard for f32: 24 random bits, [0,1) */
    float next_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
};

/* rand_core 0.6.4 SeedableRng::seed_from_u64 default (PCG32 expansion into a 32-byte seed);
 * SmallRng 0.8.5 does not override it (the xoshiro SplitMix64 override is not forwarded). */
Xoshiro256pp smallrng_seed_from_u64_pcg(uint64_t state) {
    const uint64_t MUL = 6364136223846793005ull;
    const uint64_t INC = 11634580027462260723ull;
    uint8_t seed[32];
    for (int c = 0; c < 8; ++c) {
        state = state * MUL + INC;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        seed[4 * c + 0] = (uint8_t)(x);
        seed[4 * c + 1] = (uint8_t)(x >> 8);
        seed[4 * c + 2] = (uint8_t)(x >> 16);
        seed[4 * c + 3] = (uint8_t)(x >> 24);
    }
    Xoshiro256pp r;
    bool all_zero = true;
    for (int i = 0; i < 4; ++i) {
        uint64_t v = 0;
        for (int b = 7; b >= 0; --b) v = (v << 8) | seed[8 * i + b];
        r.s[i] = v;
        if (v) all_zero = false;
    }
    if (all_zero) { /* from_seed maps the zero seed to seed_from_u64(0) of xoshiro (SplitMix64) */
        uint64_t z = 0;
        for (int i = 0; i < 4; ++i) {
            z += 0x9e3779b97f4a7c15ull;
            uint64_t w = z;
            w = (w ^ (w >> 30)) * 0xbf58476d1ce4e5b9ull;
            w = (w ^ (w >> 27)) * 0x94d049bb133111ebull;
            r.s[i] = w ^ (w >> 31);
        }
    }
    return r;
}
/* The alternative candidate (xoshiro's own SplitMix64 seeding), kept selectable because the
 * crate source is not available offline. */
Xoshiro256pp smallrng_seed_from_u64_splitmix(uint64_t state) {
    Xoshiro256pp r;
    for (int i = 0; i < 4; ++i) {
        state += 0x9e3779b97f4a7c15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        r.s[i] = z ^ (z >> 31);
    }
    return r;
}
int g_seed_variant = 0; /* 0 = PCG32 expansion (rand 0.8.5 guess), 1 = SplitMix64 */
Xoshiro256pp smallrng_seed_from_u64(uint64_t seed) {
    return g_seed_variant == 0 ? smallrng_seed_from_u64_pcg(seed) : smallrng_seed_from_u64_splitmix(seed);
}

/* rand 0.8.5 UniformInt::sample_single_inclusive, u32 flavour (used by shuffle's gen_index) */
uint32_t gen_range_u32(Xoshiro256pp& rng, uint32_t low, uint32_t high_excl) {
    uint32_t range = high_excl - low;
    int lz = __builtin_clz(range);
    uint32_t zone = (range << lz) - 1u;
    for (;;) {
        uint32_t v = rng.next_u32();
        uint64_t m = (uint64_t)v * (uint64_t)range;
        uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo <= zone) return low + hi;
    }
}
/* usize flavour (hitable.rs:183 gen_range(0..3) with axis: usize) */
uint64_t gen_range_u64(Xoshiro256pp& rng, uint64_t low, uint64_t high_excl) {
    uint64_t range = high_excl - low;
    int lz = __builtin_clzll(range);
    uint64_t zone = (range << lz) - 1ull;
    for (;;) {
        uint64_t v = rng.next_u64();
        unsigned __int128 m = (unsigned __int128)v * (unsigned __int128)range;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo <= zone) return low + hi;
    }
}
/* rand 0.8.5 SliceRandom::shuffle: for i in (1..len).rev() swap(i, gen_index(i+1)) */
template <class T>
void shuffle(Xoshiro256pp& rng, std::vector<T>& v) {
    for (size_t i = v.size() - 1; i >= 1; --i) {
        size_t j = gen_range_u32(rng, 0u, (uint32_t)(i + 1));
        std::swap(v[i], v[j]);
    }
}

/* ------------------------------------------------------------------------------------------
 * Counter-based generator shared (by specification, not by code) with the GPU kernels.
 * DESIGN.md "RNG": draw(k0,k1,ctr) = mix32((k0 ^ ctr*0x9E3779B9) + k1), mix32 = the lowbias32 finaliser.
 * ---------------------------------------------------------------------------------------- */
inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
inline void ctr_path_key(uint64_t seed, uint32_t pix, uint32_t samp, uint32_t& k0, uint32_t& k1) {
    uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
    uint32_t a = fmix32(pix ^ s_lo);
    k0 = fmix32(a + samp * 0x9E3779B9u + s_hi);
    k1 = fmix32((a ^ 0xA511E9B3u) + samp * 0xC2B2AE3Du);
}
inline uint32_t ctr_draw(uint32_t k0, uint32_t k1, uint32_t ctr) {
    return mix32((k0 ^ (ctr * 0x9E3779B9u)) + k1);
}

/* lib.rs:7-9 thread-local RNG, abstracted over the two modes. */
struct Rng {
    bool counter;
    Xoshiro256pp xo;
    uint32_t k0, k1, ctr;
    uint32_t base = 0; /* first counter of the current depth block */
    float next_f32() {
        if (counter) {
            uint32_t r = ctr_draw(k0, k1, ctr++);
            return (float)(r >> 8) * (1.0f / 16777216.0f);
        }
        return xo.next_f32();
    }
    void set_depth(int depth) {
        if (counter) ctr = base = (uint32_t)(depth + 1) * 256u;
    }
};
/* The one f32 a ConstantMedium draws inside hit() (hitable.rs:564).  Stream mode: the next value of the
 * stream, in traversal order like the reference.  Counter mode: slot 224 + medium index of the depth block,
 * so the value does not depend on the order in which media are visited. */
inline float rng_medium_draw(Rng& rng, uint32_t m) {
    if (!rng.counter) return rng.xo.next_f32();
    /* (media beyond the 32nd: a block of 65 536 counters per depth above 2^30, where no other draw of a path lies; base = (depth + 1) * 256) */
    const uint32_t ctr = m < 32u ? rng.base + 224u + m : 0x40000000u + (rng.base << 8) + m;
    return (float)(ctr_draw(rng.k0, rng.k1, ctr) >> 8) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------------------------------
 * math.rs
 * ---------------------------------------------------------------------------------------- */
struct Ray {
    V3 o, d;
}; /* math.rs:55-60 (the debug-only `s` screen coordinate is not carried) */
inline V3 ray_at(const Ray& r, float t) { return r.o + r.d * t; } /* math.rs:62-66 */

inline bool vec3a_near_zero(V3 v) { /* math.rs:8-11 */
    const float s = 1.1920929e-7f;  /* f32::EPSILON */
    return (std::fabs(v.x) < s) && (std::fabs(v.y) < s) && (std::fabs(v.z) < s);
}
inline bool vec3a_near_one(V3 v) { /* math.rs:13-15 */
    return std::fabs(length(v) - 1.0f) < 1e-6f;
}
inline V3 vec3a_random(Rng& rng) { /* math.rs:17-24, x then y then z */
    float x = rng.next_f32();
    float y = rng.next_f32();
    float z = rng.next_f32();
    return v3(x, y, z);
}
inline V3 vec3a_random_range(Rng& rng, float mn, float mx) { /* math.rs:26-28 */
    return vec3a_random(rng) * (mx - mn) + mn;
}
inline V3 random_in_unit_sphere(Rng& rng) { /* math.rs:30-37 */
    for (;;) {
        V3 v = vec3a_random_range(rng, -1.0f, 1.0f);
        if (length_squared(v) < 1.0f) return v;
    }
}
inline V3 random_on_unit_sphere(Rng& rng) { return normalize(random_in_unit_sphere(rng)); } /* :38-40 */
inline V3 random_in_hemisphere(Rng& rng, V3 n) { /* math.rs:43-50 */
    V3 v = random_in_unit_sphere(rng);
    if (dot(v, n) > 0.0f) return v;
    return -v;
}
inline V3 random_on_hemisphere(Rng& rng, V3 n) { return normalize(random_in_hemisphere(rng, n)); } /* :51-53 */

inline V3 reflect(V3 v, V3 n) { return v - 2.0f * dot(v, n) * n; } /* math.rs:68-70 */
inline V3 refract(V3 uv, V3 n, float etai_over_etat) {             /* math.rs:72-77 */
    float cos_theta = -std::fmin(dot(uv, n), 1.0f); /* unary minus binds after .min() */
    V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V3 r_out_parallel = -std::sqrt(std::fabs(1.0f - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}
inline float schlick_fresnel(float u) { return powi5(1.0f - u); } /* math.rs:79-81 */
inline float reflectance(float cosine, float ref_idx) {          /* math.rs:84-88 */
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * schlick_fresnel(cosine);
}
inline V3 smooth(V3 v) { return v * v * (3.0f - 2.0f * v); } /* math.rs:133-135 */
inline float lerpf(float from, float to, float s) { return from + (to - from) * s; } /* math.rs:154-156 */

inline V3 offset_hit_point(V3 p, V3 n) { /* math.rs:137-152 */
    const float ORIGIN = 1.0f / 32.0f;
    const float INT_SCALE = 256.0f;
    const float FLOAT_SCALE = 1.0f / 65536.0f;
    int32_t of_i_x = sat_i32(n.x * INT_SCALE);
    int32_t of_i_y = sat_i32(n.y * INT_SCALE);
    int32_t of_i_z = sat_i32(n.z * INT_SCALE);
    /* wrapping i32 add, as release-mode Rust does */
    float p_i_x = u2f((uint32_t)((uint32_t)f2u(p.x) + (uint32_t)(p.x < 0.0f ? -of_i_x : of_i_x)));
    float p_i_y = u2f((uint32_t)((uint32_t)f2u(p.y) + (uint32_t)(p.y < 0.0f ? -of_i_y : of_i_y)));
    float p_i_z = u2f((uint32_t)((uint32_t)f2u(p.z) + (uint32_t)(p.z < 0.0f ? -of_i_z : of_i_z)));
    float x = std::fabs(p.x) < ORIGIN ? p.x + n.x * FLOAT_SCALE : p_i_x;
    float y = std::fabs(p.y) < ORIGIN ? p.y + n.y * FLOAT_SCALE : p_i_y;
    float z = std::fabs(p.z) < ORIGIN ? p.z + n.z * FLOAT_SCALE : p_i_z;
    return v3(x, y, z);
}

struct AABB {
    V3 mn, mx;
}; /* math.rs:90-94 */
inline bool aabb_hit(const AABB& b, const Ray& r, float t_min, float t_max) { /* math.rs:97-113 */
    for (int i = 0; i < 3; ++i) {
        float inv_d = 1.0f / comp(r.d, i);
        float t0 = (comp(b.mn, i) - comp(r.o, i)) * inv_d;
        float t1 = (comp(b.mx, i) - comp(r.o, i)) * inv_d;
        if (inv_d < 0.0f) std::swap(t0, t1);
        t_min = std::fmax(t_min, t0);
        t_max = std::fmin(t_max, t1);
        if (t_max <= t_min) return false;
    }
    return true;
}
inline AABB aabb_surround(const AABB& a, const AABB& b) { /* math.rs:115-130 */
    AABB r;
    r.mn = v3(std::fmin(a.mn.x, b.mn.x), std::fmin(a.mn.y, b.mn.y), std::fmin(a.mn.z, b.mn.z));
    r.mx = v3(std::fmax(a.mx.x, b.mx.x), std::fmax(a.mx.y, b.mx.y), std::fmax(a.mx.z, b.mx.z));
    return r;
}

/* ------------------------------------------------------------------------------------------
 * camera.rs
 * ---------------------------------------------------------------------------------------- */
RtCamera camera_new(V3 lookfrom, V3 lookat, V3 vup, float vfov, float aspect_ratio) { /* camera.rs:14-39 */
    V3 origin = lookfrom;
    const float RADS_PER_DEG = PI_F / 180.0f; /* f32::to_radians */
    float theta = vfov * RADS_PER_DEG;
    float viewport_height = std::tan(theta / 2.0f) * 2.0f;
    float viewport_width = viewport_height * aspect_ratio;
    V3 w = normalize(lookfrom - lookat);
    V3 u = normalize(cross(vup, w));
    V3 v = cross(w, u);
    V3 horizontal = viewport_width * u;
    V3 vertical = viewport_height * v;
    V3 llc = origin - horizontal / 2.0f - vertical / 2.0f - w;
    RtCamera c;
    c.origin[0] = origin.x, c.origin[1] = origin.y, c.origin[2] = origin.z;
    c.horizontal[0] = horizontal.x, c.horizontal[1] = horizontal.y, c.horizontal[2] = horizontal.z;
    c.vertical[0] = vertical.x, c.vertical[1] = vertical.y, c.vertical[2] = vertical.z;
    c.lower_left_corner[0] = llc.x, c.lower_left_corner[1] = llc.y, c.lower_left_corner[2] = llc.z;
    return c;
}
inline V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
inline Ray camera_get_ray(const RtCamera& c, float u, float v) { /* camera.rs:40-46 */
    V3 origin = ld3(c.origin), H = ld3(c.horizontal), V = ld3(c.vertical), llc = ld3(c.lower_left_corner);
    Ray r;
    r.o = origin;
    r.d = normalize(llc + u * H + v * V - origin);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * hitable.rs — HitRecord, Sphere, HitableList, BvhNode
 * ---------------------------------------------------------------------------------------- */
struct HitRecord { /* hitable.rs:13-22 */
    V3 p{0, 0, 0}, norm{0, 0, 0}, tang{0, 0, 0};
    float t = 0.0f;
    bool front_face = false;
    int mat = -1;   /* Option<Arc<dyn Material>> -> material index */
    int prim = -1;  /* sphere index (bookkeeping only) */
    V2 uv{0, 0};
};
inline void set_face_normal(HitRecord& rec, const Ray& r, V3 outward_normal) { /* hitable.rs:25-32 */
    rec.front_face = dot(r.d, outward_normal) < 0.0f;
    rec.norm = rec.front_face ? outward_normal : -outward_normal;
}
inline V3 world_to_local_with_rot(const HitRecord& rec, V3 v, float rot) { /* hitable.rs:37-41 */
    V3 tang = std::cos(rot) * rec.tang - std::sin(rot) * cross(rec.norm, rec.tang);
    V3 bitang = cross(rec.norm, tang);
    return v3(dot(v, tang), dot(v, bitang), dot(v, rec.norm));
}
inline V2 sphere_get_uv(V3 n) { /* hitable.rs:65-71 */
    float theta = std::acos(-n.y);
    float phi = std::atan2(-n.z, n.x) + PI_F;
    float u = phi / (2.0f * PI_F);
    float v = theta / PI_F;
    return V2{u, v};
}

struct Scene; /* fwd */

struct SceneView {
    const RtFlatScene* fs;
};

inline bool sphere_hit(const RtFlatScene& fs, int idx, const Ray& r, float t_min, float t_max,
                       HitRecord& rec) { /* hitable.rs:75-102 */
    V3 c = v3(fs.sph_cx[idx], fs.sph_cy[idx], fs.sph_cz[idx]);
    float rad = fs.sph_r[idx];
    V3 oc = r.o - c;
    float a = length_squared(r.d);
    float half_b = dot(oc, r.d);
    float cc = length_squared(oc) - rad * rad;
    float discriminant = half_b * half_b - a * cc;
    if (discriminant < 0.0f) return false;
    float sqrtd = std::sqrt(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    rec.t = root;
    rec.p = ray_at(r, rec.t);
    V3 outward_normal = (rec.p - c) / rad;
    rec.tang = normalize(cross(v3(0.0f, 1.0f, 0.0f), outward_normal));
    set_face_normal(rec, r, outward_normal);
    rec.uv = sphere_get_uv(outward_normal);
    rec.mat = (int)fs.sph_mat[idx];
    rec.prim = idx;
    return true;
}

/* hitable.rs:244-362 XYRect / XZRect / YZRect::hit.  `axis` is the constant coordinate; (a, b) are the
 * two in-plane axes in the order the reference tests them and builds uv from.  `rec.tang` is NOT written
 * (hitable.rs:262-269): it keeps whatever an earlier candidate left there. */
inline bool rect_hit(const RtFlatScene& fs, int ridx, const Ray& r, float t_min, float t_max, HitRecord& rec) {
    const int axis = fs.rect_axis[ridx];
    const V3 mn = ld3(fs.rect_min + 3 * ridx), mx = ld3(fs.rect_max + 3 * ridx);
    float t = (comp(mn, axis) - comp(r.o, axis)) / comp(r.d, axis);
    if (t < t_min || t > t_max) return false;
    V3 p = ray_at(r, t);
    if (axis == RT_RECT_XY) {
        if (p.x < mn.x || p.x > mx.x || p.y < mn.y || p.y > mx.y) return false;
    } else if (axis == RT_RECT_XZ) {
        if (p.x < mn.x || p.x > mx.x || p.z < mn.z || p.z > mx.z) return false;
    } else {
        if (p.z < mn.z || p.z > mx.z || p.y < mn.y || p.y > mx.y) return false;
    }
    V3 uv = (p - mn) / (mx - mn);
    rec.uv = axis == RT_RECT_XY ? V2{uv.x, uv.y} : (axis == RT_RECT_XZ ? V2{uv.x, uv.z} : V2{uv.y, uv.z});
    rec.p = p;
    rec.t = t;
    V3 outward_normal = axis == RT_RECT_XY ? v3(0, 0, 1) : (axis == RT_RECT_XZ ? v3(0, 1, 0) : v3(1, 0, 0));
    set_face_normal(rec, r, outward_normal);
    rec.mat = (int)fs.rect_mat[ridx];
    rec.prim = (int)fs.n_spheres + ridx;
    return true;
}
inline AABB rect_bbox(const RtFlatScene& fs, int ridx) { /* hitable.rs:274-278, 314-318, 354-358 */
    const int axis = fs.rect_axis[ridx];
    V3 mn = ld3(fs.rect_min + 3 * ridx), mx = ld3(fs.rect_max + 3 * ridx);
    if (axis == RT_RECT_XY) mn.z = mn.z - 0.0001f, mx.z = mx.z + 0.0001f;
    else if (axis == RT_RECT_XZ) mn.y = mn.y - 0.0001f, mx.y = mx.y + 0.0001f;
    else mn.x = mn.x - 0.0001f, mx.x = mx.x + 0.0001f;
    return AABB{mn, mx};
}
/* primitive i: sphere i for i < n_spheres, else rect i - n_spheres (the flat scene's tie order) */
inline bool prim_hit(const RtFlatScene& fs, int i, const Ray& r, float t_min, float t_max, HitRecord& rec) {
    return i < (int)fs.n_spheres ? sphere_hit(fs, i, r, t_min, t_max, rec) : rect_hit(fs, i - (int)fs.n_spheres, r, t_min, t_max, rec);
}

/* ---- instance wrappers, hitable.rs:404-520 ------------------------------------------------- */
inline uint32_t prim_xform(const RtFlatScene& fs, int i) {
    if (i < (int)fs.n_spheres) return fs.sph_xform ? fs.sph_xform[i] : RT_NO_XFORM;
    return fs.rect_xform ? fs.rect_xform[i - (int)fs.n_spheres] : RT_NO_XFORM;
}
inline Ray xform_ray(const RtFlatScene& fs, uint32_t x, const Ray& r) {
    const float* q = fs.xf_param + 4 * (size_t)x;
    if (fs.xf_type[x] == RT_XF_TRANSLATE) return Ray{r.o - v3(q[0], q[1], q[2]), r.d}; /* hitable.rs:411 */
    const float sin_theta = q[0], cos_theta = q[1]; /* hitable.rs:483-492 */
    V3 oo = r.o, dd = r.d;
    oo.x = cos_theta * r.o.x - sin_theta * r.o.z;
    oo.z = sin_theta * r.o.x + cos_theta * r.o.z;
    dd.x = cos_theta * r.d.x - sin_theta * r.d.z;
    dd.z = sin_theta * r.d.x + cos_theta * r.d.z;
    return Ray{oo, dd};
}
/* `inner` is the ray the wrapper handed to its child (moved_r / rot_r) */
inline void xform_fix_record(const RtFlatScene& fs, uint32_t x, const Ray& inner, HitRecord& rec) {
    const float* q = fs.xf_param + 4 * (size_t)x;
    if (fs.xf_type[x] == RT_XF_TRANSLATE) { /* hitable.rs:413: only rec.p moves */
        rec.p = rec.p + v3(q[0], q[1], q[2]);
        return;
    }
    const float sin_theta = q[0], cos_theta = q[1]; /* hitable.rs:495-505 */
    V3 p = rec.p, n = rec.norm;
    p.x = cos_theta * rec.p.x + sin_theta * rec.p.z;
    p.z = -sin_theta * rec.p.x + cos_theta * rec.p.z;
    n.x = cos_theta * rec.norm.x + sin_theta * rec.norm.z;
    n.z = -sin_theta * rec.norm.x + cos_theta * rec.norm.z;
    rec.p = p;
    set_face_normal(rec, inner, n); /* quirk kept: the object-space ray against the world-space normal */
}
/* a primitive below its chain of wrappers: ray outside-in, record inside-out.  `stop`: the wrapper the caller is already inside of
 * (the one around the ConstantMedium whose boundary this primitive belongs to), where the chain ends for this call */
inline bool prim_hit_x(const RtFlatScene& fs, int i, const Ray& r, float t_min, float t_max, HitRecord& rec, uint32_t stop = RT_NO_XFORM) {
    if (prim_xform(fs, i) == stop) return prim_hit(fs, i, r, t_min, t_max, rec);
    /* any depth of nesting, as the trait objects allow (hitable.rs:404-520); the buffers are per thread and keep their capacity */
    thread_local std::vector<uint32_t> chain;
    thread_local std::vector<Ray> rays;
    chain.clear();
    for (uint32_t x = prim_xform(fs, i); x != stop && x != RT_NO_XFORM; x = fs.xf_parent[x]) chain.push_back(x);
    const int n = (int)chain.size();
    rays.resize((size_t)n + 1);
    rays[n] = r;
    for (int k = n - 1; k >= 0; --k) rays[k] = xform_ray(fs, chain[k], rays[k + 1]);
    if (!prim_hit(fs, i, rays[0], t_min, t_max, rec)) return false;
    for (int k = 0; k < n; ++k) xform_fix_record(fs, chain[k], rays[k], rec);
    return true;
}
/* world-space bounds of a wrapped primitive: corners through the wrappers, as RotateY::new does (hitable.rs:455-473) */
inline AABB xform_bbox(const RtFlatScene& fs, uint32_t x, const AABB& b) {
    const float* q = fs.xf_param + 4 * (size_t)x;
    if (fs.xf_type[x] == RT_XF_TRANSLATE) { /* hitable.rs:420-431 */
        V3 off = v3(q[0], q[1], q[2]);
        return AABB{b.mn + off, b.mx + off};
    }
    const float sin_theta = q[0], cos_theta = q[1];
    V3 mn = splat(INFINITY), mx = splat(-INFINITY);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                float x0 = i == 0 ? b.mn.x : b.mx.x, y0 = j == 0 ? b.mn.y : b.mx.y, z0 = k == 0 ? b.mn.z : b.mx.z;
                float nx = cos_theta * x0 + sin_theta * z0, nz = -sin_theta * x0 + cos_theta * z0;
                mn = v3(std::fmin(mn.x, nx), std::fmin(mn.y, y0), std::fmin(mn.z, nz));
                mx = v3(std::fmax(mx.x, nx), std::fmax(mx.y, y0), std::fmax(mx.z, nz));
            }
    return AABB{mn, mx};
}

/* ---- ConstantMedium, hitable.rs:523-588 ---------------------------------------------------- */
inline uint32_t prim_medium(const RtFlatScene& fs, int i) {
    if (i < (int)fs.n_spheres) return fs.sph_medium ? fs.sph_medium[i] : RT_NO_MEDIUM;
    return fs.rect_medium ? fs.rect_medium[i - (int)fs.n_spheres] : RT_NO_MEDIUM;
}
/* boundary.hit(r, t_min, t_max, rec): the boundary object is the list of primitives tagged with medium m
 * (a GBox's sides through its wrappers, or one sphere): closest hit in list order */
inline uint32_t medium_xform(const RtFlatScene& fs, uint32_t m) { return fs.med_xform ? fs.med_xform[m] : RT_NO_XFORM; }
inline bool boundary_hit(const RtFlatScene& fs, uint32_t m, const Ray& r, float t_min, float t_max, HitRecord& rec) {
    HitRecord temp_rec;
    float closest_so_far = t_max;
    bool hit_anything = false;
    for (uint32_t i = 0; i < fs.n_spheres + fs.n_rects; ++i) {
        if (prim_medium(fs, (int)i) != m) continue;
        if (prim_hit_x(fs, (int)i, r, t_min, closest_so_far, temp_rec, medium_xform(fs, m))) { /* r: the ray the medium received */
            hit_anything = true;
            closest_so_far = temp_rec.t;
        }
    }
    if (hit_anything) rec = temp_rec;
    return hit_anything;
}
inline bool medium_hit(const RtFlatScene& fs, uint32_t m, const Ray& r, float t_min, float t_max, HitRecord& rec, Rng& rng) {
    /* hitable.rs:536-579 (ENABLE_DEBUGGING = false: the debugging draw is short-circuited away) */
    HitRecord rec_1, rec_2;
    if (!boundary_hit(fs, m, r, -INFINITY, INFINITY, rec_1)) return false;
    if (!boundary_hit(fs, m, r, rec_1.t + 0.0001f, INFINITY, rec_2)) return false;
    if (rec_1.t < t_min) rec_1.t = t_min;
    if (rec_2.t > t_max) rec_2.t = t_max;
    if (rec_1.t >= rec_2.t) return false;
    if (rec_1.t < 0.0f) rec_1.t = 0.0f;
    float ray_len = length(r.d);
    float dist_inside_boundary = (rec_2.t - rec_1.t) * ray_len;
    float hit_dist = fs.med_neg_inv_density[m] * std::log(rng_medium_draw(rng, m));
    if (hit_dist > dist_inside_boundary) return false;
    rec.t = rec_1.t + hit_dist / ray_len;
    rec.p = ray_at(r, rec.t);
    rec.norm = v3(1.0f, 0.0f, 0.0f);
    rec.front_face = true;
    rec.mat = (int)fs.med_mat[m];
    rec.prim = (int)(fs.n_spheres + fs.n_rects + m);
    return true; /* uv and tang keep whatever an earlier candidate left (not written, hitable.rs:574-576) */
}

/* a medium below wrappers — Translate { ptr: ConstantMedium } and so on, hitable.rs:409-416, 479-509: the medium's hit() sees the
 * moved ray, the wrappers fix its record on the way out */
inline bool medium_hit_x(const RtFlatScene& fs, uint32_t m, const Ray& r, float t_min, float t_max, HitRecord& rec, Rng& rng) {
    if (medium_xform(fs, m) == RT_NO_XFORM) return medium_hit(fs, m, r, t_min, t_max, rec, rng);
    std::vector<uint32_t> chain; /* (not the per-thread buffers of prim_hit_x: the boundary searches inside use those) */
    for (uint32_t x = medium_xform(fs, m); x != RT_NO_XFORM; x = fs.xf_parent[x]) chain.push_back(x);
    const int n = (int)chain.size();
    std::vector<Ray> rays((size_t)n + 1);
    rays[n] = r;
    for (int k = n - 1; k >= 0; --k) rays[k] = xform_ray(fs, chain[k], rays[k + 1]);
    if (!medium_hit(fs, m, rays[0], t_min, t_max, rec, rng)) return false;
    for (int k = 0; k < n; ++k) xform_fix_record(fs, chain[k], rays[k], rec);
    return true;
}

/* hitable.rs:117-132 HitableList::hit over the flat primitive list (world order). */
/* any world entry: primitive i < n_spheres + n_rects (unless it only bounds a medium), else medium */
inline bool entry_hit(const RtFlatScene& fs, int i, const Ray& r, float t_min, float t_max, HitRecord& rec, Rng& rng) {
    const int np = (int)(fs.n_spheres + fs.n_rects);
    if (i >= np) return medium_hit_x(fs, (uint32_t)(i - np), r, t_min, t_max, rec, rng);
    if (prim_medium(fs, i) != RT_NO_MEDIUM) return false;
    return prim_hit_x(fs, i, r, t_min, t_max, rec);
}
inline bool list_hit(const RtFlatScene& fs, const Ray& r, float t_min, float t_max, HitRecord& rec, Rng& rng) {
    HitRecord temp_rec;
    float closest_so_far = t_max;
    bool hit_anything = false;
    for (uint32_t i = 0; i < fs.n_spheres + fs.n_rects + fs.n_media; ++i) {
        if (entry_hit(fs, (int)i, r, t_min, closest_so_far, temp_rec, rng)) {
            hit_anything = true;
            closest_so_far = temp_rec.t;
        }
    }
    if (hit_anything) rec = temp_rec;
    return hit_anything;
}

/* hitable.rs:158-241 BvhNode.  child >= 0: node index; child < 0: sphere ~child. */
struct BvhNode {
    AABB box;
    int left, right;
};
struct Bvh {
    std::vector<BvhNode> nodes;
    int root = -1;
};
inline AABB sphere_bbox(const RtFlatScene& fs, int idx); /* fwd */
inline AABB medium_bbox(const RtFlatScene& fs, uint32_t m) { /* hitable.rs:581-583: the boundary's box */
    AABB b{splat(INFINITY), splat(-INFINITY)};
    for (uint32_t i = 0; i < fs.n_spheres + fs.n_rects; ++i)
        if (prim_medium(fs, (int)i) == m) b = aabb_surround(b, sphere_bbox(fs, (int)i));
    return b;
}
inline AABB sphere_bbox(const RtFlatScene& fs, int idx) { /* hitable.rs:104-108; wrapped primitives: world bounds */
    AABB b;
    if (idx >= (int)(fs.n_spheres + fs.n_rects)) return medium_bbox(fs, (uint32_t)idx - fs.n_spheres - fs.n_rects);
    if (idx >= (int)fs.n_spheres) {
        b = rect_bbox(fs, idx - (int)fs.n_spheres);
    } else {
        V3 c = v3(fs.sph_cx[idx], fs.sph_cy[idx], fs.sph_cz[idx]);
        float r = fs.sph_r[idx];
        b = AABB{c - r, c + r};
    }
    for (uint32_t x = prim_xform(fs, idx); x != RT_NO_XFORM; x = fs.xf_parent[x]) b = xform_bbox(fs, x, b);
    return b;
}
/* f32::total_cmp key */
inline int32_t total_key(float f) {
    int32_t b = (int32_t)f2u(f);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}
int bvh_build(const RtFlatScene& fs, Bvh& bvh, std::vector<int>& objects, size_t start, size_t end,
              Xoshiro256pp& rng) { /* hitable.rs:177-221 */
    int axis = (int)gen_range_u64(rng, 0, 3);
    size_t span = end - start;
    int left, right;
    AABB box_a, box_b;
    auto child_box = [&](int c) { return c >= 0 ? bvh.nodes[(size_t)c].box : sphere_bbox(fs, ~c); };
    if (span == 1) {
        left = right = ~objects[start];
    } else {
        std::stable_sort(objects.begin() + (long)start, objects.begin() + (long)end, [&](int a, int b) {
            return total_key(comp(sphere_bbox(fs, a).mn, axis)) < total_key(comp(sphere_bbox(fs, b).mn, axis));
        });
        if (span == 2) {
            left = ~objects[start];
            right = ~objects[start + 1];
        } else {
            size_t mid = start + span / 2;
            left = bvh_build(fs, bvh, objects, start, mid, rng);
            right = bvh_build(fs, bvh, objects, mid, end, rng);
        }
    }
    box_a = child_box(left);
    box_b = child_box(right);
    BvhNode n;
    n.box = aabb_surround(box_a, box_b);
    n.left = left;
    n.right = right;
    bvh.nodes.push_back(n);
    return (int)bvh.nodes.size() - 1;
}
bool bvh_hit(const RtFlatScene& fs, const Bvh& bvh, int node, const Ray& r, float t_min, float t_max,
             HitRecord& rec, Rng& rng) { /* hitable.rs:232-240 */
    if (node < 0) return entry_hit(fs, ~node, r, t_min, t_max, rec, rng);
    const BvhNode& n = bvh.nodes[(size_t)node];
    if (!aabb_hit(n.box, r, t_min, t_max)) return false;
    /* A node over one object holds it as both children (hitable.rs:188) and calls it twice.  That is idempotent
     * for a surface; a ConstantMedium draws a second free path and the nearer wins, i.e. its density doubles.
     * Which media sit alone in a node is a property of the scene's own (nested) tree, so the flattening folds
     * that multiplicity into med_neg_inv_density (host/rtow.hpp ConstantMedium::flatten) and this tree over the
     * flat entries must not add one of its own: */
    if (n.left == n.right) return bvh_hit(fs, bvh, n.left, r, t_min, t_max, rec, rng);
    bool hit_left = bvh_hit(fs, bvh, n.left, r, t_min, t_max, rec, rng);
    bool hit_right = bvh_hit(fs, bvh, n.right, r, t_min, hit_left ? rec.t : t_max, rec, rng);
    return hit_left || hit_right;
}

/* ------------------------------------------------------------------------------------------
 * texture.rs
 * ---------------------------------------------------------------------------------------- */
inline long rem_euclid_256(int64_t v) {
    int64_t r = v % 256;
    if (r < 0) r += 256;
    return (long)r;
}
float perlin_noise(const RtFlatScene& fs, uint32_t set, V3 p) { /* texture.rs:125-146 */
    const float* rv = fs.perlin_vec + (size_t)set * 256 * 3;
    const uint16_t* px = fs.perlin_perm + (size_t)set * 3 * 256;
    const uint16_t* py = px + 256;
    const uint16_t* pz = py + 256;
    int64_t i = sat_i64(std::floor(p.x));
    int64_t j = sat_i64(std::floor(p.y));
    int64_t k = sat_i64(std::floor(p.z));
    V3 c[2][2][2];
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                /* `i + di` on isize: wraps at isize::MAX in a release build (a debug build panics), so the sum is taken unsigned */
                auto plus = [](int64_t a, int b) { return (int64_t)((uint64_t)a + (uint64_t)b); };
                unsigned index = (unsigned)px[rem_euclid_256(plus(i, di))] ^ (unsigned)py[rem_euclid_256(plus(j, dj))] ^
                                 (unsigned)pz[rem_euclid_256(plus(k, dk))];
                c[di][dj][dk] = v3(rv[3 * index], rv[3 * index + 1], rv[3 * index + 2]);
            }
    V3 uvw = p - v3(std::floor(p.x), std::floor(p.y), std::floor(p.z));
    /* trilinear_interp, texture.rs:93-112 */
    float accum = 0.0f;
    V3 uvw2 = smooth(uvw);
    float u = uvw2.x, v = uvw2.y, w = uvw2.z;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int d = 0; d < 2; ++d) {
                V3 weight = uvw - v3((float)a, (float)b, (float)d);
                accum += dot(c[a][b][d], weight) * (a == 1 ? u : 1.0f - u) * (b == 1 ? v : 1.0f - v) *
                         (d == 1 ? w : 1.0f - w);
            }
    return accum;
}
float perlin_turb(const RtFlatScene& fs, uint32_t set, V3 p) { /* texture.rs:115-124 */
    float accum = 0.0f;
    float w = 1.0f;
    for (int it = 0; it < 7; ++it) {
        accum += w * perlin_noise(fs, set, p);
        p = p * 2.0f;
        w *= 0.5f;
    }
    return std::fabs(accum);
}
V3 image_value(const RtFlatScene& fs, uint32_t img, V2 uv, uint64_t* fetches) { /* texture.rs:183-193 */
    uint32_t W = fs.img_w[img], H = fs.img_h[img];
    float u = clampf(uv.x, 0.0f, 1.0f);
    float v = 1.0f - clampf(uv.y, 0.0f, 1.0f);
    uint32_t i = std::min(sat_u32(u * (float)W), W - 1);
    uint32_t j = std::min(sat_u32(v * (float)H), H - 1);
    const float* px = fs.texels + fs.img_offset[img] + ((size_t)j * W + i) * 3;
    if (fetches) ++*fetches;
    return v3(px[0], px[1], px[2]);
}
V3 texture_value(const RtFlatScene& fs, uint32_t tex, V2 uv, V3 p, uint64_t* fetches) {
    switch (fs.tex_type[tex]) {
    case RT_TEX_CONSTANT: /* texture.rs:19-23 */
        return ld3(fs.tex_color0 + 3 * tex);
    case RT_TEX_CHECKER: { /* texture.rs:40-49 */
        float sines = std::sin(p.x * 10.0f) * std::sin(p.y * 10.0f) * std::sin(p.z * 10.0f);
        return sines < 0.0f ? ld3(fs.tex_color0 + 3 * tex) : ld3(fs.tex_color1 + 3 * tex);
    }
    case RT_TEX_PERLIN: { /* texture.rs:164-168 */
        float s = std::sin(10.0f * perlin_turb(fs, fs.tex_aux[tex], p) + fs.tex_scale[tex] * p.z);
        return (s + 1.0f) * 0.5f * splat(1.0f);
    }
    case RT_TEX_IMAGE:
        return image_value(fs, fs.tex_aux[tex], uv, fetches);
    default:
        return splat(0.0f);
    }
}

/* ------------------------------------------------------------------------------------------
 * demo_scene.rs:22-35 sky models
 * ---------------------------------------------------------------------------------------- */
V3 sky_value(const RtFlatScene& fs, V3 d, uint64_t* fetches) {
    switch (fs.sky_type) {
    case RT_SKY_GRADIENT: { /* demo_scene.rs:28-31 */
        float t = d.y * 0.5f + 0.5f;
        return lerp3(splat(1.0f), v3(0.5f, 0.7f, 1.0f), t);
    }
    case RT_SKY_ENV: { /* demo_scene.rs:22-26 */
        V2 uv = sphere_get_uv(d);
        V3 c = image_value(fs, fs.sky_image, V2{1.0f - uv.x, uv.y}, fetches);
        return c * c;
    }
    default: /* black_sky demo_scene.rs:33-35 */
        return splat(0.0f);
    }
}

/* ------------------------------------------------------------------------------------------
 * pbr.rs helpers
 * ---------------------------------------------------------------------------------------- */
inline float gtr1(float n_dot_h, float a) { /* pbr.rs:72-79 */
    if (a >= 1.0f) return FRAC_1_PI_F;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * n_dot_h * n_dot_h;
    return (a2 - 1.0f) / (PI_F * std::log(a2) * t);
}
inline float gtr2(float n_dot_h, float a) { /* pbr.rs:81-85 */
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * n_dot_h * n_dot_h;
    return a2 / (PI_F * t * t);
}
inline float gtr2_aniso(V3 h, float ax, float ay) { /* pbr.rs:87-89 */
    return 1.0f / (PI_F * ax * ay * powi2(powi2(h.x / ax) + powi2(h.y / ay) + h.z * h.z));
}
inline float smith_geo_ggx(float n_dot_v, float alpha) { /* pbr.rs:92-96 */
    float a = alpha * alpha;
    float b = n_dot_v * n_dot_v;
    return 1.0f / (n_dot_v + std::sqrt(a + b - a * b));
}
inline float smith_geo_ggx_aniso(V3 v, float ax, float ay) { /* pbr.rs:98-100 */
    return 1.0f / (v.z + std::sqrt(powi2(v.x * ax) + powi2(v.y * ay) + v.z * v.z));
}
inline float fresnel_dielectric(float n_dot_i, float n_dot_t, float eta) { /* pbr.rs:107-113 (assert dropped) */
    float rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
    float rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
    return (rs * rs + rp * rp) / 2.0f;
}
inline float fresnel_dielectric_2(float n_dot_i, float eta) { /* pbr.rs:120-129 */
    float n_dot_t_sq = 1.0f - (1.0f - n_dot_i * n_dot_i) / (eta * eta);
    if (n_dot_t_sq < 0.0f) return 1.0f;
    float n_dot_t = std::sqrt(n_dot_t_sq);
    return fresnel_dielectric(std::fabs(n_dot_i), n_dot_t, eta);
}
inline float smith_masking_gtr2_2(V3 v_world, V3 n, float roughness) { /* pbr.rs:145-152 */
    float alpha = roughness * roughness;
    float a2 = alpha * alpha;
    float v2_z_ = dot(v_world, n);
    float v2_z = v2_z_ * v2_z_;
    float lambda = (-1.0f + std::sqrt(1.0f + a2 * (1.0f - v2_z) / v2_z)) / 2.0f;
    return 1.0f / (1.0f + lambda);
}

/* ------------------------------------------------------------------------------------------
 * material.rs + pbr.rs: emitted / scatter
 * ---------------------------------------------------------------------------------------- */
struct Ctx {
    const RtFlatScene* fs;
    const Bvh* bvh;   /* nullptr => HitableList linear walk */
    int max_depth;
    bool russian_roulette = false; /* main.rs:49-53, commented out in the reference (RT_FLAG_RUSSIAN_ROULETTE) */
    uint64_t n_rays = 0;
    uint64_t n_tex = 0;
    uint64_t n_bad = 0;
    uint64_t per_depth[64] = {0};
};

V3 mat_emitted(Ctx& cx, int m, V2 uv, V3 p) { /* material.rs:12-14, 26-28 */
    const RtFlatScene& fs = *cx.fs;
    if (fs.mat_type[m] == RT_MAT_EMISSION) return texture_value(fs, fs.mat_tex0[m], uv, p, &cx.n_tex);
    return splat(0.0f);
}

bool mat_scatter(Ctx& cx, Rng& rng, int m, const Ray& r_in, const HitRecord& rec, V3& attenuation,
                 Ray& scattered) {
    const RtFlatScene& fs = *cx.fs;
    const uint32_t t0 = fs.mat_tex0[m], t1 = fs.mat_tex1[m];
    const float p0 = fs.mat_p0[m], p1 = fs.mat_p1[m], p2 = fs.mat_p2[m];
    switch (fs.mat_type[m]) {
    case RT_MAT_EMISSION: /* material.rs:21-24 */
        return false;
    case RT_MAT_DIFFUSE: { /* material.rs:35-46 */
        V3 scatter_direction = rec.norm + normalize(random_in_unit_sphere(rng));
        if (vec3a_near_zero(scatter_direction)) scatter_direction = rec.norm;
        V3 p = offset_hit_point(rec.p, rec.norm);
        scattered = Ray{p, normalize(scatter_direction)};
        attenuation = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex);
        return true;
    }
    case RT_MAT_LAMBERT: { /* material.rs:52-59 */
        V3 p = offset_hit_point(rec.p, rec.norm);
        scattered = Ray{p, random_on_hemisphere(rng, rec.norm)};
        attenuation = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex) * 2.0f * dot(rec.norm, scattered.d);
        return true;
    }
    case RT_MAT_METAL: { /* material.rs:66-73 */
        V3 reflected = reflect(r_in.d, rec.norm) + p0 * random_in_unit_sphere(rng);
        scattered = Ray{rec.p, normalize(reflected)};
        attenuation = ld3(fs.mat_color + 3 * m);
        return dot(reflected, rec.norm) > 0.0f;
    }
    case RT_MAT_DIELECTRIC: { /* material.rs:79-97 */
        attenuation = splat(1.0f);
        float ref_idx = rec.front_face ? 1.0f / p0 : p0;
        float cos_theta = -std::fmin(dot(r_in.d, rec.norm), 1.0f);
        float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
        bool cannot_refract = sin_theta * ref_idx > 1.0f;
        float rnd_num = rng.next_f32();
        V3 dir = (cannot_refract || reflectance(cos_theta, ref_idx) > rnd_num) ? reflect(r_in.d, rec.norm)
                                                                              : refract(r_in.d, rec.norm, ref_idx);
        scattered = Ray{rec.p, normalize(dir)};
        return true;
    }
    case RT_MAT_ISOTROPIC: { /* material.rs:103-113 */
        scattered = Ray{rec.p, random_on_unit_sphere(rng)};
        attenuation = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex);
        return true;
    }
    case RT_MAT_OREN_NAYAR: { /* pbr.rs:17-42 */
        V3 p = offset_hit_point(rec.p, rec.norm);
        V3 dir_o = random_on_hemisphere(rng, rec.norm);
        float cos_i = std::fabs(dot(rec.norm, r_in.d));
        float cos_o = dot(rec.norm, dir_o);
        float sin_i = std::sqrt(1.0f - cos_i * cos_i);
        float sin_o = std::sqrt(1.0f - cos_o * cos_o);
        float max_cos = std::fmax(cos_i * cos_o + sin_i * sin_o, 0.0f);
        float r2 = p0 * p0;
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
        scattered = Ray{p, dir_o};
        attenuation = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex) * w * 2.0f * cos_o;
        return true;
    }
    case RT_MAT_BURLEY_DIFFUSE: { /* pbr.rs:50-69 */
        V3 p = offset_hit_point(rec.p, rec.norm);
        V3 dir_o = random_on_hemisphere(rng, rec.norm);
        float n_dot_i = dot(rec.norm, -r_in.d);
        float n_dot_o = dot(rec.norm, dir_o);
        V3 h = normalize(dir_o - r_in.d);
        float h_dot_o = dot(h, dir_o);
        float fl = schlick_fresnel(n_dot_o);
        float fv = schlick_fresnel(n_dot_i);
        float fd90 = 0.5f + 2.0f * h_dot_o * h_dot_o * p0;
        float fd = lerpf(1.0f, fd90, fl) * lerpf(1.0f, fd90, fv);
        scattered = Ray{p, dir_o};
        attenuation = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex) * fd * 2.0f * n_dot_o;
        return true;
    }
    case RT_MAT_ROUGH_PLASTIC: { /* pbr.rs:160-189 */
        V3 p = offset_hit_point(rec.p, rec.norm);
        V3 dir_o = random_on_hemisphere(rng, rec.norm);
        float n_dot_i = dot(rec.norm, -r_in.d);
        float n_dot_o = dot(rec.norm, dir_o);
        V3 h = normalize(dir_o - r_in.d);
        float h_dot_i = dot(h, -r_in.d);
        float h_dot_o = dot(h, dir_o);
        float n_dot_h = dot(rec.norm, h);
        V3 kd = texture_value(fs, t1, rec.uv, rec.p, &cx.n_tex);
        V3 ks = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex);
        float roughness = clampf(p0, 0.01f, 1.0f);
        float eta = p1;
        float f_o = fresnel_dielectric_2(h_dot_o, eta);
        float d = gtr2(n_dot_h, roughness);
        float g = smith_masking_gtr2_2(-r_in.d, rec.norm, roughness) * smith_masking_gtr2_2(dir_o, rec.norm, roughness);
        V3 spec_contrib = ks * (g * f_o * d) / (4.0f * n_dot_i * n_dot_o);
        float f_i = fresnel_dielectric_2(h_dot_i, eta);
        V3 diff_contrib = kd * (1.0f - f_o) * (1.0f - f_i) * FRAC_1_PI_F;
        scattered = Ray{p, dir_o};
        attenuation = (spec_contrib + diff_contrib) * n_dot_o * 2.0f * PI_F;
        return true;
    }
    case RT_MAT_DISNEY_DIFFUSE: { /* pbr.rs:198-222 */
        V3 p = offset_hit_point(rec.p, rec.norm);
        V3 dir_o = random_on_hemisphere(rng, rec.norm);
        float n_dot_i = dot(rec.norm, -r_in.d);
        float n_dot_o = dot(rec.norm, dir_o);
        V3 h = normalize(dir_o - r_in.d);
        float h_dot_o = dot(h, dir_o);
        float fo = schlick_fresnel(n_dot_o);
        float fi = schlick_fresnel(n_dot_i);
        float fd90 = 0.5f + 2.0f * h_dot_o * h_dot_o * p0;
        float fd = lerpf(1.0f, fd90, fo) * lerpf(1.0f, fd90, fi);
        float fss90 = p0 * h_dot_o * h_dot_o;
        float fss_wi = lerpf(1.0f, fss90, fi);
        float fss_wo = lerpf(1.0f, fss90, fo);
        float fss = 1.25f * (fss_wi * fss_wo * (1.0f / (n_dot_i + n_dot_o) - 0.5f) + 0.5f);
        scattered = Ray{p, dir_o};
        attenuation = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex) * lerpf(fd, fss, p1) * 2.0f * n_dot_o;
        return true;
    }
    case RT_MAT_DISNEY_METAL: { /* pbr.rs:232-278 */
        V3 p = offset_hit_point(rec.p, rec.norm);
        V3 dir_o = random_on_hemisphere(rng, rec.norm);
        float n_dot_i = dot(rec.norm, -r_in.d);
        float n_dot_o = dot(rec.norm, dir_o);
        V3 h = normalize(dir_o - r_in.d);
        float h_dot_o = dot(h, dir_o);
        float n_dot_h = dot(rec.norm, h);
        V3 albedo = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex);
        V3 fm = lerp3(albedo, splat(1.0f), schlick_fresnel(h_dot_o));
        const float alpha_min = 0.0001f;
        float roughness = p0, anisotropic = p1;
        float dm, gm;
        if (anisotropic > -10.0f) {
            float aspect = std::sqrt(1.0f - 0.9f * anisotropic);
            float ax = std::fmax(roughness * roughness / aspect, alpha_min);
            float ay = std::fmax(roughness * roughness * aspect, alpha_min);
            float rot = p2 * 2.0f * PI_F;
            V3 h_local = world_to_local_with_rot(rec, h, rot);
            dm = gtr2_aniso(h_local, ax, ay);
            V3 i_local = world_to_local_with_rot(rec, -r_in.d, rot);
            V3 o_local = world_to_local_with_rot(rec, dir_o, rot);
            gm = smith_geo_ggx_aniso(i_local, ax, ay) * smith_geo_ggx_aniso(o_local, ax, ay);
        } else {
            float r2 = std::fmax(roughness * roughness, alpha_min);
            dm = gtr2(n_dot_h, r2);
            gm = smith_geo_ggx(n_dot_i, r2) * smith_geo_ggx(n_dot_o, r2);
        }
        V3 metal_w = fm * dm * gm;
        scattered = Ray{p, dir_o};
        attenuation = metal_w * n_dot_o * 2.0f * PI_F;
        return true;
    }
    case RT_MAT_DISNEY_SHEEN: { /* pbr.rs:286-308 */
        V3 p = offset_hit_point(rec.p, rec.norm);
        V3 dir_o = random_on_hemisphere(rng, rec.norm);
        float n_dot_o = dot(rec.norm, dir_o);
        V3 h = normalize(dir_o - r_in.d);
        float h_dot_o = dot(h, dir_o);
        V3 albedo = texture_value(fs, t0, rec.uv, rec.p, &cx.n_tex);
        float luminance = dot(v3(0.3f, 0.6f, 0.1f), albedo);
        V3 c_tint = luminance > 0.0f ? albedo / luminance : splat(1.0f);
        V3 c_sheen = lerp3(splat(1.0f), c_tint, p0);
        V3 f_sheen = c_sheen * schlick_fresnel(h_dot_o);
        scattered = Ray{p, dir_o};
        attenuation = f_sheen * n_dot_o * 2.0f * PI_F;
        return true;
    }
    case RT_MAT_DISNEY_CLEARCOAT: { /* pbr.rs:314-335 */
        V3 p = offset_hit_point(rec.p, rec.norm);
        V3 dir_o = random_on_hemisphere(rng, rec.norm);
        float n_dot_i = dot(rec.norm, -r_in.d);
        float n_dot_o = dot(rec.norm, dir_o);
        V3 h = normalize(dir_o - r_in.d);
        float h_dot_o = dot(h, dir_o);
        float n_dot_h = dot(rec.norm, h);
        float fc = lerpf(0.4f, 1.0f, schlick_fresnel(h_dot_o));
        float dc = gtr1(n_dot_h, lerpf(0.1f, 0.001f, p0));
        float gc = smith_geo_ggx(n_dot_i, 0.25f) * smith_geo_ggx(n_dot_o, 0.25f);
        float cc = 0.25f * fc * dc * gc;
        scattered = Ray{p, dir_o};
        attenuation = splat(cc) * n_dot_o * 2.0f * PI_F;
        return true;
    }
    default:
        return false;
    }
}

inline bool world_hit(Ctx& cx, Rng& rng, const Ray& r, float t_min, float t_max, HitRecord& rec) {
    if (cx.bvh) {
        /* world = vec![BvhNode] (demo_scene.rs:223-227) walked by HitableList::hit */
        HitRecord temp_rec;
        bool h = bvh_hit(*cx.fs, *cx.bvh, cx.bvh->root, r, t_min, t_max, temp_rec, rng);
        if (h) rec = temp_rec;
        return h;
    }
    return list_hit(*cx.fs, r, t_min, t_max, rec, rng);
}

/* main.rs:38-60 ray_color, recursive exactly as the reference. */
V3 ray_color(Ctx& cx, Rng& rng, const Ray& r, int depth) {
    if (!vec3a_near_one(r.d)) { /* main.rs:39 assert!: the reference panics; we drop the path */
        ++cx.n_bad;
        return splat(0.0f);
    }
    if (depth > cx.max_depth) return splat(0.0f);
    ++cx.n_rays;
    if (depth < 64) ++cx.per_depth[depth];
    rng.set_depth(depth);
    HitRecord rec;
    if (world_hit(cx, rng, r, 1e-3f, std::numeric_limits<float>::max(), rec)) {
        Ray scattered{splat(0.0f), splat(0.0f)};
        V3 attenuation = splat(1.0f);
        V3 ret = mat_emitted(cx, rec.mat, rec.uv, rec.p);
        if (mat_scatter(cx, rng, rec.mat, r, rec, attenuation, scattered)) {
            if (cx.russian_roulette) { /* main.rs:49-53 */
                float russian_roulette = rng.next_f32();
                float threshold = std::fmax(attenuation.x, std::fmax(attenuation.y, attenuation.z)); /* max_element */
                if (russian_roulette < threshold) ret = ret + attenuation * ray_color(cx, rng, scattered, depth + 1) / threshold;
            } else {
                ret = ret + attenuation * ray_color(cx, rng, scattered, depth + 1);
            }
        }
        return ret;
    }
    return sky_value(*cx.fs, r.d, &cx.n_tex);
}

/* The same estimator unrolled: L = T_n * (emitted | sky), T_{k+1} = T_k * a_k.  This is the
 * evaluation order of the GPU wavefront kernels (DESIGN.md "Estimator"); it differs from the
 * recursion above only by fp32 rounding of the product chain. */
V3 ray_color_iterative(Ctx& cx, Rng& rng, Ray r) {
    V3 T = splat(1.0f);
    for (int depth = 0;; ++depth) {
        if (!vec3a_near_one(r.d)) {
            ++cx.n_bad;
            return splat(0.0f);
        }
        if (depth > cx.max_depth) return splat(0.0f);
        ++cx.n_rays;
        if (depth < 64) ++cx.per_depth[depth];
        rng.set_depth(depth);
        HitRecord rec;
        if (!world_hit(cx, rng, r, 1e-3f, std::numeric_limits<float>::max(), rec)) {
            return T * sky_value(*cx.fs, r.d, &cx.n_tex);
        }
        Ray scattered{splat(0.0f), splat(0.0f)};
        V3 attenuation = splat(1.0f);
        bool emissive = cx.fs->mat_type[rec.mat] == RT_MAT_EMISSION;
        if (!mat_scatter(cx, rng, rec.mat, r, rec, attenuation, scattered)) {
            if (emissive) return T * mat_emitted(cx, rec.mat, rec.uv, rec.p);
            return splat(0.0f);
        }
        T = T * attenuation;
        if (cx.russian_roulette) {
            float rr = rng.next_f32();
            float threshold = std::fmax(attenuation.x, std::fmax(attenuation.y, attenuation.z));
            if (!(rr < threshold)) return splat(0.0f);
            T = T / threshold;
        }
        r = scattered;
    }
}

} // namespace

/* ==========================================================================================
 * extern "C" surface of the oracle (ctypes / bench.py cpu_baseline / tests only)
 * ======================================================================================== */
extern "C" {

typedef struct OrcOptions {
    uint32_t rng_mode;   /* 0 = stream (reference order), 1 = counter (GPU generator) */
    uint32_t estimator;  /* 0 = recursive (main.rs:38-60), 1 = iterative (GPU order)   */
    uint32_t accel;      /* 0 = HitableList linear walk, 1 = BvhNode (demo_scene.rs:223-227) */
    uint32_t n_threads;  /* 0 = hardware_concurrency (threadpool default, main.rs:73)  */
    uint64_t bvh_seed;   /* seed of the main-thread RNG used for BVH axes (lib.rs:8: 1995) */
    uint32_t bvh_skip_perlin; /* number of Perlin::default() constructions to replay first (RNG order) */
    uint32_t seed_variant;    /* 0 = PCG32 seed expansion, 1 = SplitMix64 */
} OrcOptions;

void orc_set_seed_variant(int v) { g_seed_variant = v; }

/* replay Perlin::default() (texture.rs:60-91) on `rng`, optionally returning the tables */
static void perlin_default(Xoshiro256pp& rng, float* vec_out, uint16_t* perm_out) {
    for (int i = 0; i < 256; ++i) {
        float x = rng.next_f32(), y = rng.next_f32(), z = rng.next_f32();
        V3 v = v3(x, y, z) * (1.0f - -1.0f) + -1.0f;
        if (vec_out) vec_out[3 * i] = v.x, vec_out[3 * i + 1] = v.y, vec_out[3 * i + 2] = v.z;
    }
    std::vector<uint16_t> p(256);
    for (int i = 0; i < 256; ++i) p[(size_t)i] = (uint16_t)i;
    for (int k = 0; k < 3; ++k) {
        shuffle(rng, p);
        if (perm_out) std::memcpy(perm_out + 256 * k, p.data(), 256 * sizeof(uint16_t));
    }
}

/* world-list entries that the BVH is built over: primitives that do not merely bound a medium, then the media */
static std::vector<int> world_entries(const RtFlatScene* fs) {
    std::vector<int> objects;
    const uint32_t np = fs->n_spheres + fs->n_rects;
    for (uint32_t i = 0; i < np + fs->n_media; ++i)
        if (i >= np || prim_medium(*fs, (int)i) == RT_NO_MEDIUM) objects.push_back((int)i);
    return objects;
}

static int validate_scene(const RtFlatScene* fs) {
    if (!fs) return RT_ERR_INVALID;
    for (uint32_t i = 0; i < fs->n_spheres; ++i)
        if (fs->sph_mat[i] >= fs->n_materials) return RT_ERR_INVALID;
    for (uint32_t i = 0; i < fs->n_rects; ++i)
        if (fs->rect_mat[i] >= fs->n_materials || fs->rect_axis[i] > RT_RECT_XY) return RT_ERR_INVALID;
    if (fs->n_media > RT_MAX_MEDIA) return RT_ERR_INVALID;
    for (uint32_t i = 0; i < fs->n_media; ++i)
        if (fs->med_mat[i] >= fs->n_materials) return RT_ERR_INVALID;
    for (uint32_t i = 0; i < fs->n_xforms; ++i)
        if (fs->xf_type[i] > RT_XF_ROTATE_Y || (fs->xf_parent[i] != RT_NO_XFORM && fs->xf_parent[i] >= i)) return RT_ERR_INVALID;
    return RT_OK;
}

uint32_t orc_shard_rows(uint32_t ny, uint32_t band, uint32_t count, uint32_t id) {
    if (count <= 1) return ny;
    if (band == 0) band = 1;
    uint32_t n = 0;
    for (uint32_t j = 0; j < ny; ++j)
        if ((j / band) % count == id) ++n;
    return n;
}

/*
 * The pixel loop, main.rs:77-108.  out_rgb_f32: linear mean before gamma, local row 0 = lowest
 * image row of the shard.  out_rgb8: gamma 2, *255.99 as u8, rows flipped (main.rs:127).
 */
int orc_render(const RtFlatScene* fs, const RtCamera* cam, const RtParams* prm, const OrcOptions* opt,
               float* out_rgb_f32, uint8_t* out_rgb8, RtStats* stats) {
    if (validate_scene(fs) != RT_OK || !cam || !prm || !opt) return RT_ERR_INVALID;
    if (prm->nx == 0 || prm->ny == 0 || prm->spp == 0) return RT_ERR_INVALID;
    g_seed_variant = (int)opt->seed_variant;
    const uint32_t nx = prm->nx, ny = prm->ny, spp = prm->spp;
    const uint32_t count = prm->shard_count <= 1 ? 1 : prm->shard_count;
    const uint32_t band = prm->shard_band == 0 ? 1 : prm->shard_band;
    std::vector<uint32_t> rows;
    for (uint32_t j = 0; j < ny; ++j)
        if (count == 1 || (j / band) % count == prm->shard_id) rows.push_back(j);
    const size_t nrows = rows.size();

    Bvh bvh;
    if (opt->accel == 1 && !world_entries(fs).empty()) {
        Xoshiro256pp main_rng = smallrng_seed_from_u64(opt->bvh_seed);
        for (uint32_t k = 0; k < opt->bvh_skip_perlin; ++k) perlin_default(main_rng, nullptr, nullptr);
        std::vector<int> objects = world_entries(fs);
        bvh.root = bvh_build(*fs, bvh, objects, 0, objects.size(), main_rng);
    }

    unsigned nthreads = opt->n_threads ? opt->n_threads : std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    std::vector<float> fb(nrows * nx * 3, 0.0f);
    std::atomic<uint32_t> next_col{0};
    struct Acc {
        uint64_t rays = 0, tex = 0, bad = 0;
        uint64_t per_depth[64] = {0};
    };
    std::vector<Acc> accs(nthreads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](unsigned tid) {
        Ctx cx;
        cx.fs = fs;
        cx.bvh = (opt->accel == 1 && bvh.root >= 0) ? &bvh : nullptr;
        cx.max_depth = prm->max_depth;
        cx.russian_roulette = (prm->flags & RT_FLAG_RUSSIAN_ROULETTE) != 0;
        std::vector<Ray> rays(spp);
        for (;;) {
            uint32_t i = next_col.fetch_add(1); /* one job per column, main.rs:77 */
            if (i >= nx) break;
            Rng rng;
            rng.counter = opt->rng_mode == 1;
            rng.k0 = rng.k1 = rng.ctr = 0;
            if (!rng.counter) rng.xo = smallrng_seed_from_u64(prm->seed + (uint64_t)i); /* main.rs:81-83 */
            for (size_t lj = 0; lj < nrows; ++lj) {
                const uint32_t j = rows[lj];
                V3 c = splat(0.0f);
                if (!rng.counter) {
                    /* main.rs:86-94: all jitters first */
                    for (uint32_t s = 0; s < spp; ++s) {
                        float u = ((float)i + rng.next_f32()) / (float)nx;
                        float v = ((float)j + rng.next_f32()) / (float)ny;
                        rays[s] = camera_get_ray(*cam, u, v);
                    }
                    for (uint32_t s = 0; s < spp; ++s) { /* main.rs:95-97 */
                        V3 L = opt->estimator == 0 ? ray_color(cx, rng, rays[s], 0) : ray_color_iterative(cx, rng, rays[s]);
                        c = c + L;
                    }
                } else {
                    const uint32_t pix = j * nx + i;
                    for (uint32_t s = 0; s < spp; ++s) {
                        ctr_path_key(prm->seed, pix, s, rng.k0, rng.k1);
                        rng.ctr = 0;
                        float u = ((float)i + rng.next_f32()) / (float)nx;
                        float v = ((float)j + rng.next_f32()) / (float)ny;
                        Ray r = camera_get_ray(*cam, u, v);
                        V3 L = opt->estimator == 0 ? ray_color(cx, rng, r, 0) : ray_color_iterative(cx, rng, r);
                        c = c + L;
                    }
                }
                c = c / (float)spp; /* main.rs:98 */
                float* px = &fb[(lj * nx + i) * 3];
                px[0] = c.x, px[1] = c.y, px[2] = c.z;
            }
        }
        accs[tid].rays = cx.n_rays;
        accs[tid].tex = cx.n_tex;
        accs[tid].bad = cx.n_bad;
        std::memcpy(accs[tid].per_depth, cx.per_depth, sizeof(cx.per_depth));
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();

    if (out_rgb_f32) std::memcpy(out_rgb_f32, fb.data(), fb.size() * sizeof(float));
    if (out_rgb8) {
        for (size_t lj = 0; lj < nrows; ++lj) {
            size_t dst_row = nrows - 1 - lj; /* main.rs:127 flip */
            for (uint32_t i = 0; i < nx; ++i)
                for (int ch = 0; ch < 3; ++ch) {
                    float c = std::pow(fb[(lj * nx + i) * 3 + ch], 0.5f); /* main.rs:99 */
                    out_rgb8[(dst_row * nx + i) * 3 + ch] = sat_u8(c * 255.99f); /* main.rs:101-105 */
                }
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->n_paths = (uint64_t)nrows * nx * spp;
        for (auto& a : accs) {
            stats->n_rays += a.rays;
            stats->n_texture_fetches += a.tex;
            stats->n_bad_dir += a.bad;
            for (int d = 0; d < 64; ++d) stats->rays_per_depth[d] += a.per_depth[d];
        }
        stats->n_rays_secondary = stats->n_rays - std::min(stats->n_rays, stats->n_paths);
        stats->seconds_total = std::chrono::duration<double>(t1 - t0).count();
        stats->bytes_algorithmic = 96ull * stats->n_rays + 24ull * stats->n_paths + 12ull * stats->n_texture_fetches;
        stats->n_slices = 1;
    }
    return RT_OK;
}

/* One closest-hit + shade step over caller-given rays (mirror of rt_debug_bounce). */
int orc_debug_bounce(const RtFlatScene* fs, const RtBounceIO* io, uint32_t accel) {
    if (validate_scene(fs) != RT_OK || !io) return RT_ERR_INVALID;
    Bvh bvh;
    if (accel == 1 && !world_entries(fs).empty()) {
        Xoshiro256pp main_rng = smallrng_seed_from_u64(1995);
        std::vector<int> objects = world_entries(fs);
        bvh.root = bvh_build(*fs, bvh, objects, 0, objects.size(), main_rng);
    }
    Ctx cx;
    cx.fs = fs;
    cx.bvh = bvh.root >= 0 ? &bvh : nullptr;
    cx.max_depth = 1 << 30;
    for (uint32_t n = 0; n < io->n; ++n) {
        Ray r{ld3(io->in_o + 3 * n), ld3(io->in_d + 3 * n)};
        Rng rng;
        rng.counter = true;
        rng.k0 = io->in_key[2 * n], rng.k1 = io->in_key[2 * n + 1];
        rng.ctr = 0;
        rng.set_depth((int)io->depth);
        HitRecord rec;
        V3 rad = splat(0.0f), att = splat(1.0f);
        Ray sc{splat(0.0f), splat(0.0f)};
        bool alive = false;
        int hit = -1;
        float t = 0.0f;
        if (world_hit(cx, rng, r, 1e-3f, std::numeric_limits<float>::max(), rec)) {
            hit = rec.prim;
            t = rec.t;
            rad = mat_emitted(cx, rec.mat, rec.uv, rec.p);
            alive = mat_scatter(cx, rng, rec.mat, r, rec, att, sc);
        } else {
            rad = sky_value(*fs, r.d, nullptr);
        }
        io->out_hit[n] = hit;
        io->out_t[n] = t;
        io->out_radiance[3 * n] = rad.x, io->out_radiance[3 * n + 1] = rad.y, io->out_radiance[3 * n + 2] = rad.z;
        io->out_attenuation[3 * n] = att.x, io->out_attenuation[3 * n + 1] = att.y, io->out_attenuation[3 * n + 2] = att.z;
        io->out_o[3 * n] = sc.o.x, io->out_o[3 * n + 1] = sc.o.y, io->out_o[3 * n + 2] = sc.o.z;
        io->out_d[3 * n] = sc.d.x, io->out_d[3 * n + 1] = sc.d.y, io->out_d[3 * n + 2] = sc.d.z;
        io->out_alive[n] = alive ? 1 : 0;
    }
    return RT_OK;
}

/* ---- known-answer entry points ------------------------------------------- */
void orc_xoshiro_from_state(const uint64_t s[4], uint64_t* out, uint32_t n) {
    Xoshiro256pp r;
    for (int i = 0; i < 4; ++i) r.s[i] = s[i];
    for (uint32_t i = 0; i < n; ++i) out[i] = r.next_u64();
}
void orc_smallrng_f32(uint64_t seed, int variant, float* out, uint32_t n) {
    Xoshiro256pp r = variant == 0 ? smallrng_seed_from_u64_pcg(seed) : smallrng_seed_from_u64_splitmix(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.next_f32();
}
void orc_smallrng_state(uint64_t seed, int variant, uint64_t out[4]) {
    Xoshiro256pp r = variant == 0 ? smallrng_seed_from_u64_pcg(seed) : smallrng_seed_from_u64_splitmix(seed);
    for (int i = 0; i < 4; ++i) out[i] = r.s[i];
}
void orc_shuffle_u16(uint64_t seed, uint16_t* data, uint32_t n) {
    Xoshiro256pp r = smallrng_seed_from_u64_pcg(seed);
    std::vector<uint16_t> v(data, data + n);
    shuffle(r, v);
    std::memcpy(data, v.data(), n * sizeof(uint16_t));
}
uint64_t orc_gen_range_usize(uint64_t seed, uint64_t lo, uint64_t hi, uint32_t skip) {
    Xoshiro256pp r = smallrng_seed_from_u64_pcg(seed);
    uint64_t v = 0;
    for (uint32_t i = 0; i <= skip; ++i) v = gen_range_u64(r, lo, hi);
    return v;
}
void orc_ctr_path_key(uint64_t seed, uint32_t pix, uint32_t samp, uint32_t out[2]) {
    ctr_path_key(seed, pix, samp, out[0], out[1]);
}
uint32_t orc_ctr_draw(uint32_t k0, uint32_t k1, uint32_t ctr) { return ctr_draw(k0, k1, ctr); }
void orc_sphere_get_uv(const float n[3], float uv[2]) {
    V2 r = sphere_get_uv(ld3(n));
    uv[0] = r.x, uv[1] = r.y;
}
void orc_camera_new(const float from[3], const float at[3], const float vup[3], float vfov, float aspect,
                    RtCamera* out) {
    *out = camera_new(ld3(from), ld3(at), ld3(vup), vfov, aspect);
}
void orc_camera_get_ray(const RtCamera* cam, float u, float v, float o[3], float d[3]) {
    Ray r = camera_get_ray(*cam, u, v);
    o[0] = r.o.x, o[1] = r.o.y, o[2] = r.o.z, d[0] = r.d.x, d[1] = r.d.y, d[2] = r.d.z;
}
/* Sphere::hit on a single sphere: returns 1/0 and fills t,p,n,front,tang,uv (14 floats) */
int orc_sphere_hit(const float c[3], float rad, const float o[3], const float d[3], float t_min, float t_max,
                   float out[14]) {
    RtFlatScene fs;
    std::memset(&fs, 0, sizeof(fs));
    uint32_t mat = 0;
    fs.n_spheres = 1, fs.sph_cx = &c[0], fs.sph_cy = &c[1], fs.sph_cz = &c[2], fs.sph_r = &rad, fs.sph_mat = &mat;
    HitRecord rec;
    Ray r{ld3(o), ld3(d)};
    if (!sphere_hit(fs, 0, r, t_min, t_max, rec)) return 0;
    out[0] = rec.t;
    out[1] = rec.p.x, out[2] = rec.p.y, out[3] = rec.p.z;
    out[4] = rec.norm.x, out[5] = rec.norm.y, out[6] = rec.norm.z;
    out[7] = rec.front_face ? 1.0f : 0.0f;
    out[8] = rec.tang.x, out[9] = rec.tang.y, out[10] = rec.tang.z;
    out[11] = rec.uv.x, out[12] = rec.uv.y;
    out[13] = 0.0f;
    return 1;
}
/* XYRect/XZRect/YZRect::hit on a single rectangle: returns 1/0, fills t,p,n,front,uv (10 floats) */
int orc_rect_hit(int axis, const float mn[3], const float mx[3], const float o[3], const float d[3], float t_min, float t_max,
                 float out[10]) {
    RtFlatScene fs;
    std::memset(&fs, 0, sizeof(fs));
    uint8_t ax = (uint8_t)axis;
    uint32_t mat = 0;
    fs.n_rects = 1, fs.rect_axis = &ax, fs.rect_min = mn, fs.rect_max = mx, fs.rect_mat = &mat;
    HitRecord rec;
    Ray r{ld3(o), ld3(d)};
    if (!rect_hit(fs, 0, r, t_min, t_max, rec)) return 0;
    out[0] = rec.t;
    out[1] = rec.p.x, out[2] = rec.p.y, out[3] = rec.p.z;
    out[4] = rec.norm.x, out[5] = rec.norm.y, out[6] = rec.norm.z;
    out[7] = rec.front_face ? 1.0f : 0.0f;
    out[8] = rec.uv.x, out[9] = rec.uv.y;
    return 1;
}
void orc_offset_hit_point(const float p[3], const float n[3], float out[3]) {
    V3 r = offset_hit_point(ld3(p), ld3(n));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
float orc_reflectance(float cosine, float ref_idx) { return reflectance(cosine, ref_idx); }
void orc_reflect(const float v[3], const float n[3], float out[3]) {
    V3 r = reflect(ld3(v), ld3(n));
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_refract(const float v[3], const float n[3], float eta, float out[3]) {
    V3 r = refract(ld3(v), ld3(n), eta);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_sky_gradient(const float d[3], float out[3]) {
    RtFlatScene fs;
    std::memset(&fs, 0, sizeof(fs));
    fs.sky_type = RT_SKY_GRADIENT;
    V3 r = sky_value(fs, ld3(d), nullptr);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
void orc_texture_value(const RtFlatScene* fs, uint32_t tex, const float uv[2], const float p[3], float out[3]) {
    V3 r = texture_value(*fs, tex, V2{uv[0], uv[1]}, ld3(p), nullptr);
    out[0] = r.x, out[1] = r.y, out[2] = r.z;
}
int orc_aabb_hit(const float mn[3], const float mx[3], const float o[3], const float d[3], float t_min,
                 float t_max) {
    AABB b{ld3(mn), ld3(mx)};
    Ray r{ld3(o), ld3(d)};
    return aabb_hit(b, r, t_min, t_max) ? 1 : 0;
}
/* Perlin::default() tables from the main-thread RNG (lib.rs:8 seed 1995): texture.rs:60-91 */
void orc_perlin_tables(uint64_t seed, uint32_t n_sets, float* vec_out, uint16_t* perm_out) {
    Xoshiro256pp r = smallrng_seed_from_u64(seed);
    for (uint32_t k = 0; k < n_sets; ++k)
        perlin_default(r, vec_out + (size_t)k * 768, perm_out + (size_t)k * 768);
}

/*
 * demo_scene.rs:150-221 — the thread-RNG draws of final_scene in source order: PerlinTex::new(0.1) (:164),
 * 1000 x vec3a_random_range(0., 165.) (:176-178), one gen_range(0..3) per node of BvhNode::new over the 1000
 * spheres (:179, hitable.rs:177-221: the node itself, then its left half, then its right half), then the 400
 * box heights gen::<f32>() * 100. + 1. (:192).  Independent restatement of what the host mirror must produce.
 */
static void final_scene_bvh_axes(Xoshiro256pp& r, size_t span) {
    (void)gen_range_u64(r, 0, 3);
    if (span > 2) {
        final_scene_bvh_axes(r, span / 2);
        final_scene_bvh_axes(r, span - span / 2);
    }
}
/* Hitable::bbox of world entry `idx` of a flat scene (sphere, rect or medium, through its wrapper chain):
 * hitable.rs:104-108, 274-278, 420-431, 449-474, 581-583.  out = {min.xyz, max.xyz}. */
void orc_entry_bbox(const RtFlatScene* fs, uint32_t idx, float out[6]) {
    const AABB b = sphere_bbox(*fs, (int)idx);
    out[0] = b.mn.x, out[1] = b.mn.y, out[2] = b.mn.z, out[3] = b.mx.x, out[4] = b.mx.y, out[5] = b.mx.z;
}
void orc_final_scene_layout(uint64_t seed, float* centres /* [3000] */, float* heights /* [400] */) {
    Xoshiro256pp rng = smallrng_seed_from_u64(seed);
    std::vector<float> vec(768);
    std::vector<uint16_t> perm(768);
    perlin_default(rng, vec.data(), perm.data());
    for (int i = 0; i < 1000; ++i) {
        float x = rng.next_f32(), y = rng.next_f32(), z = rng.next_f32();
        V3 c = V3{x, y, z} * (165.0f - 0.0f) + 0.0f;
        centres[3 * i] = c.x, centres[3 * i + 1] = c.y, centres[3 * i + 2] = c.z;
    }
    final_scene_bvh_axes(rng, 1000);
    for (int i = 0; i < 400; ++i) heights[i] = rng.next_f32() * 100.0f + 1.0f;
}

/*
 * demo_scene.rs:56-77 — the random-spheres layout drawn from SmallRng::seed_from_u64(95).
 * Emits, for each of the 529 small spheres: center xyz, kind (0 diffuse, 1 metal, 2 glass),
 * colour rgb and fuzz (8 floats per sphere).  The oracle's independent restatement of what the
 * host-side `sphere_scene` mirror must produce.
 */
uint32_t orc_sphere_scene_layout(uint64_t seed, float* out /* [529*8] */) {
    Xoshiro256pp rng = smallrng_seed_from_u64(seed);
    uint32_t n = 0;
    for (int a = -11; a <= 11; ++a)
        for (int b = -11; b <= 11; ++b) {
            float choose_mat = rng.next_f32();
            float cx = (float)a + 0.9f * rng.next_f32();
            float cz = (float)b + 0.9f * rng.next_f32();
            float kind, cr = 0, cg = 0, cb = 0, fuzz = 0;
            if (choose_mat < 0.8f) {
                float ax = rng.next_f32(), ay = rng.next_f32(), az = rng.next_f32();
                float bx = rng.next_f32(), by = rng.next_f32(), bz = rng.next_f32();
                cr = ax * bx, cg = ay * by, cb = az * bz;
                kind = 0;
            } else if (choose_mat < 0.95f) {
                float ax = rng.next_f32(), ay = rng.next_f32(), az = rng.next_f32();
                cr = ax * 0.5f + 0.5f, cg = ay * 0.5f + 0.5f, cb = az * 0.5f + 0.5f;
                fuzz = rng.next_f32();
                kind = 1;
            } else {
                kind = 2;
            }
            float* o = out + 8 * n;
            o[0] = cx, o[1] = 0.2f, o[2] = cz, o[3] = kind, o[4] = cr, o[5] = cg, o[6] = cb, o[7] = fuzz;
            ++n;
        }
    return n;
}

uint32_t orc_bvh_stats(const RtFlatScene* fs, uint64_t seed, uint32_t skip_perlin, uint32_t* n_nodes) {
    Bvh bvh;
    Xoshiro256pp main_rng = smallrng_seed_from_u64(seed);
    for (uint32_t k = 0; k < skip_perlin; ++k) perlin_default(main_rng, nullptr, nullptr);
    std::vector<int> objects = world_entries(fs);
    bvh.root = bvh_build(*fs, bvh, objects, 0, objects.size(), main_rng);
    if (n_nodes) *n_nodes = (uint32_t)bvh.nodes.size();
    return (uint32_t)bvh.root;
}

} /* extern "C" */
