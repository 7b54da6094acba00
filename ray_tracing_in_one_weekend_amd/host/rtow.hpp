// rtow.hpp — host-side C++ mirror of the reference's scene-construction API (the drop-in
// surface).  Same names, argument meaning and defaults as the Rust types:
//
//   Camera::new_(lookfrom, lookat, vup, vfov_deg, aspect)      camera.rs:14-39
//   Sphere{c, r, mat, name}                                    hitable.rs:57-62
//   XYRect/XZRect/YZRect{min, max, mat}, GBox::new_(min,max,mat)  hitable.rs:244-402
//   Translate{offset, ptr}, RotateY::new_(ptr, angle)          hitable.rs:404-520
//   ConstantMedium::new_(boundary, density, phase_tex)         hitable.rs:523-588
//   HitableList = vector<shared_ptr<Hitable>>                  hitable.rs:114
//   BvhNode::new_(objects, start, end)                         hitable.rs:177-221
//   Emission{emit} Diffuse{albedo} Lambert{albedo} Metal{albedo,fuzz} Dielectric{ior}
//   Isotropic{albedo}                                          material.rs
//   OrenNayar BurleyDiffuse RoughPlastic DisneyDiffuse DisneyMetal DisneySheen DisneyClearcoat  pbr.rs
//   ConstantTex{col} CheckerTex::new_(odd,even) PerlinTex::new_(scale) ImageTex::new_(path)  texture.rs
//   SKY_COLOR (lib.rs:11), RNG (lib.rs:7-9), ENV_TEX (demo_scene.rs:19)
//
// The one addition a Rust host would also need (trait objects give no introspection,
// SURVEY.md §8(b)): every Hitable/Material/Texture has `flatten(FlatSceneBuilder&)`, which
// emits the structure-of-arrays RtFlatScene that crosses the C-ABI (include/rtow_mi355x.h).
// The GPU replaces BVH traversal by its own closest-hit search, so BvhNode flattens to the
// spheres it holds.
//
// Rust `new` is spelled `new_` (C++ keyword).  This header is product code: it must not
// include anything from oracle/.
#pragma once
#include "../../include/rtow_mi355x.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <set>
#include <utility>
#include <vector>

namespace rtow {

// ---- glam::Vec3A (the subset the constructors use) ------------------------------------------
struct Vec3A {
    float x = 0, y = 0, z = 0;
    static Vec3A ZERO() { return Vec3A{0, 0, 0}; }
    static Vec3A ONE() { return Vec3A{1, 1, 1}; }
    Vec3A operator+(Vec3A b) const { return {x + b.x, y + b.y, z + b.z}; }
    Vec3A operator-(Vec3A b) const { return {x - b.x, y - b.y, z - b.z}; }
    Vec3A operator*(Vec3A b) const { return {x * b.x, y * b.y, z * b.z}; }
    Vec3A operator*(float s) const { return {x * s, y * s, z * s}; }
    Vec3A operator/(float s) const { return {x / s, y / s, z / s}; }
    Vec3A operator+(float s) const { return {x + s, y + s, z + s}; }
    float dot(Vec3A b) const { return (x * b.x + y * b.y) + z * b.z; }
    float length() const { return std::sqrt(dot(*this)); }
    Vec3A normalize() const { return *this * (1.0f / length()); }
    Vec3A cross(Vec3A b) const { return {y * b.z - z * b.y, z * b.x - x * b.z, x * b.y - y * b.x}; }
};
inline Vec3A operator*(float s, Vec3A v) { return {s * v.x, s * v.y, s * v.z}; }
inline Vec3A vec3a(float x, float y, float z) { return Vec3A{x, y, z}; }

// ---- rand 0.8.5 SmallRng (xoshiro256++, rand_core PCG32 seed expansion) -----------------------
class SmallRng {
  public:
    static SmallRng seed_from_u64(uint64_t state) {
        SmallRng r;
        uint32_t w[8];
        for (int i = 0; i < 8; ++i) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            w[i] = (xs >> rot) | (xs << ((32u - rot) & 31u));
        }
        for (int i = 0; i < 4; ++i) r.s_[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        if (!(r.s_[0] | r.s_[1] | r.s_[2] | r.s_[3])) { // zero seed -> SplitMix64(0), as xoshiro's from_seed does
            uint64_t z = 0;
            for (int i = 0; i < 4; ++i) {
                z += 0x9e3779b97f4a7c15ull;
                uint64_t v = z;
                v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ull;
                v = (v ^ (v >> 27)) * 0x94d049bb133111ebull;
                r.s_[i] = v ^ (v >> 31);
            }
        }
        return r;
    }
    uint64_t next_u64() {
        uint64_t result = rotl(s_[0] + s_[3], 23) + s_[0];
        uint64_t t = s_[1] << 17;
        s_[2] ^= s_[0], s_[3] ^= s_[1], s_[1] ^= s_[2], s_[0] ^= s_[3], s_[2] ^= t;
        s_[3] = rotl(s_[3], 45);
        return result;
    }
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
    float gen_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); } // rng.gen::<f32>()
    uint32_t gen_range_u32(uint32_t high_excl) {                               // gen_range(0..high) for u32
        uint32_t zone = (high_excl << __builtin_clz(high_excl)) - 1u;
        for (;;) {
            uint64_t m = (uint64_t)next_u32() * high_excl;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }
    uint64_t gen_range_usize(uint64_t high_excl) { // gen_range(0..high) for usize
        uint64_t zone = (high_excl << __builtin_clzll(high_excl)) - 1ull;
        for (;;) {
            unsigned __int128 m = (unsigned __int128)next_u64() * high_excl;
            if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
        }
    }
    template <class T>
    void shuffle(std::vector<T>& v) { // SliceRandom::shuffle
        for (size_t i = v.size() - 1; i >= 1; --i) std::swap(v[i], v[gen_range_u32((uint32_t)(i + 1))]);
    }

  private:
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t s_[4] = {0, 0, 0, 0};
};

// lib.rs:7-9 — thread-local RNG seeded 1995 (Perlin tables, BVH axes draw from it)
inline SmallRng& RNG() {
    thread_local SmallRng rng = SmallRng::seed_from_u64(1995);
    return rng;
}
inline void RNG_reseed(uint64_t seed) { RNG() = SmallRng::seed_from_u64(seed); }

// ---- flat-scene builder -------------------------------------------------------------------------
class Texture;
class Material;
class ImageTex;

class FlatSceneBuilder {
  public:
    // interned by object identity so that shared Arc<dyn _> stay shared
    uint32_t intern_texture(const Texture* t);
    uint32_t intern_material(const Material* m);
    uint32_t add_image(const ImageTex* img);
    uint32_t add_perlin(const float* vec768, const uint16_t* perm768) {
        perlin_vec.insert(perlin_vec.end(), vec768, vec768 + 768);
        perlin_perm.insert(perlin_perm.end(), perm768, perm768 + 768);
        return (uint32_t)(perlin_vec.size() / 768 - 1);
    }
    uint32_t push_texture(uint8_t type, Vec3A c0, Vec3A c1, float scale, uint32_t aux) {
        tex_type.push_back(type);
        push3(tex_color0, c0), push3(tex_color1, c1);
        tex_scale.push_back(scale), tex_aux.push_back(aux);
        return (uint32_t)tex_type.size() - 1;
    }
    uint32_t push_material(uint8_t type, Vec3A color, float p0, float p1, float p2, float p3, uint32_t t0, uint32_t t1) {
        mat_type.push_back(type);
        push3(mat_color, color);
        mat_p0.push_back(p0), mat_p1.push_back(p1), mat_p2.push_back(p2), mat_p3.push_back(p3);
        mat_tex0.push_back(t0), mat_tex1.push_back(t1);
        return (uint32_t)mat_type.size() - 1;
    }
    void push_rect(uint8_t axis, Vec3A mn, Vec3A mx, uint32_t mat) {
        rect_axis.push_back(axis);
        push3(rect_min, mn), push3(rect_max, mx);
        rect_mat.push_back(mat);
        rect_xform.push_back(cur_xform);
        rect_medium.push_back(cur_medium);
    }
    void push_sphere(Vec3A c, float r, uint32_t mat, const std::string& name) {
        sph_cx.push_back(c.x), sph_cy.push_back(c.y), sph_cz.push_back(c.z), sph_r.push_back(r), sph_mat.push_back(mat);
        sph_name.push_back(name);
        sph_xform.push_back(cur_xform);
        sph_medium.push_back(cur_medium);
    }
    // hitable.rs:523-533 ConstantMedium: primitives pushed until end_medium() bound medium `m`
    uint32_t begin_medium(float neg_inv_density, uint32_t mat) {
        if (cur_medium != RT_NO_MEDIUM) throw std::runtime_error("flatten: a ConstantMedium inside a ConstantMedium boundary");
        if (med_mat.size() >= RT_MAX_MEDIA) throw std::runtime_error("flatten: more than RT_MAX_MEDIA media");
        med_neg_inv_density.push_back(neg_inv_density);
        med_mat.push_back(mat);
        med_xform.push_back(cur_xform); // the wrappers open at this point are AROUND the medium (hitable.rs:409-416, 479-509)
        cur_medium = (uint32_t)med_mat.size() - 1;
        return cur_medium;
    }
    void end_medium() { cur_medium = RT_NO_MEDIUM; }
    // Opens an instance wrapper (hitable.rs:404-520): primitives pushed until the matching
    // pop_xform() lie below it.  Returns the previous innermost wrapper (to restore).
    uint32_t push_xform(uint8_t type, float p0, float p1, float p2) {
        xf_type.push_back(type);
        xf_param.push_back(p0), xf_param.push_back(p1), xf_param.push_back(p2), xf_param.push_back(0.0f);
        xf_parent.push_back(cur_xform);
        const uint32_t prev = cur_xform;
        cur_xform = (uint32_t)xf_type.size() - 1;
        return prev;
    }
    void pop_xform(uint32_t prev) { cur_xform = prev; }
    RtFlatScene view() const {
        RtFlatScene s;
        std::memset(&s, 0, sizeof(s));
        s.n_spheres = (uint32_t)sph_r.size();
        s.sph_cx = sph_cx.data(), s.sph_cy = sph_cy.data(), s.sph_cz = sph_cz.data(), s.sph_r = sph_r.data();
        s.sph_mat = sph_mat.data();
        s.n_rects = (uint32_t)rect_axis.size();
        s.rect_axis = rect_axis.data(), s.rect_min = rect_min.data(), s.rect_max = rect_max.data(), s.rect_mat = rect_mat.data();
        s.n_xforms = (uint32_t)xf_type.size();
        s.xf_type = xf_type.data(), s.xf_param = xf_param.data(), s.xf_parent = xf_parent.data();
        s.sph_xform = sph_xform.data(), s.rect_xform = rect_xform.data();
        s.n_media = (uint32_t)med_mat.size();
        s.med_neg_inv_density = med_neg_inv_density.data(), s.med_mat = med_mat.data();
        s.sph_medium = sph_medium.data(), s.rect_medium = rect_medium.data(), s.med_xform = med_xform.data();
        s.n_materials = (uint32_t)mat_type.size();
        s.mat_type = mat_type.data(), s.mat_color = mat_color.data();
        s.mat_p0 = mat_p0.data(), s.mat_p1 = mat_p1.data(), s.mat_p2 = mat_p2.data(), s.mat_p3 = mat_p3.data();
        s.mat_tex0 = mat_tex0.data(), s.mat_tex1 = mat_tex1.data();
        s.n_textures = (uint32_t)tex_type.size();
        s.tex_type = tex_type.data(), s.tex_color0 = tex_color0.data(), s.tex_color1 = tex_color1.data();
        s.tex_scale = tex_scale.data(), s.tex_aux = tex_aux.data();
        s.n_perlin = (uint32_t)(perlin_vec.size() / 768);
        s.perlin_vec = perlin_vec.data(), s.perlin_perm = perlin_perm.data();
        s.n_images = (uint32_t)img_w.size();
        s.img_w = img_w.data(), s.img_h = img_h.data(), s.img_offset = img_offset.data();
        s.texels = texels.data(), s.n_texel_floats = texels.size();
        s.sky_type = sky_type, s.sky_image = sky_image;
        return s;
    }

    std::vector<float> sph_cx, sph_cy, sph_cz, sph_r;
    std::vector<uint32_t> sph_mat;
    std::vector<std::string> sph_name;
    std::vector<uint8_t> rect_axis;
    std::vector<float> rect_min, rect_max;
    std::vector<uint32_t> rect_mat;
    std::vector<uint8_t> xf_type;
    std::vector<float> xf_param;
    std::vector<uint32_t> xf_parent, sph_xform, rect_xform;
    uint32_t cur_xform = RT_NO_XFORM;
    std::vector<float> med_neg_inv_density;
    std::vector<uint32_t> med_mat, sph_medium, rect_medium, med_xform;
    uint32_t cur_medium = RT_NO_MEDIUM;
    uint32_t visit_mult = 1; // how many times BvhNode::hit calls the object being flattened per visit (see ConstantMedium)
    std::vector<uint8_t> mat_type;
    std::vector<float> mat_color, mat_p0, mat_p1, mat_p2, mat_p3;
    std::vector<uint32_t> mat_tex0, mat_tex1;
    std::vector<uint8_t> tex_type;
    std::vector<float> tex_color0, tex_color1, tex_scale;
    std::vector<uint32_t> tex_aux;
    std::vector<float> perlin_vec;
    std::vector<uint16_t> perlin_perm;
    std::vector<uint32_t> img_w, img_h;
    std::vector<uint64_t> img_offset;
    std::vector<float> texels;
    uint32_t sky_type = RT_SKY_GRADIENT, sky_image = 0;

  private:
    static void push3(std::vector<float>& v, Vec3A c) { v.push_back(c.x), v.push_back(c.y), v.push_back(c.z); }
    std::unordered_map<const void*, uint32_t> tex_ids_, mat_ids_, img_ids_;
};

// ---- texture.rs ------------------------------------------------------------------------------------
class Texture {
  public:
    virtual ~Texture() = default;
    virtual uint32_t flatten(FlatSceneBuilder& b) const = 0; // returns the texture index
};
using TexturePtr = std::shared_ptr<const Texture>;

struct ConstantTex : Texture { // texture.rs:15-17
    Vec3A col;
    explicit ConstantTex(Vec3A c) : col(c) {}
    uint32_t flatten(FlatSceneBuilder& b) const override { return b.push_texture(RT_TEX_CONSTANT, col, Vec3A{}, 0.0f, 0); }
};
class CheckerTex : public Texture { // texture.rs:25-37; odd/even are ConstantTex by construction
  public:
    static std::shared_ptr<CheckerTex> new_(Vec3A odd_col, Vec3A even_col) {
        return std::shared_ptr<CheckerTex>(new CheckerTex(odd_col, even_col));
    }
    uint32_t flatten(FlatSceneBuilder& b) const override { return b.push_texture(RT_TEX_CHECKER, odd_, even_, 0.0f, 0); }

  private:
    CheckerTex(Vec3A o, Vec3A e) : odd_(o), even_(e) {}
    Vec3A odd_, even_;
};
class PerlinTex : public Texture { // texture.rs:53-91,153-161: tables drawn from the thread RNG
  public:
    static std::shared_ptr<PerlinTex> new_(float scale) { return std::shared_ptr<PerlinTex>(new PerlinTex(scale)); }
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_texture(RT_TEX_PERLIN, Vec3A{}, Vec3A{}, scale_, b.add_perlin(vec_, perm_));
    }
    const float* rand_vec() const { return vec_; }
    const uint16_t* perm() const { return perm_; }

  private:
    explicit PerlinTex(float scale) : scale_(scale) {
        SmallRng& rng = RNG();
        for (int i = 0; i < 256; ++i) { // vec3a_random_range(-1., 1.)
            float x = rng.gen_f32(), y = rng.gen_f32(), z = rng.gen_f32();
            Vec3A v = vec3a(x, y, z) * (1.0f - -1.0f) + -1.0f;
            vec_[3 * i] = v.x, vec_[3 * i + 1] = v.y, vec_[3 * i + 2] = v.z;
        }
        std::vector<uint16_t> p(256);
        for (int i = 0; i < 256; ++i) p[(size_t)i] = (uint16_t)i;
        for (int k = 0; k < 3; ++k) { // perm_x, perm_y, perm_z: successive shuffles of the same vector
            rng.shuffle(p);
            std::memcpy(perm_ + 256 * k, p.data(), 256 * sizeof(uint16_t));
        }
    }
    float scale_;
    float vec_[768];
    uint16_t perm_[768];
};

// Decoded images registered by path (JPEG decode happens in the embedding host; see
// INTEGRATION.md).  ImageTex::new_(path) resolves against this registry first, then falls
// back to binary PPM (P6) files.
struct DecodedImage {
    uint32_t w = 0, h = 0;
    std::vector<float> rgb; // Rgb<f32> = u8 / 255 (texture.rs:177 to_rgb32f)
};
inline std::map<std::string, std::shared_ptr<const DecodedImage>>& image_registry() {
    static std::map<std::string, std::shared_ptr<const DecodedImage>> reg;
    return reg;
}
inline void register_image(const std::string& path, uint32_t w, uint32_t h, const float* rgb) {
    auto img = std::make_shared<DecodedImage>();
    img->w = w, img->h = h;
    img->rgb.assign(rgb, rgb + (size_t)w * h * 3);
    image_registry()[path] = img;
}
std::shared_ptr<const DecodedImage> load_ppm(const std::string& path); // demo_scene.cpp

class ImageTex : public Texture { // texture.rs:170-181
  public:
    static std::shared_ptr<ImageTex> new_(const std::string& path) {
        auto it = image_registry().find(path);
        std::shared_ptr<const DecodedImage> img = it != image_registry().end() ? it->second : load_ppm(path);
        if (!img) throw std::runtime_error("ImageTex::new: cannot open " + path);
        return std::shared_ptr<ImageTex>(new ImageTex(img));
    }
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_texture(RT_TEX_IMAGE, Vec3A{}, Vec3A{}, 0.0f, b.add_image(this));
    }
    const DecodedImage& img() const { return *img_; }

  private:
    explicit ImageTex(std::shared_ptr<const DecodedImage> i) : img_(std::move(i)) {}
    std::shared_ptr<const DecodedImage> img_;
};

// ---- material.rs / pbr.rs ------------------------------------------------------------------------
class Material {
  public:
    virtual ~Material() = default;
    virtual uint32_t flatten(FlatSceneBuilder& b) const = 0; // returns the material index
};
using MaterialPtr = std::shared_ptr<const Material>;

#define RTOW_TEX_MATERIAL(Name, TAG, FIELD)                                                          \
    struct Name : Material {                                                                         \
        TexturePtr FIELD;                                                                            \
        explicit Name(TexturePtr t) : FIELD(std::move(t)) {}                                         \
        uint32_t flatten(FlatSceneBuilder& b) const override {                                       \
            return b.push_material(TAG, Vec3A{}, 0, 0, 0, 0, b.intern_texture(FIELD.get()), RT_NO_TEX); \
        }                                                                                            \
    }
RTOW_TEX_MATERIAL(Emission, RT_MAT_EMISSION, emit);    // material.rs:17-19
RTOW_TEX_MATERIAL(Diffuse, RT_MAT_DIFFUSE, albedo);    // material.rs:31-33
RTOW_TEX_MATERIAL(Lambert, RT_MAT_LAMBERT, albedo);    // material.rs:48-50
RTOW_TEX_MATERIAL(Isotropic, RT_MAT_ISOTROPIC, albedo);// material.rs:99-101
#undef RTOW_TEX_MATERIAL

struct Metal : Material { // material.rs:61-64
    Vec3A albedo;
    float fuzz;
    Metal(Vec3A a, float f) : albedo(a), fuzz(f) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_METAL, albedo, fuzz, 0, 0, 0, RT_NO_TEX, RT_NO_TEX);
    }
};
struct Dielectric : Material { // material.rs:75-77
    float ior;
    explicit Dielectric(float i) : ior(i) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_DIELECTRIC, Vec3A{}, ior, 0, 0, 0, RT_NO_TEX, RT_NO_TEX);
    }
};
struct OrenNayar : Material { // pbr.rs:12-15
    TexturePtr albedo;
    float roughness;
    OrenNayar(TexturePtr a, float r) : albedo(std::move(a)), roughness(r) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_OREN_NAYAR, Vec3A{}, roughness, 0, 0, 0, b.intern_texture(albedo.get()), RT_NO_TEX);
    }
};
struct BurleyDiffuse : Material { // pbr.rs:45-48
    TexturePtr albedo;
    float roughness;
    BurleyDiffuse(TexturePtr a, float r) : albedo(std::move(a)), roughness(r) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_BURLEY_DIFFUSE, Vec3A{}, roughness, 0, 0, 0, b.intern_texture(albedo.get()), RT_NO_TEX);
    }
};
struct RoughPlastic : Material { // pbr.rs:153-158
    TexturePtr spec_color, diff_color;
    float roughness, eta;
    RoughPlastic(TexturePtr s, TexturePtr d, float r, float e)
        : spec_color(std::move(s)), diff_color(std::move(d)), roughness(r), eta(e) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_ROUGH_PLASTIC, Vec3A{}, roughness, eta, 0, 0, b.intern_texture(spec_color.get()),
                               b.intern_texture(diff_color.get()));
    }
};
struct DisneyDiffuse : Material { // pbr.rs:192-196
    TexturePtr albedo;
    float roughness, subsurface;
    DisneyDiffuse(TexturePtr a, float r, float s) : albedo(std::move(a)), roughness(r), subsurface(s) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_DISNEY_DIFFUSE, Vec3A{}, roughness, subsurface, 0, 0, b.intern_texture(albedo.get()),
                               RT_NO_TEX);
    }
};
struct DisneyMetal : Material { // pbr.rs:224-229
    TexturePtr albedo;
    float roughness, anisotropic, rot;
    DisneyMetal(TexturePtr a, float r, float an, float ro) : albedo(std::move(a)), roughness(r), anisotropic(an), rot(ro) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_DISNEY_METAL, Vec3A{}, roughness, anisotropic, rot, 0, b.intern_texture(albedo.get()),
                               RT_NO_TEX);
    }
};
struct DisneySheen : Material { // pbr.rs:281-284
    TexturePtr albedo;
    float tint;
    DisneySheen(TexturePtr a, float t) : albedo(std::move(a)), tint(t) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_DISNEY_SHEEN, Vec3A{}, tint, 0, 0, 0, b.intern_texture(albedo.get()), RT_NO_TEX);
    }
};
struct DisneyClearcoat : Material { // pbr.rs:310-312
    float clearcoat_gloss;
    explicit DisneyClearcoat(float g) : clearcoat_gloss(g) {}
    uint32_t flatten(FlatSceneBuilder& b) const override {
        return b.push_material(RT_MAT_DISNEY_CLEARCOAT, Vec3A{}, clearcoat_gloss, 0, 0, 0, RT_NO_TEX, RT_NO_TEX);
    }
};

inline uint32_t FlatSceneBuilder::intern_texture(const Texture* t) {
    if (!t) throw std::runtime_error("flatten: null texture");
    auto it = tex_ids_.find(t);
    if (it != tex_ids_.end()) return it->second;
    uint32_t id = t->flatten(*this);
    tex_ids_[t] = id;
    return id;
}
inline uint32_t FlatSceneBuilder::intern_material(const Material* m) {
    if (!m) throw std::runtime_error("flatten: null material");
    auto it = mat_ids_.find(m);
    if (it != mat_ids_.end()) return it->second;
    uint32_t id = m->flatten(*this);
    mat_ids_[m] = id;
    return id;
}
inline uint32_t FlatSceneBuilder::add_image(const ImageTex* t) {
    const DecodedImage* key = &t->img();
    auto it = img_ids_.find(key);
    if (it != img_ids_.end()) return it->second;
    uint32_t id = (uint32_t)img_w.size();
    img_w.push_back(key->w), img_h.push_back(key->h), img_offset.push_back(texels.size());
    texels.insert(texels.end(), key->rgb.begin(), key->rgb.end());
    img_ids_[key] = id;
    return id;
}

// ---- hitable.rs --------------------------------------------------------------------------------------
struct AABB { // math.rs:90-94 (#[derive(Default)]: zeros)
    Vec3A min, max;
    AABB surround(AABB rhs) const { // math.rs:115-130; f32::min/max
        return AABB{vec3a(std::fmin(min.x, rhs.min.x), std::fmin(min.y, rhs.min.y), std::fmin(min.z, rhs.min.z)),
                    vec3a(std::fmax(max.x, rhs.max.x), std::fmax(max.y, rhs.max.y), std::fmax(max.z, rhs.max.z))};
    }
};

class Hitable {
  public:
    virtual ~Hitable() = default;
    virtual void flatten(FlatSceneBuilder& b) const = 0;
    virtual bool bbox(AABB& aabb) const = 0; // hitable.rs:52 (BvhNode::new sorts by it)
    virtual std::string memo() const = 0;    // hitable.rs:53
};
using HitablePtr = std::shared_ptr<const Hitable>;
using HitableList = std::vector<HitablePtr>; // hitable.rs:114

struct Sphere : Hitable { // hitable.rs:57-62
    Vec3A c;
    float r;
    MaterialPtr mat;
    std::string name;
    Sphere(Vec3A c_, float r_, MaterialPtr m, std::string n) : c(c_), r(r_), mat(std::move(m)), name(std::move(n)) {}
    void flatten(FlatSceneBuilder& b) const override { b.push_sphere(c, r, b.intern_material(mat.get()), name); }
    bool bbox(AABB& aabb) const override { // hitable.rs:104-108
        aabb.min = c + (-r), aabb.max = c + r;
        return true;
    }
    std::string memo() const override { return name; }
};

// hitable.rs:244-362 — axis-aligned rectangles.  The plane coordinate is min[axis] (max[axis] is ignored by hit()).
#define RTOW_RECT(Name, AXIS, MEMO)                                                                      \
    struct Name : Hitable {                                                                              \
        Vec3A min, max;                                                                                  \
        MaterialPtr mat;                                                                                 \
        Name(Vec3A mn, Vec3A mx, MaterialPtr m) : min(mn), max(mx), mat(std::move(m)) {}                 \
        void flatten(FlatSceneBuilder& b) const override { b.push_rect(AXIS, min, max, b.intern_material(mat.get())); } \
        bool bbox(AABB& aabb) const override { /* hitable.rs:274-278, 314-318, 354-358 */                 \
            const int k = AXIS == RT_RECT_XY ? 2 : AXIS == RT_RECT_XZ ? 1 : 0;                             \
            float mn[3] = {min.x, min.y, min.z}, mx[3] = {max.x, max.y, max.z};                           \
            mn[k] = mn[k] - 0.0001f, mx[k] = mx[k] + 0.0001f;                                             \
            aabb.min = vec3a(mn[0], mn[1], mn[2]), aabb.max = vec3a(mx[0], mx[1], mx[2]);                 \
            return true;                                                                                 \
        }                                                                                                \
        std::string memo() const override { return MEMO; }                                               \
    }
RTOW_RECT(XYRect, RT_RECT_XY, "XYRect"); // hitable.rs:244-282
RTOW_RECT(XZRect, RT_RECT_XZ, "XZRect"); // hitable.rs:284-322
RTOW_RECT(YZRect, RT_RECT_YZ, "YZRect"); // hitable.rs:324-362
#undef RTOW_RECT

// hitable.rs:364-402 — six rectangles in the order of GBox::new; hit() is the list walk over them.
class GBox : public Hitable {
  public:
    static std::shared_ptr<GBox> new_(Vec3A min, Vec3A max, MaterialPtr mat) {
        auto g = std::shared_ptr<GBox>(new GBox());
        g->aabb_ = AABB{min, max};
        g->sides_ = {
            std::make_shared<XYRect>(vec3a(min.x, min.y, min.z), vec3a(max.x, max.y, min.z), mat),
            std::make_shared<XYRect>(vec3a(min.x, min.y, max.z), vec3a(max.x, max.y, max.z), mat),
            std::make_shared<XZRect>(vec3a(min.x, min.y, min.z), vec3a(max.x, min.y, max.z), mat),
            std::make_shared<XZRect>(vec3a(min.x, max.y, min.z), vec3a(max.x, max.y, max.z), mat),
            std::make_shared<YZRect>(vec3a(min.x, min.y, min.z), vec3a(min.x, max.y, max.z), mat),
            std::make_shared<YZRect>(vec3a(max.x, min.y, min.z), vec3a(max.x, max.y, max.z), mat),
        };
        return g;
    }
    void flatten(FlatSceneBuilder& b) const override {
        for (auto& s : sides_) s->flatten(b);
    }
    bool bbox(AABB& aabb) const override { // hitable.rs:394-397
        aabb = aabb_;
        return true;
    }
    std::string memo() const override { throw std::runtime_error("GBox::memo: todo!() in the reference (hitable.rs:399-401)"); }

  private:
    AABB aabb_;
    HitableList sides_;
};

// hitable.rs:404-436 — Translate { offset, ptr }
struct Translate : Hitable {
    Vec3A offset;
    HitablePtr ptr;
    Translate(Vec3A o, HitablePtr p) : offset(o), ptr(std::move(p)) {}
    void flatten(FlatSceneBuilder& b) const override {
        const uint32_t prev = b.push_xform(RT_XF_TRANSLATE, offset.x, offset.y, offset.z);
        ptr->flatten(b);
        b.pop_xform(prev);
    }
    bool bbox(AABB& aabb) const override { // hitable.rs:420-431
        AABB got;
        if (!ptr->bbox(got)) return false;
        aabb = AABB{got.min + offset, got.max + offset};
        return true;
    }
    std::string memo() const override { throw std::runtime_error("Translate::memo: todo!() in the reference (hitable.rs:433-435)"); }
};

// hitable.rs:438-520 — RotateY::new(ptr, angle_degrees); sin/cos exactly as radians.sin_cos() of f32::to_radians
class RotateY : public Hitable {
  public:
    static std::shared_ptr<RotateY> new_(HitablePtr ptr, float angle) {
        auto r = std::shared_ptr<RotateY>(new RotateY());
        r->ptr_ = std::move(ptr);
        r->angle_ = angle;
        const float radians = angle * (3.14159265358979323846f / 180.0f);
        r->sin_theta_ = std::sin(radians);
        r->cos_theta_ = std::cos(radians);
        AABB in; // hitable.rs:449-474: bounds of the 8 rotated corners
        r->has_box_ = r->ptr_->bbox(in);
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    const float x = i == 0 ? in.min.x : in.max.x;
                    const float y = j == 0 ? in.min.y : in.max.y;
                    const float z = k == 0 ? in.min.z : in.max.z;
                    const float tester[3] = {r->cos_theta_ * x + r->sin_theta_ * z, y, -r->sin_theta_ * x + r->cos_theta_ * z};
                    for (int c = 0; c < 3; ++c) mn[c] = std::fmin(mn[c], tester[c]), mx[c] = std::fmax(mx[c], tester[c]);
                }
        r->aabb_ = AABB{vec3a(mn[0], mn[1], mn[2]), vec3a(mx[0], mx[1], mx[2])};
        return r;
    }
    bool bbox(AABB& aabb) const override { // hitable.rs:511-514
        aabb = aabb_;
        return has_box_;
    }
    void flatten(FlatSceneBuilder& b) const override {
        const uint32_t prev = b.push_xform(RT_XF_ROTATE_Y, sin_theta_, cos_theta_, angle_);
        ptr_->flatten(b);
        b.pop_xform(prev);
    }
    std::string memo() const override { throw std::runtime_error("RotateY::memo: todo!() in the reference (hitable.rs:517-519)"); }

  private:
    HitablePtr ptr_;
    float angle_ = 0, sin_theta_ = 0, cos_theta_ = 1;
    bool has_box_ = false;
    AABB aabb_;
};

// hitable.rs:523-588 — ConstantMedium::new(boundary, density, phase_fn_texture); the phase function is
// Isotropic { albedo } (material.rs:99-113)
class ConstantMedium : public Hitable {
  public:
    static std::shared_ptr<ConstantMedium> new_(HitablePtr boundary, float density, TexturePtr phase_fn) {
        auto m = std::shared_ptr<ConstantMedium>(new ConstantMedium());
        m->boundary_ = std::move(boundary);
        m->phase_fn_ = std::make_shared<Isotropic>(std::move(phase_fn));
        m->neg_inv_density_ = -1.0f / density;
        return m;
    }
    // A BvhNode over a single object holds it as both children and hit() calls both (hitable.rs:188, 236-237).
    // That is idempotent for a surface, but a medium draws a fresh free path on each call and the nearer of the
    // two wins: the medium behaves as one of b.visit_mult times the density.  The flat scene carries that rate.
    void flatten(FlatSceneBuilder& b) const override {
        b.begin_medium(neg_inv_density_ / (float)b.visit_mult, b.intern_material(phase_fn_.get()));
        boundary_->flatten(b);
        b.end_medium();
    }
    bool bbox(AABB& aabb) const override { return boundary_->bbox(aabb); } // hitable.rs:581-583
    std::string memo() const override { throw std::runtime_error("ConstantMedium::memo: todo!() in the reference (hitable.rs:585-587)"); }

  private:
    HitablePtr boundary_;
    MaterialPtr phase_fn_;
    float neg_inv_density_ = 0;
};

// hitable.rs:158-221.  The accelerated path does its own closest-hit search, so flatten() emits the primitives
// of [start, end) in construction order; the constructor still does everything the reference's does — one
// gen_range(0..3) per node, the stable sort of objects[start..end) by box_compare, the surrounding box — because
// the shape of the tree decides which objects sit alone in a node and are therefore hit twice per visit.
class BvhNode : public Hitable {
  public:
    static std::shared_ptr<BvhNode> new_(HitableList& objects, size_t start, size_t end) {
        if (end <= start || end > objects.size()) throw std::runtime_error("BvhNode::new: empty span (unimplemented!() in the reference)");
        auto n = std::shared_ptr<BvhNode>(new BvhNode());
        n->items_.assign(objects.begin() + (long)start, objects.begin() + (long)end);
        n->aabb_ = n->build(objects, start, end);
        return n;
    }
    void flatten(FlatSceneBuilder& b) const override {
        for (auto& h : items_) {
            const uint32_t saved = b.visit_mult;
            if (twice_.count(h.get())) b.visit_mult *= 2u;
            h->flatten(b);
            b.visit_mult = saved;
        }
    }
    bool bbox(AABB& aabb) const override { // hitable.rs:228-231
        aabb = aabb_;
        return true;
    }
    std::string memo() const override { return "BvhNode"; }
    // objects that ended up alone in a node (left == right, hitable.rs:188)
    size_t n_visited_twice() const { return twice_.size(); }

  private:
    static int32_t total_key(float f) { // f32::total_cmp
        int32_t b;
        std::memcpy(&b, &f, 4);
        b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
        return b;
    }
    static float axis_min(const HitablePtr& h, size_t axis) { // box_compare, hitable.rs:163-174
        AABB box;
        (void)h->bbox(box); // "No bounding box in bvh_node constructor." is only a message in the reference
        return axis == 0 ? box.min.x : axis == 1 ? box.min.y : box.min.z;
    }
    AABB build(HitableList& objects, size_t start, size_t end) { // hitable.rs:177-221
        const size_t axis = (size_t)RNG().gen_range_usize(3);
        const size_t span = end - start;
        AABB box_a, box_b;
        if (span == 1) {
            twice_.insert(objects[start].get());
            (void)objects[start]->bbox(box_a);
            box_b = box_a;
        } else {
            std::stable_sort(objects.begin() + (long)start, objects.begin() + (long)end, [&](const HitablePtr& a, const HitablePtr& b) {
                return total_key(axis_min(a, axis)) < total_key(axis_min(b, axis));
            });
            if (span == 2) {
                (void)objects[start]->bbox(box_a);
                (void)objects[start + 1]->bbox(box_b);
            } else {
                const size_t mid = start + span / 2;
                box_a = build(objects, start, mid);
                box_b = build(objects, mid, end);
            }
        }
        return box_a.surround(box_b);
    }
    HitableList items_;
    std::set<const Hitable*> twice_;
    AABB aabb_;
};

// ---- lib.rs:11 SKY_COLOR, demo_scene.rs:19 ENV_TEX ------------------------------------------------------
enum class SkyFn { sky_color, black_sky, tex_sky_color }; // demo_scene.rs:22-35
struct SkyState {
    SkyFn fn = SkyFn::sky_color;
    bool set = false;
    std::shared_ptr<const ImageTex> env_tex; // ENV_TEX
};
inline SkyState& SKY_COLOR() {
    static SkyState s;
    return s;
}
// OnceCell::set panics when called twice; a library cannot, so the last call wins.
inline void SKY_COLOR_set(SkyFn fn) { SKY_COLOR().fn = fn, SKY_COLOR().set = true; }
inline void ENV_TEX_set(std::shared_ptr<const ImageTex> t) { SKY_COLOR().env_tex = std::move(t); }

// ---- camera.rs --------------------------------------------------------------------------------------------
class Camera {
  public:
    static Camera new_(Vec3A lookfrom, Vec3A lookat, Vec3A vup, float vfov, float aspect_ratio) { // camera.rs:14-39
        Camera c;
        c.origin = lookfrom;
        const float RADS_PER_DEG = 3.14159265358979323846f / 180.0f;
        float theta = vfov * RADS_PER_DEG;
        float viewport_height = std::tan(theta / 2.0f) * 2.0f;
        float viewport_width = viewport_height * aspect_ratio;
        Vec3A w = (lookfrom - lookat).normalize();
        Vec3A u = vup.cross(w).normalize();
        Vec3A v = w.cross(u);
        c.horizontal = viewport_width * u;
        c.vertical = viewport_height * v;
        c.lower_left_corner = c.origin - c.horizontal / 2.0f - c.vertical / 2.0f - w;
        return c;
    }
    // the accessor the private fields need (SURVEY.md §8(b) "Obstacle to flattening")
    RtCamera flatten() const {
        RtCamera r;
        const Vec3A* src[4] = {&origin, &horizontal, &vertical, &lower_left_corner};
        float* dst[4] = {r.origin, r.horizontal, r.vertical, r.lower_left_corner};
        for (int k = 0; k < 4; ++k) dst[k][0] = src[k]->x, dst[k][1] = src[k]->y, dst[k][2] = src[k]->z;
        return r;
    }

  private:
    Vec3A origin, horizontal, vertical, lower_left_corner;
};

// Flattens a world + the global sky state into `b`.
inline void flatten_world(const HitableList& world, FlatSceneBuilder& b) {
    for (auto& h : world) h->flatten(b);
    const SkyState& sky = SKY_COLOR();
    switch (sky.fn) {
    case SkyFn::sky_color: b.sky_type = RT_SKY_GRADIENT; break;
    case SkyFn::black_sky: b.sky_type = RT_SKY_BLACK; break;
    case SkyFn::tex_sky_color:
        if (!sky.env_tex) throw std::runtime_error("tex_sky_color: ENV_TEX not set (ENV_TEX.get().unwrap() panics in the reference)");
        b.sky_type = RT_SKY_ENV;
        b.sky_image = b.add_image(sky.env_tex.get());
        break;
    }
}

// ---- demo_scene.rs ---------------------------------------------------------------------------------------------
using SceneFn = std::pair<HitableList, Camera> (*)(float aspect_ratio);
std::pair<HitableList, Camera> sphere_scene(float aspect_ratio); // demo_scene.rs:37-86  "random-spheres"
std::pair<HitableList, Camera> test_sphere(float aspect_ratio);  // demo_scene.rs:229-244
std::pair<HitableList, Camera> simple_light_scene(float aspect_ratio); // demo_scene.rs:88-110 (spheres + XYRect light)
std::pair<HitableList, Camera> cornell_box(float aspect_ratio);        // demo_scene.rs:112-148 (walls + two smoke boxes)
std::pair<HitableList, Camera> final_scene(float aspect_ratio);        // demo_scene.rs:150-221
// Build-authored scenes from reference constructors (BASELINE.json configs 4 and 5; the
// reference ships no scene for them).
std::pair<HitableList, Camera> earth_env_scene(float aspect_ratio);
std::pair<HitableList, Camera> pbr_sweep_scene(float aspect_ratio);

} // namespace rtow
