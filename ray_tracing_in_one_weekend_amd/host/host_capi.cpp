// host_capi.cpp — C handle API (include/rtow_host.h) over the C++ mirror in rtow.hpp.
#include "../../include/rtow_host.h"
#include "rtow.hpp"

#include <exception>

using namespace rtow;

struct RthScene {
    std::vector<TexturePtr> textures;
    std::vector<MaterialPtr> materials;
    HitableList world;
    bool has_camera = false;
    Camera camera;
    RtCamera rt_camera{};
    bool finished = false;
    FlatSceneBuilder builder;
    RtFlatScene flat{};
};

namespace {
thread_local std::string g_err;
template <class F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
    } catch (...) {
        g_err = "unknown C++ exception";
    }
    return RT_ERR_INVALID;
}
template <class F>
uint32_t guarded_handle(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
    } catch (...) {
        g_err = "unknown C++ exception";
    }
    return RTH_INVALID;
}
int finish(RthScene* s, bool use_bvh) {
    if (!s->has_camera) throw std::runtime_error("scene has no camera");
    HitableList world = s->world;
    if (use_bvh && !world.empty()) {
        HitableList wrapped;
        wrapped.push_back(BvhNode::new_(world, 0, world.size()));
        world = wrapped;
    }
    s->builder = FlatSceneBuilder();
    flatten_world(world, s->builder);
    s->flat = s->builder.view();
    s->rt_camera = s->camera.flatten();
    s->finished = true;
    return RT_OK;
}
Vec3A v(const float* p) { return vec3a(p[0], p[1], p[2]); }
} // namespace

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

int rth_register_image(const char* path, uint32_t w, uint32_t h, const float* rgb) {
    return guarded([&] {
        if (!path || !rgb || !w || !h) throw std::runtime_error("rth_register_image: bad argument");
        register_image(path, w, h, rgb);
        return RT_OK;
    });
}

void rth_rng_reseed(uint64_t seed) { RNG_reseed(seed); }

int rth_scene_build(const char* name, float aspect_ratio, RthScene** out) {
    return guarded([&] {
        if (!name || !out) throw std::runtime_error("rth_scene_build: bad argument");
        *out = nullptr;
        std::string n(name);
        SceneFn fn = nullptr;
        if (n == "sphere_scene") fn = sphere_scene;
        else if (n == "test_sphere") fn = test_sphere;
        else if (n == "simple_light_scene") fn = simple_light_scene;
        else if (n == "cornell_box") fn = cornell_box;
        else if (n == "final_scene") fn = final_scene;
        else if (n == "earth_env_scene") fn = earth_env_scene;
        else if (n == "pbr_sweep_scene") fn = pbr_sweep_scene;
        else throw std::runtime_error("rth_scene_build: unknown scene '" + n + "'");
        // the reference builds one scene per process, so the thread RNG starts from lib.rs:8's seed
        RNG_reseed(1995);
        auto wc = fn(aspect_ratio);
        std::unique_ptr<RthScene> s(new RthScene());
        s->world = wc.first;
        s->camera = wc.second;
        s->has_camera = true;
        finish(s.get(), false); // the scene fn already wrapped the world (build_bvh) where the reference does
        *out = s.release();
        return RT_OK;
    });
}

int rth_scene_new(RthScene** out) {
    return guarded([&] {
        if (!out) throw std::runtime_error("rth_scene_new: out is NULL");
        *out = new RthScene();
        RNG_reseed(1995); // fresh-process RNG state (lib.rs:8); rth_rng_reseed() overrides
        SKY_COLOR_set(SkyFn::sky_color);
        return RT_OK;
    });
}

static uint32_t add_tex(RthScene* s, TexturePtr t) {
    s->textures.push_back(std::move(t));
    return (uint32_t)s->textures.size() - 1;
}
static TexturePtr get_tex(RthScene* s, uint32_t h, const char* what) {
    if (h >= s->textures.size()) throw std::runtime_error(std::string("material needs texture: ") + what);
    return s->textures[h];
}

uint32_t rth_tex_constant(RthScene* s, const float col[3]) {
    return guarded_handle([&] { return add_tex(s, std::make_shared<ConstantTex>(v(col))); });
}
uint32_t rth_tex_checker(RthScene* s, const float odd[3], const float even[3]) {
    return guarded_handle([&] { return add_tex(s, CheckerTex::new_(v(odd), v(even))); });
}
uint32_t rth_tex_perlin(RthScene* s, float scale) {
    return guarded_handle([&] { return add_tex(s, PerlinTex::new_(scale)); });
}
uint32_t rth_tex_image(RthScene* s, const char* path) {
    return guarded_handle([&] { return add_tex(s, ImageTex::new_(path ? path : "")); });
}

uint32_t rth_material(RthScene* s, uint32_t type, uint32_t tex0, uint32_t tex1, const float color[3], const float p[4]) {
    return guarded_handle([&]() -> uint32_t {
        MaterialPtr m;
        const float z4[4] = {0, 0, 0, 0}, z3[3] = {0, 0, 0};
        if (!p) p = z4;
        if (!color) color = z3;
        switch (type) {
        case RT_MAT_EMISSION: m = std::make_shared<Emission>(get_tex(s, tex0, "Emission.emit")); break;
        case RT_MAT_DIFFUSE: m = std::make_shared<Diffuse>(get_tex(s, tex0, "Diffuse.albedo")); break;
        case RT_MAT_LAMBERT: m = std::make_shared<Lambert>(get_tex(s, tex0, "Lambert.albedo")); break;
        case RT_MAT_METAL: m = std::make_shared<Metal>(v(color), p[0]); break;
        case RT_MAT_DIELECTRIC: m = std::make_shared<Dielectric>(p[0]); break;
        case RT_MAT_ISOTROPIC: m = std::make_shared<Isotropic>(get_tex(s, tex0, "Isotropic.albedo")); break;
        case RT_MAT_OREN_NAYAR: m = std::make_shared<OrenNayar>(get_tex(s, tex0, "OrenNayar.albedo"), p[0]); break;
        case RT_MAT_BURLEY_DIFFUSE: m = std::make_shared<BurleyDiffuse>(get_tex(s, tex0, "BurleyDiffuse.albedo"), p[0]); break;
        case RT_MAT_ROUGH_PLASTIC:
            m = std::make_shared<RoughPlastic>(get_tex(s, tex0, "RoughPlastic.spec_color"), get_tex(s, tex1, "RoughPlastic.diff_color"),
                                               p[0], p[1]);
            break;
        case RT_MAT_DISNEY_DIFFUSE: m = std::make_shared<DisneyDiffuse>(get_tex(s, tex0, "DisneyDiffuse.albedo"), p[0], p[1]); break;
        case RT_MAT_DISNEY_METAL: m = std::make_shared<DisneyMetal>(get_tex(s, tex0, "DisneyMetal.albedo"), p[0], p[1], p[2]); break;
        case RT_MAT_DISNEY_SHEEN: m = std::make_shared<DisneySheen>(get_tex(s, tex0, "DisneySheen.albedo"), p[0]); break;
        case RT_MAT_DISNEY_CLEARCOAT: m = std::make_shared<DisneyClearcoat>(p[0]); break;
        default: throw std::runtime_error("rth_material: unknown material type");
        }
        s->materials.push_back(m);
        return (uint32_t)s->materials.size() - 1;
    });
}

uint32_t rth_sphere(RthScene* s, const float c[3], float r, uint32_t material, const char* name) {
    return guarded_handle([&]() -> uint32_t {
        if (material >= s->materials.size()) throw std::runtime_error("rth_sphere: bad material handle");
        s->world.push_back(std::make_shared<Sphere>(v(c), r, s->materials[material], name ? name : ""));
        return (uint32_t)s->world.size() - 1;
    });
}

uint32_t rth_rect(RthScene* s, uint32_t axis, const float mn[3], const float mx[3], uint32_t material) {
    return guarded_handle([&]() -> uint32_t {
        if (material >= s->materials.size()) throw std::runtime_error("rth_rect: bad material handle");
        HitablePtr r;
        switch (axis) {
        case RT_RECT_XY: r = std::make_shared<XYRect>(v(mn), v(mx), s->materials[material]); break;
        case RT_RECT_XZ: r = std::make_shared<XZRect>(v(mn), v(mx), s->materials[material]); break;
        case RT_RECT_YZ: r = std::make_shared<YZRect>(v(mn), v(mx), s->materials[material]); break;
        default: throw std::runtime_error("rth_rect: unknown axis");
        }
        s->world.push_back(r);
        return (uint32_t)s->world.size() - 1;
    });
}

uint32_t rth_gbox(RthScene* s, const float mn[3], const float mx[3], uint32_t material) {
    return guarded_handle([&]() -> uint32_t {
        if (material >= s->materials.size()) throw std::runtime_error("rth_gbox: bad material handle");
        s->world.push_back(GBox::new_(v(mn), v(mx), s->materials[material]));
        return (uint32_t)s->world.size() - 1;
    });
}

// Replaces world entry `hitable` (a handle returned by rth_sphere/rth_rect/rth_gbox/rth_translate/rth_rotate_y)
// by the wrapper around it, like `let box_1 = Arc::new(RotateY::new(box_1, 15.))` (demo_scene.rs:121-122).
uint32_t rth_translate(RthScene* s, uint32_t hitable, const float offset[3]) {
    return guarded_handle([&]() -> uint32_t {
        if (hitable >= s->world.size()) throw std::runtime_error("rth_translate: bad hitable handle");
        s->world[hitable] = std::make_shared<Translate>(v(offset), s->world[hitable]);
        return hitable;
    });
}
// Hitable::bbox of world entry `hitable` (hitable.rs:52): {min.xyz, max.xyz}.  Returns 1/0 like the trait method, < 0 on error.
int rth_hitable_bbox(RthScene* s, uint32_t hitable, float out[6]) {
    int has = 0;
    const int rc = guarded([&] {
        if (!out || hitable >= s->world.size()) throw std::runtime_error("rth_hitable_bbox: bad argument");
        AABB b;
        has = s->world[hitable]->bbox(b) ? 1 : 0;
        out[0] = b.min.x, out[1] = b.min.y, out[2] = b.min.z, out[3] = b.max.x, out[4] = b.max.y, out[5] = b.max.z;
        return RT_OK;
    });
    return rc != RT_OK ? -1 : has;
}
uint32_t rth_rotate_y(RthScene* s, uint32_t hitable, float angle_degrees) {
    return guarded_handle([&]() -> uint32_t {
        if (hitable >= s->world.size()) throw std::runtime_error("rth_rotate_y: bad hitable handle");
        s->world[hitable] = RotateY::new_(s->world[hitable], angle_degrees);
        return hitable;
    });
}

// Replaces world entry `hitable` by ConstantMedium::new(hitable, density, phase texture) (hitable.rs:529-533).
uint32_t rth_constant_medium(RthScene* s, uint32_t hitable, float density, uint32_t phase_tex) {
    return guarded_handle([&]() -> uint32_t {
        if (hitable >= s->world.size()) throw std::runtime_error("rth_constant_medium: bad hitable handle");
        s->world[hitable] = ConstantMedium::new_(s->world[hitable], density, get_tex(s, phase_tex, "ConstantMedium.phase_fn"));
        return hitable;
    });
}

int rth_set_sky(RthScene* s, uint32_t sky, const char* env_path) {
    return guarded([&] {
        (void)s;
        switch (sky) {
        case RT_SKY_GRADIENT: SKY_COLOR_set(SkyFn::sky_color); break;
        case RT_SKY_BLACK: SKY_COLOR_set(SkyFn::black_sky); break;
        case RT_SKY_ENV:
            ENV_TEX_set(ImageTex::new_(env_path ? env_path : ""));
            SKY_COLOR_set(SkyFn::tex_sky_color);
            break;
        default: throw std::runtime_error("rth_set_sky: unknown sky");
        }
        return RT_OK;
    });
}

int rth_set_camera(RthScene* s, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                   float aspect_ratio) {
    return guarded([&] {
        s->camera = Camera::new_(v(lookfrom), v(lookat), v(vup), vfov, aspect_ratio);
        s->has_camera = true;
        return RT_OK;
    });
}

int rth_scene_finish(RthScene* s, int use_bvh) {
    return guarded([&] { return finish(s, use_bvh != 0); });
}

const RtFlatScene* rth_scene_flat(const RthScene* s) { return (s && s->finished) ? &s->flat : nullptr; }

int rth_scene_camera(const RthScene* s, RtCamera* out) {
    if (!s || !out || !s->finished) return RT_ERR_INVALID;
    *out = s->rt_camera;
    return RT_OK;
}

const char* rth_scene_sphere_name(const RthScene* s, uint32_t index) {
    if (!s || !s->finished || index >= s->builder.sph_name.size()) return "";
    return s->builder.sph_name[index].c_str();
}

void rth_scene_free(RthScene* s) { delete s; }

} // extern "C"
