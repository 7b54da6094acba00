// demo_scene.cpp — host-side mirror of the reference scene builders that feed the accelerated
// path (demo_scene.rs:37-86 sphere_scene, demo_scene.rs:229-244 test_sphere) plus the two
// build-authored scenes BASELINE.json configs 4 and 5 call for.  The other three reference
// scenes: simple_light_scene (spheres + an XYRect) and cornell_box (rectangles, RotateY/Translate boxes that bound
// two ConstantMedium) and final_scene (1000-sphere and 400-box sub-BVHs, an instanced sphere cloud, a
// sphere-bounded medium) are mirrored below, so all five scenes of demo_scene.rs build here.
#include "rtow.hpp"

#include <cstdio>

namespace rtow {

std::shared_ptr<const DecodedImage> load_ppm(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    auto img = std::make_shared<DecodedImage>();
    char magic[3] = {0, 0, 0};
    int w = 0, h = 0, maxv = 0;
    auto next_int = [&](int& v) {
        int c = std::fgetc(f);
        for (;;) {
            while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = std::fgetc(f);
            if (c == '#') {
                while (c != '\n' && c != EOF) c = std::fgetc(f);
                continue;
            }
            break;
        }
        if (c < '0' || c > '9') return false;
        v = 0;
        while (c >= '0' && c <= '9') v = v * 10 + (c - '0'), c = std::fgetc(f);
        return true; // the single whitespace after the token has been consumed
    };
    bool ok = std::fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '6' && next_int(w) && next_int(h) &&
              next_int(maxv) && w > 0 && h > 0 && maxv == 255;
    if (ok) {
        std::vector<unsigned char> raw((size_t)w * h * 3);
        ok = std::fread(raw.data(), 1, raw.size(), f) == raw.size();
        if (ok) {
            img->w = (uint32_t)w, img->h = (uint32_t)h;
            img->rgb.resize(raw.size());
            for (size_t i = 0; i < raw.size(); ++i) img->rgb[i] = (float)raw[i] / 255.0f; // to_rgb32f
        }
    }
    std::fclose(f);
    return ok ? img : nullptr;
}

static std::pair<HitableList, Camera> build_bvh(HitableList& world, Camera cam) { // demo_scene.rs:223-227
    HitableList out;
    out.push_back(BvhNode::new_(world, 0, world.size()));
    return {out, cam};
}

// demo_scene.rs:37-86 — "random-spheres": 4 fixed spheres + 23x23 small ones, no rejection
// around the big spheres (unlike the book), 533 spheres in total.
std::pair<HitableList, Camera> sphere_scene(float aspect_ratio) {
    SKY_COLOR_set(SkyFn::sky_color); // :38
    auto perlin = PerlinTex::new_(4.0f);                   // :41 (first consumer of the thread RNG)
    auto earth_map = ImageTex::new_("res/earthmap.jpg");   // :42
    auto material_ground = std::make_shared<Diffuse>(perlin);
    auto material_1 = std::make_shared<Emission>(earth_map);
    auto material_2 = std::make_shared<Dielectric>(1.5f);
    auto material_3 = std::make_shared<Metal>(vec3a(0.8f, 0.6f, 0.2f), 0.0f);
    HitableList world = {
        std::make_shared<Sphere>(vec3a(1.0f, -1000.0f, -1.0f), 1000.0f, material_ground, "Ground"),
        std::make_shared<Sphere>(vec3a(0.0f, 1.0f, 3.0f), 1.0f, material_1, "Sphere_1"),
        std::make_shared<Sphere>(vec3a(-4.0f, 1.0f, 0.0f), 1.0f, material_2, "Sphere_2"),
        std::make_shared<Sphere>(vec3a(4.0f, 1.0f, 0.0f), 1.0f, material_3, "Sphere_3"),
    };
    SmallRng rng = SmallRng::seed_from_u64(95); // :56
    for (int a = -11; a <= 11; ++a) {
        for (int b = -11; b <= 11; ++b) {
            float choose_mat = rng.gen_f32();
            float cx = (float)a + 0.9f * rng.gen_f32();
            float cz = (float)b + 0.9f * rng.gen_f32();
            Vec3A center = vec3a(cx, 0.2f, cz);
            MaterialPtr mat;
            if (choose_mat < 0.8f) {
                float ax = rng.gen_f32(), ay = rng.gen_f32(), az = rng.gen_f32();
                float bx = rng.gen_f32(), by = rng.gen_f32(), bz = rng.gen_f32();
                Vec3A col = vec3a(ax, ay, az) * vec3a(bx, by, bz);
                mat = std::make_shared<Diffuse>(std::make_shared<ConstantTex>(col));
            } else if (choose_mat < 0.95f) {
                float ax = rng.gen_f32(), ay = rng.gen_f32(), az = rng.gen_f32();
                Vec3A albedo = vec3a(ax, ay, az) * 0.5f + 0.5f;
                float fuzz = rng.gen_f32();
                mat = std::make_shared<Metal>(albedo, fuzz);
            } else {
                mat = material_2;
            }
            world.push_back(std::make_shared<Sphere>(center, 0.2f, mat, "Sphere " + std::to_string(a) + ", " + std::to_string(b)));
        }
    }
    Camera cam = Camera::new_(vec3a(13.0f, 2.0f, 3.0f), vec3a(0.0f, 0.0f, 0.0f), vec3a(0.0f, 1.0f, 0.0f), 20.0f, aspect_ratio);
    return build_bvh(world, cam); // :85
}

// demo_scene.rs:229-244 — two Lambert spheres, gradient sky, no BVH.
std::pair<HitableList, Camera> test_sphere(float aspect_ratio) {
    SKY_COLOR_set(SkyFn::sky_color);
    auto ground = std::make_shared<Lambert>(std::make_shared<ConstantTex>(vec3a(0.5f, 0.5f, 0.5f)));
    HitableList world = {
        std::make_shared<Sphere>(vec3a(0.0f, -100.5f, -1.0f), 100.0f, ground, "Ground"),
        std::make_shared<Sphere>(vec3a(0.0f, 0.0f, -1.0f), 0.5f, ground, "Test"),
    };
    Camera cam = Camera::new_(vec3a(0.0f, 0.0f, 0.0f), vec3a(0.0f, 0.0f, -1.0f), vec3a(0.0f, 1.0f, 0.0f), 90.0f, aspect_ratio);
    return {world, cam};
}

// demo_scene.rs:88-110 — Perlin ground and sphere lit by an emissive sphere and an emissive XYRect, black sky.
std::pair<HitableList, Camera> simple_light_scene(float aspect_ratio) {
    SKY_COLOR_set(SkyFn::black_sky); // :89
    auto perlin = PerlinTex::new_(4.0f);
    auto mat_perlin = std::make_shared<Diffuse>(perlin);
    auto material_1 = std::make_shared<Emission>(std::make_shared<ConstantTex>(vec3a(4.0f, 4.0f, 4.0f)));
    HitableList world = {
        std::make_shared<Sphere>(vec3a(1.0f, -1000.0f, -1.0f), 1000.0f, mat_perlin, "Ground"),
        std::make_shared<Sphere>(vec3a(0.0f, 2.0f, 0.0f), 2.0f, mat_perlin, "Sphere_1"),
        std::make_shared<Sphere>(vec3a(0.0f, 6.5f, 0.0f), 2.0f, material_1, "Sphere_2"),
        std::make_shared<XYRect>(vec3a(3.0f, 1.0f, -2.0f), vec3a(5.0f, 3.0f, -2.0f), material_1),
    };
    Camera cam = Camera::new_(vec3a(26.0f, 3.0f, 6.0f), vec3a(0.0f, 0.0f, 0.0f), vec3a(0.0f, 1.0f, 0.0f), 20.0f, aspect_ratio);
    return build_bvh(world, cam); // :109
}

// demo_scene.rs:112-148 — Cornell box; the two boxes only bound the black and the white smoke.
std::pair<HitableList, Camera> cornell_box(float aspect_ratio) {
    SKY_COLOR_set(SkyFn::black_sky); // :113
    auto red = std::make_shared<Diffuse>(std::make_shared<ConstantTex>(vec3a(0.65f, 0.05f, 0.05f)));
    auto white = std::make_shared<Diffuse>(std::make_shared<ConstantTex>(vec3a(0.73f, 0.73f, 0.73f)));
    auto green = std::make_shared<Diffuse>(std::make_shared<ConstantTex>(vec3a(0.12f, 0.45f, 0.15f)));
    auto light = std::make_shared<Emission>(std::make_shared<ConstantTex>(vec3a(7.0f, 7.0f, 7.0f)));
    HitablePtr box_1 = GBox::new_(Vec3A::ZERO(), vec3a(165.0f, 330.0f, 165.0f), white);
    box_1 = RotateY::new_(box_1, 15.0f);
    box_1 = std::make_shared<Translate>(vec3a(265.0f, 0.0f, 295.0f), box_1);
    auto mediun_1 = ConstantMedium::new_(box_1, 0.01f, std::make_shared<ConstantTex>(Vec3A::ZERO()));
    HitablePtr box_2 = GBox::new_(Vec3A::ZERO(), vec3a(165.0f, 165.0f, 165.0f), white);
    box_2 = RotateY::new_(box_2, -18.0f);
    box_2 = std::make_shared<Translate>(vec3a(130.0f, 0.0f, 65.0f), box_2);
    auto mediun_2 = ConstantMedium::new_(box_2, 0.01f, std::make_shared<ConstantTex>(Vec3A::ONE()));
    HitableList world = {
        std::make_shared<XZRect>(vec3a(113.0f, 554.0f, 127.0f), vec3a(443.0f, 554.0f, 432.0f), light),
        std::make_shared<XYRect>(vec3a(0.0f, 0.0f, 555.0f), vec3a(555.0f, 555.0f, 555.0f), white),
        std::make_shared<XZRect>(vec3a(0.0f, 0.0f, 0.0f), vec3a(555.0f, 0.0f, 555.0f), white),
        std::make_shared<XZRect>(vec3a(0.0f, 555.0f, 0.0f), vec3a(555.0f, 555.0f, 555.0f), white),
        std::make_shared<YZRect>(vec3a(0.0f, 0.0f, 0.0f), vec3a(0.0f, 555.0f, 555.0f), red),
        std::make_shared<YZRect>(vec3a(555.0f, 0.0f, 0.0f), vec3a(555.0f, 555.0f, 555.0f), green),
        mediun_1,
        mediun_2,
    };
    Camera cam = Camera::new_(vec3a(278.0f, 278.0f, -800.0f), vec3a(278.0f, 278.0f, 0.0f), vec3a(0.0f, 1.0f, 0.0f), 40.0f, aspect_ratio);
    return build_bvh(world, cam); // :147
}

// demo_scene.rs:150-221 — "the next week" final scene.  Thread-RNG consumption order is the reference's:
// PerlinTex::new (:164), 1000 sphere centres (:176-178), the sphere BvhNode axes (:179), 400 box heights
// (:184-198), the box BvhNode axes (:200), the world BvhNode axes (:220).
std::pair<HitableList, Camera> final_scene(float aspect_ratio) {
    SKY_COLOR_set(SkyFn::black_sky); // :151
    auto ground = std::make_shared<Diffuse>(std::make_shared<ConstantTex>(vec3a(0.48f, 0.83f, 0.53f)));
    auto white = std::make_shared<Diffuse>(std::make_shared<ConstantTex>(vec3a(0.73f, 0.73f, 0.73f)));
    auto brown = std::make_shared<BurleyDiffuse>(std::make_shared<ConstantTex>(vec3a(0.7f, 0.3f, 0.1f)), 0.9f);
    auto light = std::make_shared<Emission>(std::make_shared<ConstantTex>(vec3a(7.0f, 7.0f, 7.0f)));
    auto dielectric = std::make_shared<Dielectric>(1.5f);
    auto metal = std::make_shared<Metal>(vec3a(0.8f, 0.8f, 0.9f), 1.0f);

    auto earth_map = ImageTex::new_("res/earthmap.jpg");
    auto earth = std::make_shared<Sphere>(vec3a(400.0f, 200.0f, 400.0f), 100.0f, std::make_shared<Diffuse>(earth_map), "EarthSphere");
    auto perlin = PerlinTex::new_(0.1f);
    auto perlin_sphere = std::make_shared<Sphere>(vec3a(220.0f, 280.0f, 300.0f), 80.0f, std::make_shared<Diffuse>(perlin), "PerlinSphere");
    auto brown_sphere = std::make_shared<Sphere>(vec3a(400.0f, 400.0f, 200.0f), 50.0f, brown, "BrownSphere");
    auto dielectric_sphere = std::make_shared<Sphere>(vec3a(260.0f, 150.0f, 45.0f), 50.0f, dielectric, "DielectricSphere");
    auto metal_sphere = std::make_shared<Sphere>(vec3a(0.0f, 150.0f, 145.0f), 50.0f, metal, "MetalSphere");
    auto boundary = std::make_shared<Sphere>(vec3a(360.0f, 150.0f, 145.0f), 50.0f, dielectric, "boundary");
    auto medium = ConstantMedium::new_(boundary, 0.2f, std::make_shared<ConstantTex>(vec3a(0.2f, 0.4f, 0.9f)));

    HitableList spheres;
    for (int i = 0; i < 1000; ++i) { // vec3a_random_range(0., 165.), math.rs:26-28
        SmallRng& rng = RNG();
        float x = rng.gen_f32(), y = rng.gen_f32(), z = rng.gen_f32();
        spheres.push_back(std::make_shared<Sphere>(vec3a(x, y, z) * (165.0f - 0.0f) + 0.0f, 10.0f, white, "Ground"));
    }
    HitablePtr cloud = BvhNode::new_(spheres, 0, 1000);
    cloud = RotateY::new_(cloud, 15.0f);
    cloud = std::make_shared<Translate>(vec3a(-100.0f, 270.0f, 395.0f), cloud);

    HitableList boxes;
    for (int i = 0; i < 20; ++i) {
        for (int j = 0; j < 20; ++j) {
            const float w = 100.0f;
            float x0 = -1000.0f + (float)i * w;
            float z0 = -1000.0f + (float)j * w;
            float y0 = 0.0f;
            float x1 = x0 + w;
            float y1 = RNG().gen_f32() * 100.0f + 1.0f;
            float z1 = z0 + w;
            boxes.push_back(GBox::new_(vec3a(x0, y0, z0), vec3a(x1, y1, z1), ground));
        }
    }
    HitablePtr box_field = BvhNode::new_(boxes, 0, 20 * 20);

    HitableList world = {
        // min.y = 544 is the plane (hitable.rs:284-322 reads min[1]); max.y = 554 is ignored by hit()
        std::make_shared<XZRect>(vec3a(123.0f, 544.0f, 147.0f), vec3a(423.0f, 554.0f, 412.0f), light),
        cloud,
        earth,
        perlin_sphere,
        brown_sphere,
        dielectric_sphere,
        metal_sphere,
        box_field,
        boundary,
        medium,
    };
    Camera cam = Camera::new_(vec3a(478.0f, 278.0f, -600.0f), vec3a(278.0f, 278.0f, 0.0f), vec3a(0.0f, 1.0f, 0.0f), 40.0f, aspect_ratio);
    return build_bvh(world, cam); // :220
}

// BASELINE.json config 4: earthmap-textured sphere under the newport_loft environment sky.
// Built from reference constructors only: Diffuse{ImageTex} as final_scene does
// (demo_scene.rs:160-162), Emission{earth_map} as sphere_scene does (:45,51), sky =
// tex_sky_color over ENV_TEX (demo_scene.rs:22-26, main.rs:63).
std::pair<HitableList, Camera> earth_env_scene(float aspect_ratio) {
    ENV_TEX_set(ImageTex::new_("res/newport_loft.jpg")); // main.rs:63
    SKY_COLOR_set(SkyFn::tex_sky_color);
    auto earth_map = ImageTex::new_("res/earthmap.jpg");
    auto ground = std::make_shared<Diffuse>(CheckerTex::new_(vec3a(0.2f, 0.3f, 0.1f), vec3a(0.9f, 0.9f, 0.9f))); // demo_scene.rs:40
    HitableList world = {
        std::make_shared<Sphere>(vec3a(0.0f, -1000.0f, 0.0f), 1000.0f, ground, "Ground"),
        std::make_shared<Sphere>(vec3a(0.0f, 2.0f, 0.0f), 2.0f, std::make_shared<Diffuse>(earth_map), "EarthSphere"),
        std::make_shared<Sphere>(vec3a(-4.5f, 1.5f, 1.5f), 1.5f, std::make_shared<Dielectric>(1.5f), "Glass"),
        std::make_shared<Sphere>(vec3a(4.5f, 1.5f, -1.0f), 1.5f, std::make_shared<Metal>(vec3a(0.9f, 0.9f, 0.9f), 0.0f), "Mirror"),
        std::make_shared<Sphere>(vec3a(1.5f, 0.6f, 4.0f), 0.6f, std::make_shared<Emission>(earth_map), "EarthLamp"),
    };
    Camera cam = Camera::new_(vec3a(13.0f, 3.0f, 6.0f), vec3a(0.0f, 1.5f, 0.0f), vec3a(0.0f, 1.0f, 0.0f), 30.0f, aspect_ratio);
    return build_bvh(world, cam);
}

// BASELINE.json config 5: GGX metal/rough sweep, 25 x 20 = 500 spheres + ground = 501 spheres,
// every pbr.rs material, parameters on a deterministic grid (no RNG, so the scene is seed-free).
std::pair<HitableList, Camera> pbr_sweep_scene(float aspect_ratio) {
    SKY_COLOR_set(SkyFn::sky_color);
    auto white = std::make_shared<ConstantTex>(vec3a(1.0f, 1.0f, 1.0f));
    HitableList world = {
        std::make_shared<Sphere>(vec3a(0.0f, -1000.0f, 0.0f), 1000.0f,
                                 std::make_shared<Diffuse>(std::make_shared<ConstantTex>(vec3a(0.5f, 0.5f, 0.5f))), "Ground"),
    };
    for (int b = 0; b < 20; ++b) {
        for (int a = 0; a < 25; ++a) {
            float fa = (float)a / 24.0f; // sweep along x
            int sub = b % 4;
            float fs = (float)sub / 3.0f;
            Vec3A col = vec3a(0.25f + 0.7f * fa, 0.9f - 0.6f * fs, 0.3f + 0.6f * (1.0f - fa));
            auto tex = std::make_shared<ConstantTex>(col);
            MaterialPtr mat;
            switch (b / 4) {
            case 0: // anisotropic GGX metal: roughness along x, anisotropy + rotation along z
                mat = std::make_shared<DisneyMetal>(tex, 0.05f + 0.95f * fa, fs, 0.125f * (float)sub);
                break;
            case 1: // rough dielectric-coated plastic
                mat = std::make_shared<RoughPlastic>(white, tex, 0.01f + 0.99f * fa, 1.3f + 0.2f * fs);
                break;
            case 2: // clearcoat lobe only
                mat = std::make_shared<DisneyClearcoat>(fa);
                break;
            case 3: // the diffuse family
                if (sub == 0) mat = std::make_shared<OrenNayar>(tex, fa);
                else if (sub == 1) mat = std::make_shared<BurleyDiffuse>(tex, fa);
                else if (sub == 2) mat = std::make_shared<DisneyDiffuse>(tex, fa, 0.5f);
                else mat = std::make_shared<DisneySheen>(tex, fa);
                break;
            default: // isotropic-looking metal (anisotropic = 0 still takes the aniso branch, pbr.rs:247)
                mat = std::make_shared<DisneyMetal>(tex, 0.05f + 0.95f * fa, 0.0f, 0.0f);
                break;
            }
            Vec3A c = vec3a(((float)a - 12.0f) * 0.9f, 0.35f, ((float)b - 9.5f) * 0.9f);
            world.push_back(std::make_shared<Sphere>(c, 0.35f, mat, "Sweep " + std::to_string(a) + ", " + std::to_string(b)));
        }
    }
    Camera cam = Camera::new_(vec3a(0.0f, 16.0f, 24.0f), vec3a(0.0f, 0.0f, 0.0f), vec3a(0.0f, 1.0f, 0.0f), 40.0f, aspect_ratio);
    return build_bvh(world, cam);
}

} // namespace rtow
