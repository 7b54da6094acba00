#include "rtow_host.h"
#include <cstdio>
#include <vector>
int main() {
    std::vector<float> img(64 * 32 * 3, 0.5f);
    rth_register_image("res/earthmap.jpg", 64, 32, img.data());
    rth_register_image("res/newport_loft.jpg", 64, 32, img.data());
    const char* names[] = {"sphere_scene", "test_sphere", "simple_light_scene", "cornell_box", "final_scene", "earth_env_scene", "pbr_sweep_scene"};
    for (const char* n : names) {
        RthScene* s = nullptr;
        int rc = rth_scene_build(n, 1.5f, &s);
        const RtFlatScene* f = rth_scene_flat(s);
        std::printf("%s rc=%d spheres=%u rects=%u media=%u\n", n, rc, f ? f->n_spheres : 0, f ? f->n_rects : 0, f ? f->n_media : 0);
        rth_scene_free(s);
    }
    RthScene* s = nullptr;
    rth_scene_new(&s);
    float c[3] = {0.5f, 0.5f, 0.5f};
    uint32_t t = rth_tex_constant(s, c);
    float z[3] = {0, 0, 0}, p4[4] = {1.5f, 0, 0, 0};
    uint32_t m = rth_material(s, 1, t, 0xFFFFFFFFu, z, p4);
    for (int i = 0; i < 7; ++i) {
        float ctr[3] = {(float)i, 0.f, (float)-i};
        uint32_t h = rth_sphere(s, ctr, 0.5f, m, "x");
        float off[3] = {1, 2, 3};
        h = rth_translate(s, h, off);
        h = rth_rotate_y(s, h, 33.0f);
        if (i % 3 == 0) rth_constant_medium(s, h, 0.3f, t);
        float bb[6];
        rth_hitable_bbox(s, h, bb);
    }
    float lf[3] = {0, 0, 5}, la[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    rth_set_camera(s, lf, la, up, 40.f, 1.f);
    std::printf("finish rc=%d\n", rth_scene_finish(s, 1));
    unsigned char px[4 * 3 * 3] = {0};
    std::printf("png rc=%d\n", rth_png_write("host_san.png", px, 4, 3));
    char name[64];
    std::printf("name rc=%d %s\n", rth_output_file_name(-1, name, 64), name);
    rth_scene_free(s);
    return 0;
}
