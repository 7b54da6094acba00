/*
 * host_main.c — a host for the MI355X path in plain C99, through nothing but include/rtow_host.h and
 * include/rtow_mi355x.h: the program a maintainer of the reference would write in place of main() (main.rs:62-129),
 * here in the one language both the reference's FFI and this image's toolchain speak.  It is compiled by
 * __graft_entry__.build() (gcc -std=c99 -Iinclude) and run on the GPU box by tests/test_gpu_parity.py, which holds its
 * image against the ctypes path byte for byte.
 *
 *   main.rs:63      let sky / scene selection          -> rth_scene_build(name, aspect)     (demo_scene.rs scene fns)
 *   main.rs:64-67   samples_per_pixel, nx, ny          -> RtParams
 *   main.rs:72-73   ThreadPool::new(num_cpus)          -> rt_ctx_create(device)
 *   main.rs:74      let (world, cam) = scene(aspect)   -> rth_scene_flat / rth_scene_camera + rt_scene_upload
 *   main.rs:77-108  the per-column pixel loop          -> rt_render
 *   main.rs:109-128 receive loop, flip, img.save(name) -> the RGB8 image rt_render returns (already flipped) + rth_png_write
 *
 * usage: host_main [scene [nx ny spp [max_depth [out.png [out.rgb8 [image_dir]]]]]]
 *        (default: test_sphere 800 400 128 50 — the workload main.rs:64-74 ships with)
 * image_dir: decoded textures as binary PPM (P6, 8 bit), <image_dir>/earthmap.ppm and newport_loft.ppm, registered under the
 * paths the scene functions pass to ImageTex::new (main.rs:63, demo_scene.rs:42,160).  JPEG decoding is host I/O outside the
 * accelerated path (image::open in the reference); this program takes the pixels already decoded.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtow_host.h"
#include "rtow_mi355x.h"

static int fail(const char* what, const char* text) {
    fprintf(stderr, "host_main: %s: %s\n", what, text ? text : "?");
    return 1;
}

/* texture.rs:176-177 image::open(path).to_rgb32f() for an already decoded 8-bit image: Rgb<f32> = u8 / 255 */
static int register_ppm(const char* dir, const char* file, const char* as_path) {
    char path[1024];
    unsigned w = 0, h = 0, maxv = 0;
    size_t n, i;
    unsigned char* raw;
    float* rgb;
    FILE* f;
    int rc;
    snprintf(path, sizeof path, "%s/%s", dir, file);
    f = fopen(path, "rb");
    if (!f) return 0; /* a scene that needs the image will say so */
    if (fscanf(f, "P6 %u %u %u", &w, &h, &maxv) != 3 || maxv != 255 || fgetc(f) == EOF || w == 0 || h == 0) {
        fclose(f);
        return fail("register_ppm: not an 8-bit P6 file", path);
    }
    n = (size_t)w * h * 3u;
    raw = (unsigned char*)malloc(n);
    rgb = (float*)malloc(n * sizeof(float));
    if (!raw || !rgb || fread(raw, 1, n, f) != n) {
        fclose(f), free(raw), free(rgb);
        return fail("register_ppm: short file", path);
    }
    fclose(f);
    for (i = 0; i < n; ++i) rgb[i] = (float)raw[i] / 255.0f;
    rc = rth_register_image(as_path, w, h, rgb);
    free(raw), free(rgb);
    return rc == 0 ? 0 : fail("rth_register_image", rth_last_error());
}

int main(int argc, char** argv) {
    const char* scene_name = argc > 1 ? argv[1] : "test_sphere";
    const uint32_t nx = argc > 4 ? (uint32_t)atoi(argv[2]) : 800u;   /* main.rs:66 */
    const uint32_t ny = argc > 4 ? (uint32_t)atoi(argv[3]) : 400u;   /* main.rs:67 */
    const uint32_t spp = argc > 4 ? (uint32_t)atoi(argv[4]) : 128u;  /* main.rs:64 */
    const int max_depth = argc > 5 ? atoi(argv[5]) : 50;             /* main.rs:36 MAX_DEPTH */
    char name_buf[64];
    const char* png_path = argc > 6 ? argv[6] : NULL;
    const char* raw_path = argc > 7 ? argv[7] : NULL;
    const char* image_dir = argc > 8 ? argv[8] : NULL;
    RthScene* scene = NULL;
    RtCtx* ctx = NULL;
    RtCamera cam;
    RtParams prm;
    RtStats st;
    float* frame = NULL;
    uint8_t* rgb8 = NULL;
    size_t n;
    int rc = 1;

    if (rt_abi_version() != RT_ABI_VERSION) return fail("rt_abi_version", "the library was built from another header");
    if (nx == 0 || ny == 0 || spp == 0) return fail("arguments", "nx, ny and spp must be positive");

    if (image_dir && (register_ppm(image_dir, "earthmap.ppm", "res/earthmap.jpg") || register_ppm(image_dir, "newport_loft.ppm", "res/newport_loft.jpg")))
        return 1;

    /* main.rs:72-73 — the workers: here one GPU.  The frame is known before the world is built (main.rs:64-67), so the context
     * is told now and requests its work buffers on a thread of its own while this one builds the scene. */
    if (rt_ctx_create(0, &ctx) != 0) {
        fail("rt_ctx_create", rt_last_error(NULL));
        goto done;
    }
    memset(&prm, 0, sizeof prm);
    prm.nx = nx, prm.ny = ny, prm.spp = spp, prm.max_depth = max_depth, prm.seed = 95u;
    if (rt_prepare(ctx, &prm) != 0) {
        fail("rt_prepare", rt_last_error(ctx));
        goto done;
    }

    /* main.rs:74 — the scene function builds world + camera (and sets the sky, demo_scene.rs:38) */
    if (rth_scene_build(scene_name, (float)nx / (float)ny, &scene) != 0) {
        fail("rth_scene_build", rth_last_error());
        goto done;
    }
    if (rth_scene_camera(scene, &cam) != 0) {
        fail("rth_scene_camera", rth_last_error());
        goto done;
    }
    if (rt_scene_upload(ctx, rth_scene_flat(scene)) != 0) {
        fail("rt_scene_upload", rt_last_error(ctx));
        goto done;
    }

    /* main.rs:76 ImageBuffer::new(nx, ny) — in page-locked memory, so that the frame arrives at PCIe rate */
    n = (size_t)nx * ny * 3u;
    frame = (float*)rt_host_alloc(n * sizeof(float));
    rgb8 = (uint8_t*)rt_host_alloc(n);
    if (!frame || !rgb8) {
        fail("rt_host_alloc", "out of page-locked memory");
        goto done;
    }

    /* main.rs:77-108 — every pixel x every sample; main.rs:82 seeds column i with 95 + i, here 95 keys the counter RNG */
    if (rt_render(ctx, &cam, &prm, frame, rgb8, &st) != 0) {
        fail("rt_render", rt_last_error(ctx));
        goto done;
    }
    printf("%s %ux%u, %u spp, depth %d: %llu rays in %.3f s (device %.3f s) = %.1f Mray/s; centre pixel %.6f %.6f %.6f\n", scene_name, nx, ny,
           spp, max_depth, (unsigned long long)st.n_rays, st.seconds_total, st.seconds_device, (double)st.n_rays / st.seconds_device / 1e6,
           frame[((size_t)(ny / 2) * nx + nx / 2) * 3], frame[((size_t)(ny / 2) * nx + nx / 2) * 3 + 1], frame[((size_t)(ny / 2) * nx + nx / 2) * 3 + 2]);

    /* main.rs:110-112,128 — the time-stamped file name and img.save */
    if (!png_path) {
        if (rth_output_file_name(-1, name_buf, (uint32_t)sizeof name_buf) != 0) {
            fail("rth_output_file_name", rth_last_error());
            goto done;
        }
        png_path = name_buf;
    }
    if (rth_png_write(png_path, rgb8, nx, ny) != 0) {
        fail("rth_png_write", rth_last_error());
        goto done;
    }
    if (raw_path) { /* the same bytes, headerless, for a byte-for-byte comparison */
        FILE* f = fopen(raw_path, "wb");
        if (!f || fwrite(rgb8, 1, n, f) != n || fclose(f) != 0) {
            fail("fwrite", raw_path);
            goto done;
        }
    }
    printf("saved %s\n", png_path);
    rc = 0;
done:
    rt_host_free(frame);
    rt_host_free(rgb8);
    if (ctx) rt_ctx_destroy(ctx);
    if (scene) rth_scene_free(scene);
    return rc;
}
