"""The reference's `main()` (main.rs:62-129) on the accelerated path:

    python -m ray_tracing_in_one_weekend_amd.render --scene sphere_scene --nx 1920 --ny 1080 --spp 256

builds the scene with the host mirror of demo_scene.rs, renders it on the GPU, prints progress, saves partial
images while rendering and the final PNG under the reference's time-stamped name.  Defaults are the constants in
main.rs: 800x400, aspect nx/ny, 128 spp, MAX_DEPTH 50, seed base 95, test_sphere.
There is no CPU fallback: without librtow_mi355x.so and a GPU this exits with the library's error."""
import argparse
import sys
import time

from . import Renderer, Scene, make_params, output_file_name, register_default_images, save_png
from . import _ffi

SCENES = ("test_sphere", "sphere_scene", "simple_light_scene", "cornell_box", "final_scene", "earth_env_scene",
          "pbr_sweep_scene")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--scene", default="test_sphere", choices=SCENES)          # main.rs:74
    ap.add_argument("--nx", type=int, default=800)                             # main.rs:66
    ap.add_argument("--ny", type=int, default=400)                             # main.rs:67
    ap.add_argument("--spp", type=int, default=128)                            # main.rs:64
    ap.add_argument("--max-depth", type=int, default=50)                       # main.rs:36
    ap.add_argument("--seed", type=int, default=95)                            # main.rs:82
    ap.add_argument("--russian-roulette", action="store_true", help="main.rs:49-53 (commented out there)")
    ap.add_argument("--preview-every", type=int, default=0, metavar="SPP",
                    help="save a partial image every SPP samples (main.rs:114-123 saves every 10 columns)")
    ap.add_argument("--out", default=None, help="default: the reference's <local time>.png (main.rs:110-112)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    ny = a.ny
    aspect = a.nx / ny                                                        # main.rs:68
    file_name = a.out or output_file_name()

    t0 = time.perf_counter()                                                  # EZTimer, main.rs:70
    register_default_images()
    scene = Scene.build(a.scene, aspect)
    rend = Renderer(a.device)
    rend.upload(scene)
    flags = _ffi.FLAG_RUSSIAN_ROULETTE if a.russian_roulette else 0
    params = make_params(a.nx, ny, a.spp, max_depth=a.max_depth, seed=a.seed, spp_slice=a.preview_every, flags=flags)
    if a.preview_every:
        def progress(done, total, rgb8):
            print(f"{done}/{total}", file=sys.stderr)                          # main.rs:116-118
            save_png(file_name, rgb8)                                          # main.rs:119-123
        rend.set_progress(progress)
    _, rgb8, st = rend.render(scene.camera, params, want_rgb8=True)
    print(f"elapsed {time.perf_counter() - t0:.3f} s", file=sys.stderr)        # drop(t), main.rs:126
    save_png(file_name, rgb8)                                                  # main.rs:127-128
    print(f"{file_name}: {a.nx}x{ny}, {a.spp} spp, {st.n_rays} rays, {st.n_rays / st.seconds_device / 1e6:.0f} Mray/s on the device",
          file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
