"""Image decode for ImageTex (texture.rs:175-181 `image::open(path).to_rgb32f()`).

JPEG decode is host I/O outside the accelerated path: it is done once here
(PIL) and the SAME decoded texels (u8 / 255 as f32) are handed to both the GPU library and the
CPU oracle, so decoder differences (jpeg-decoder vs libjpeg-turbo) cannot enter parity.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi

ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
# the relative paths the reference passes to ImageTex::new (main.rs:63, demo_scene.rs:42,160)
DEFAULT_IMAGES = ("res/earthmap.jpg", "res/newport_loft.jpg")


def decode_rgb32f(path):
    """Returns (h, w, 3) float32 = u8 / 255.0, row 0 = top, like `to_rgb32f`."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return (a.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def register_image(ref_path, pixels):
    """Registers decoded pixels under the path string the scene code uses."""
    lib = _ffi.load_host_library()
    px = np.ascontiguousarray(pixels, dtype=np.float32)
    h, w, c = px.shape
    assert c == 3
    rc = lib.rth_register_image(ref_path.encode(), w, h, px.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise RuntimeError(lib.rth_last_error().decode())


def register_default_images(asset_dir=None):
    """Decodes assets/res/*.jpg and registers them as "res/<name>.jpg"."""
    asset_dir = asset_dir or ASSET_DIR
    for rel in DEFAULT_IMAGES:
        p = os.path.join(asset_dir, rel)
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p} not found (ImageTex::new({rel!r}) would panic in the reference)")
        register_image(rel, decode_rgb32f(p))
