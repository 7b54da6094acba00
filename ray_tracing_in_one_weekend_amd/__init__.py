"""MI355X-native wavefront path tracer behind the construction API of
zhouhang95/ray_tracing_in_one_weekend."""
from . import _ffi
from ._ffi import (GpuLibraryMissing, RtCamera, RtFlatScene, RtParams, RtStats)
from .api import MultiRenderer, Renderer, RtError, Scene, grid_build, make_params, output_file_name, save_png
from .images import decode_rgb32f, register_default_images, register_image

__all__ = ["Renderer", "MultiRenderer", "Scene", "RtError", "make_params", "RtCamera", "RtFlatScene", "RtParams", "RtStats",
           "GpuLibraryMissing", "save_png", "output_file_name", "register_default_images", "register_image", "decode_rgb32f", "_ffi"]
