"""MI355X-native ray-tracing core for ComputeRaytracer's hot path.

The product is computeraytracer_amd/lib/libcrt.so (C ABI: include/crt/abi.h);
this package only binds it for the Python harnesses.
"""
from . import abi  # noqa: F401
from .abi import Context, HostScene, builtin_scene, frame_seeds  # noqa: F401
