"""MI355X-native (gfx950, HIP) voxel path tracer: the trace -> temporal -> denoise hot path of
nolanderc/gpu-voxel-raytracer behind a C ABI (include/vxrt.h, libvxrt.so)."""
from . import _build  # noqa: F401
from .host import (ALL, DENOISE, DENOISE_EDGE, DENOISE_INTERIOR, TEMPORAL, TIMED, TRACE, ACCUM_COLOR, ALBEDO_NODE, DENOISED, NORMAL_DEPTH,  # noqa: F401
                   SAMPLED_COLOR, Camera, Context, DenoiseUniforms, TemporalUniforms, Uniforms, VxrtError)
