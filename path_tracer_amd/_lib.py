"""ctypes binding of libptgpu.so (include/ptgpu.h) — the C-ABI drop-in boundary.

The library is built in-tree by path_tracer_amd/csrc/build.sh (hipcc, gfx950).  There is
no fallback of any kind: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTGPU_LIB") or os.path.join(_HERE, "libptgpu.so")  # PTGPU_LIB: diagnostic builds (tools/)

PT_OK = 0
PT_ERR_BAD_ARG = -1
PT_ERR_NO_DEVICE = -2
PT_ERR_HIP = -3
PT_ERR_OOM = -4
PT_ERR_DIRECTION_NOT_UNIT = -5
PT_ERR_VALUE_EXCEEDS_MAX = -6
PT_ERR_SCENE_NOT_AVAILABLE = -7
PT_ERR_NO_SPACE = -8
PT_ERR_UNSUPPORTED = -9
PT_ERR_SCRATCH_LIMIT = -10
PT_ERR_RCCL = -11

PT_LIMIT_PARTIAL_BYTES = 0
PT_LIMIT_PRIMARY_BYTES = 1
PT_LIMIT_PRIMARY_BOUNCES = 2
PT_LIMIT_TREE_MIN_SPHERES = 3
PT_TREE_MIN_SPHERES_DEFAULT = 256      # the library's default
PT_PRIMARY_BYTES_AUTO = (1 << 64) - 1  # the library's default for PT_LIMIT_PRIMARY_BYTES: sized per render from hipMemGetInfo (at most 96 GiB)
PT_PRIMARY_BYTES_DEFAULT = PT_PRIMARY_BYTES_AUTO
PT_SPLIT_TILES, PT_SPLIT_SAMPLES = 0, 1

PT_LAMBERTIAN, PT_METAL, PT_DIELECTRIC = 0, 1, 2
PT_ILLUM_RADIANCE, PT_ILLUM_NORMAL, PT_ILLUM_ALBEDO = 0, 1, 2
PT_SAMPLER_REJECTION, PT_SAMPLER_DIRECT = 0, 1

(PT_FN_CAST_RAY, PT_FN_SPHERE_HIT, PT_FN_SCENE_HIT, PT_FN_SCATTER, PT_FN_REFRACT, PT_FN_SCHLICK, PT_FN_REFLECT,
 PT_FN_LERP, PT_FN_PHILOX, PT_FN_SCENE_HIT_FILTERED, PT_FN_FILTER_BIT, PT_FN_SCENE_HIT_TREE, PT_FN_SAMPLE_DISK,
 PT_FN_SAMPLE_BALL, PT_FN_TREE_BOX_SAFETY) = range(15)
FN_STRIDES = {  # fn -> (floats in, floats out) per record, as include/ptgpu.h documents
    PT_FN_CAST_RAY: (18, 7), PT_FN_SPHERE_HIT: (12, 8), PT_FN_SCENE_HIT: (6, 8), PT_FN_SCATTER: (27, 4),
    PT_FN_REFRACT: (7, 4), PT_FN_SCHLICK: (2, 1), PT_FN_REFLECT: (6, 3), PT_FN_LERP: (9, 4), PT_FN_PHILOX: (5, 4), PT_FN_SCENE_HIT_FILTERED: (6, 8),
    PT_FN_FILTER_BIT: (10, 3), PT_FN_SCENE_HIT_TREE: (6, 8), PT_FN_SAMPLE_DISK: (17, 4), PT_FN_SAMPLE_BALL: (17, 4),
    PT_FN_TREE_BOX_SAFETY: (6, 4)}

# pt_sphere: 36 bytes
SPHERE_DTYPE = np.dtype(
    [("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("radius", "<f4"), ("kind", "<u4"),
     ("ar", "<f4"), ("ag", "<f4"), ("ab", "<f4"), ("param", "<f4")])
assert SPHERE_DTYPE.itemsize == 36


class pt_camera(C.Structure):
    _fields_ = [("lookfrom", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("lens_radius", C.c_float)]


class pt_config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("subpixels", C.c_uint32),
                ("rays_per_subsample", C.c_uint32), ("rebounds", C.c_uint32), ("mode", C.c_uint32),
                ("seed", C.c_uint64), ("t_min", C.c_float), ("t_max", C.c_float),
                ("blend_start", C.c_float * 3), ("blend_stop", C.c_float * 3),
                ("idx_start", C.c_uint32), ("idx_stop", C.c_uint32),
                ("sample_start", C.c_uint32), ("sample_stop", C.c_uint32),
                ("chunk", C.c_uint32), ("max_px_value", C.c_uint32),
                ("tile_pixels", C.c_uint32), ("tile_count", C.c_uint32), ("tile_index", C.c_uint32),
                ("illumination", C.c_uint32), ("sampler", C.c_uint32), ("reserved_", C.c_uint32)]


class pt_stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("candidates", C.c_uint64), ("not_unit", C.c_uint64), ("wave_steps", C.c_uint64),
                ("flush_iters", C.c_uint64), ("kernel_ms", C.c_double),
                ("total_ms", C.c_double), ("segments", C.c_uint64), ("primary_rays", C.c_uint64),
                ("list2_rays", C.c_uint64), ("primary_ms", C.c_double), ("box_tests", C.c_uint64),
                ("lists_ms", C.c_double), ("megakernel_ms", C.c_double), ("primary_candidates", C.c_uint64)]


assert C.sizeof(pt_camera) == 76 and C.sizeof(pt_config) == 112 and C.sizeof(pt_stats) == 136

# every symbol include/ptgpu.h declares
EXPORTS = ["ptgpu_create", "ptgpu_destroy", "ptgpu_set_scene", "ptgpu_set_camera", "ptgpu_render_sum_device",
           "ptgpu_render_sum_device_noalloc", "ptgpu_noalloc_scratch_bytes", "ptgpu_set_limit", "ptgpu_trim", "ptgpu_work_bytes",
           "ptgpu_render_sum_host", "ptgpu_resolve_device", "ptgpu_render_ppm", "ptgpu_render_mean_host",
           "ptgpu_debug_eval", "ptgpu_debug_exhaustive", "ptgpu_debug_list_safety", "ptgpu_get_stats", "ptgpu_scene_info", "ptgpu_tree_info", "ptgpu_debug_tree_check", "ptgpu_strerror", "ptgpu_last_hip_error", "ptgpu_device_info",
           "ptgpu_abi_version",
           "ptgpu_multi_create", "ptgpu_multi_destroy", "ptgpu_multi_num_gpus", "ptgpu_multi_uses_rccl", "ptgpu_multi_ctx",
           "ptgpu_multi_set_scene", "ptgpu_multi_set_camera", "ptgpu_multi_render_sum_host", "ptgpu_multi_render_ppm",
           "ptgpu_multi_plan", "ptgpu_multi_get_stats", "ptgpu_multi_reduce_ms", "ptgpu_multi_last_rccl_error"]


class PtGpuError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = strerror(status) if _lib is not None else str(status)
        super().__init__("%s: %s (%d)" % (what, msg, status))


_lib = None


def load():
    """dlopen libptgpu.so and declare its prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with path_tracer_amd/csrc/build.sh (hipcc --offload-arch=gfx950); "
            "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.ptgpu_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ptgpu_destroy.argtypes = [vp]
    L.ptgpu_set_scene.argtypes = [vp, vp, C.c_uint32]
    L.ptgpu_set_camera.argtypes = [vp, C.POINTER(pt_camera)]
    L.ptgpu_render_sum_device.argtypes = [vp, C.POINTER(pt_config), vp, vp]
    L.ptgpu_render_sum_device_noalloc.argtypes = [vp, C.POINTER(pt_config), vp, vp]
    L.ptgpu_noalloc_scratch_bytes.restype = C.c_uint64
    L.ptgpu_set_limit.argtypes = [vp, C.c_int, C.c_uint64]
    L.ptgpu_trim.argtypes = [vp]
    L.ptgpu_work_bytes.argtypes = [vp]
    L.ptgpu_work_bytes.restype = C.c_uint64
    L.ptgpu_render_sum_host.argtypes = [vp, C.POINTER(pt_config), vp]
    L.ptgpu_render_mean_host.argtypes = [vp, C.POINTER(pt_config), vp]
    L.ptgpu_resolve_device.argtypes = [vp, C.POINTER(pt_config), vp, vp, vp, vp]
    L.ptgpu_render_ppm.argtypes = [vp, C.POINTER(pt_config), vp, C.c_size_t, C.c_uint32, C.c_uint32]
    L.ptgpu_debug_eval.argtypes = [vp, C.c_int, vp, vp, C.c_uint32, C.c_float, C.c_float]
    L.ptgpu_debug_exhaustive.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    L.ptgpu_debug_list_safety.argtypes = [vp, C.POINTER(pt_config), C.POINTER(C.c_uint64)]
    L.ptgpu_get_stats.argtypes = [vp, C.POINTER(pt_stats)]
    L.ptgpu_scene_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ptgpu_tree_info.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ptgpu_debug_tree_check.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.ptgpu_strerror.argtypes = [C.c_int]
    L.ptgpu_strerror.restype = C.c_char_p
    L.ptgpu_last_hip_error.argtypes = [vp]
    L.ptgpu_device_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    L.ptgpu_abi_version.restype = C.c_uint32
    L.ptgpu_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.ptgpu_multi_destroy.argtypes = [vp]
    L.ptgpu_multi_num_gpus.argtypes = [vp]
    L.ptgpu_multi_uses_rccl.argtypes = [vp]
    L.ptgpu_multi_ctx.argtypes = [vp, C.c_int]
    L.ptgpu_multi_ctx.restype = vp
    L.ptgpu_multi_set_scene.argtypes = [vp, vp, C.c_uint32]
    L.ptgpu_multi_set_camera.argtypes = [vp, C.POINTER(pt_camera)]
    L.ptgpu_multi_render_sum_host.argtypes = [vp, C.POINTER(pt_config), C.c_int, vp]
    L.ptgpu_multi_render_ppm.argtypes = [vp, C.POINTER(pt_config), C.c_int, vp, C.c_size_t, C.c_uint32, C.c_uint32]
    L.ptgpu_multi_plan.argtypes = [C.POINTER(pt_config), C.c_int, C.c_int, C.c_int, C.POINTER(pt_config), C.POINTER(C.c_int)]
    L.ptgpu_multi_get_stats.argtypes = [vp, C.c_int, C.POINTER(pt_stats)]
    L.ptgpu_multi_reduce_ms.argtypes = [vp]
    L.ptgpu_multi_reduce_ms.restype = C.c_double
    L.ptgpu_multi_last_rccl_error.argtypes = [vp]
    _lib = L
    return L


def strerror(status):
    return load().ptgpu_strerror(int(status)).decode()


def check(status, what):
    if status != PT_OK:
        raise PtGpuError(status, what)
