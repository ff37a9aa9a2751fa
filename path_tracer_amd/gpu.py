"""GpuContext — thin Python handle on a libptgpu context (one GPU).

Everything here goes through the C ABI of include/ptgpu.h; nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PT_OK, PtGpuError, check, pt_camera, pt_config, pt_stats


class GpuContext:
    def __init__(self, device=0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        check(self._L.ptgpu_create(int(device), C.byref(self._h)), "ptgpu_create")
        self.device = int(device)

    def close(self):
        if self._h:
            self._L.ptgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads -------------------------------------------------------------------
    def set_scene(self, spheres):
        sp = np.ascontiguousarray(spheres, dtype=_lib.SPHERE_DTYPE)
        check(self._L.ptgpu_set_scene(self._h, sp.ctypes.data, len(sp)), "ptgpu_set_scene")
        self.nspheres = len(sp)

    def set_camera(self, cam):
        if not isinstance(cam, pt_camera):
            c = pt_camera()
            C.memmove(C.byref(c), C.byref(cam), C.sizeof(pt_camera))
            cam = c
        check(self._L.ptgpu_set_camera(self._h, C.byref(cam)), "ptgpu_set_camera")

    @staticmethod
    def as_config(cfg):
        if isinstance(cfg, pt_config):
            return cfg
        c = pt_config()
        assert C.sizeof(cfg) == C.sizeof(pt_config)
        C.memmove(C.byref(c), C.byref(cfg), C.sizeof(pt_config))
        return c

    # ---- renders ---------------------------------------------------------------------
    def render_sum(self, cfg, allow_not_unit=False):
        """Radiance sums as an (H, W, 3) float32 array (ptgpu_render_sum_host)."""
        cfg = self.as_config(cfg)
        out = np.zeros((cfg.height, cfg.width, 3), dtype=np.float32)
        rc = self._L.ptgpu_render_sum_host(self._h, C.byref(cfg), out.ctypes.data)
        if rc != PT_OK and not (allow_not_unit and rc == _lib.PT_ERR_DIRECTION_NOT_UNIT):
            raise PtGpuError(rc, "ptgpu_render_sum_host")
        return out

    def render_mean(self, cfg):
        cfg = self.as_config(cfg)
        out = np.zeros((cfg.height, cfg.width, 3), dtype=np.float32)
        check(self._L.ptgpu_render_mean_host(self._h, C.byref(cfg), out.ctypes.data), "ptgpu_render_mean_host")
        return out

    def render_sum_device(self, cfg, d_ptr, stream=0):
        """Radiance sums into device memory at address d_ptr (W*H*3 floats)."""
        cfg = self.as_config(cfg)
        check(self._L.ptgpu_render_sum_device(self._h, C.byref(cfg), C.c_void_p(int(d_ptr)), C.c_void_p(int(stream))),
              "ptgpu_render_sum_device")

    def render_sum_device_noalloc(self, cfg, d_ptr, stream=0):
        """render_sum_device without any device allocation (static 128 MiB scratch inside the library)."""
        cfg = self.as_config(cfg)
        check(self._L.ptgpu_render_sum_device_noalloc(self._h, C.byref(cfg), C.c_void_p(int(d_ptr)), C.c_void_p(int(stream))),
              "ptgpu_render_sum_device_noalloc")

    def set_limit(self, which, value):
        check(self._L.ptgpu_set_limit(self._h, int(which), int(value)), "ptgpu_set_limit")

    def trim(self):
        """ptgpu_trim: free every work buffer the renders have grown (the scene stays)."""
        check(self._L.ptgpu_trim(self._h), "ptgpu_trim")

    def work_bytes(self):
        return int(self._L.ptgpu_work_bytes(self._h))

    def resolve_device(self, cfg, d_sum, d_records=0, d_rgb8=0, stream=0):
        cfg = self.as_config(cfg)
        check(self._L.ptgpu_resolve_device(self._h, C.byref(cfg), C.c_void_p(int(d_sum)), C.c_void_p(int(d_records)),
                                           C.c_void_p(int(d_rgb8)), C.c_void_p(int(stream))), "ptgpu_resolve_device")

    def render_ppm(self, cfg, slice_, header_size, px_size):
        """render() replacement: fills the pixel records of `slice_` (uint8 array)."""
        cfg = self.as_config(cfg)
        assert slice_.dtype == np.uint8 and slice_.flags["C_CONTIGUOUS"]
        check(self._L.ptgpu_render_ppm(self._h, C.byref(cfg), slice_.ctypes.data, slice_.size, int(header_size),
                                       int(px_size)), "ptgpu_render_ppm")

    def debug_eval(self, fn, inputs, t_min=0.1, t_max=1000.0):
        n_in, n_out = _lib.FN_STRIDES[fn]
        a = np.ascontiguousarray(inputs)
        assert a.dtype in (np.float32, np.uint32) and a.ndim == 2 and a.shape[1] == n_in
        out = np.zeros((a.shape[0], n_out), dtype=np.float32)
        check(self._L.ptgpu_debug_eval(self._h, int(fn), a.ctypes.data, out.ctypes.data, a.shape[0], float(t_min),
                                       float(t_max)), "ptgpu_debug_eval")
        return out

    def debug_list_safety(self, cfg):
        """ptgpu_debug_list_safety: dict of the check's counts (missing_a, missing_b and closest_differs must be 0)"""
        cfg = self.as_config(cfg)
        out = (C.c_uint64 * 7)()
        check(self._L.ptgpu_debug_list_safety(self._h, C.byref(cfg), out), "ptgpu_debug_list_safety")
        keys = ("accepted_a", "missing_a", "accepted_b", "missing_b", "closest_differs", "rays_a", "rays_b")
        return dict(zip(keys, (int(v) for v in out)))

    def debug_exhaustive(self, which):
        """ptgpu_debug_exhaustive: mismatches of a lean f32 sequence against the compiler's over all 2^32 inputs (must be 0)"""
        bad = C.c_uint64(1)
        check(self._L.ptgpu_debug_exhaustive(self._h, int(which), C.byref(bad)), "ptgpu_debug_exhaustive")
        return bad.value

    # ---- diagnostics -------------------------------------------------------------------
    def stats(self):
        st = pt_stats()
        check(self._L.ptgpu_get_stats(self._h, C.byref(st)), "ptgpu_get_stats")
        return st

    def scene_info(self):
        n, shared = C.c_uint32(), C.c_uint32()
        check(self._L.ptgpu_scene_info(self._h, C.byref(n), C.byref(shared)), "ptgpu_scene_info")
        return {"nspheres": n.value, "shared_ordinate_tests": shared.value}

    def tree_info(self):
        """inner nodes (0: the uploaded scene is traced by the scan) and depth of the scene's sphere tree"""
        nodes, depth = C.c_uint32(), C.c_uint32()
        check(self._L.ptgpu_tree_info(self._h, C.byref(nodes), C.byref(depth)), "ptgpu_tree_info")
        return {"nodes": nodes.value, "depth": depth.value}

    def device_info(self):
        cu, khz = C.c_int(), C.c_int()
        arch = C.create_string_buffer(64)
        check(self._L.ptgpu_device_info(self._h, C.byref(cu), C.byref(khz), arch, 64), "ptgpu_device_info")
        return {"compute_units": cu.value, "clock_khz": khz.value, "arch": arch.value.decode()}


SPLITS = {"tiles": _lib.PT_SPLIT_TILES, "samples": _lib.PT_SPLIT_SAMPLES}


def multi_plan(cfg, split, num_gpus, gpu):
    """ptgpu_multi_plan: the pt_config GPU `gpu` of `num_gpus` renders, or None when its share is empty.  Pure host
    arithmetic in the library (no device needed)."""
    cfg = GpuContext.as_config(cfg)
    out, has = pt_config(), C.c_int()
    check(_lib.load().ptgpu_multi_plan(C.byref(cfg), SPLITS.get(split, split), int(num_gpus), int(gpu), C.byref(out), C.byref(has)),
          "ptgpu_multi_plan")
    return out if has.value else None


class MultiGpu:
    """ptgpu_multi_*: all GPUs of one node from one process — one host thread, context and stream per GPU and an
    RCCL reduce(SUM) of the fp32 sums to GPU 0 (replaces the thread block of main.zig:142-165)."""

    def __init__(self, num_gpus=1, devices=None):
        self._L = _lib.load()
        self._h = C.c_void_p()
        dev = (C.c_int * num_gpus)(*devices) if devices is not None else None
        check(self._L.ptgpu_multi_create(dev, int(num_gpus), C.byref(self._h)), "ptgpu_multi_create")
        self.num_gpus = self._L.ptgpu_multi_num_gpus(self._h)

    def close(self):
        if self._h:
            self._L.ptgpu_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_rccl(self):
        return bool(self._L.ptgpu_multi_uses_rccl(self._h))

    def set_scene(self, spheres):
        sp = np.ascontiguousarray(spheres, dtype=_lib.SPHERE_DTYPE)
        check(self._L.ptgpu_multi_set_scene(self._h, sp.ctypes.data, len(sp)), "ptgpu_multi_set_scene")

    def set_camera(self, cam):
        c = pt_camera()
        C.memmove(C.byref(c), C.byref(cam), C.sizeof(pt_camera))
        check(self._L.ptgpu_multi_set_camera(self._h, C.byref(c)), "ptgpu_multi_set_camera")

    def render_sum(self, cfg, split="samples"):
        cfg = GpuContext.as_config(cfg)
        out = np.zeros((cfg.height, cfg.width, 3), dtype=np.float32)
        check(self._L.ptgpu_multi_render_sum_host(self._h, C.byref(cfg), SPLITS[split], out.ctypes.data), "ptgpu_multi_render_sum_host")
        return out

    def render_ppm(self, cfg, split, slice_, header_size, px_size):
        cfg = GpuContext.as_config(cfg)
        assert slice_.dtype == np.uint8 and slice_.flags["C_CONTIGUOUS"]
        check(self._L.ptgpu_multi_render_ppm(self._h, C.byref(cfg), SPLITS[split], slice_.ctypes.data, slice_.size, int(header_size),
                                             int(px_size)), "ptgpu_multi_render_ppm")

    def stats(self, gpu=-1):
        st = pt_stats()
        check(self._L.ptgpu_multi_get_stats(self._h, int(gpu), C.byref(st)), "ptgpu_multi_get_stats")
        return st

    def reduce_ms(self):
        return self._L.ptgpu_multi_reduce_ms(self._h)
