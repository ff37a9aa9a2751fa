"""ctypes binding of the CPU oracle (oracle/libptoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package (path_tracer_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libptoracle.so")

SPHERE_DTYPE = np.dtype(
    [("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4"), ("radius", "<f4"), ("kind", "<u4"),
     ("ar", "<f4"), ("ag", "<f4"), ("ab", "<f4"), ("param", "<f4")])
assert SPHERE_DTYPE.itemsize == 36


class Camera(C.Structure):
    _fields_ = [("lookfrom", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("lens_radius", C.c_float)]


class Config(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("subpixels", C.c_uint32),
                ("rays_per_subsample", C.c_uint32), ("rebounds", C.c_uint32), ("mode", C.c_uint32),
                ("seed", C.c_uint64), ("t_min", C.c_float), ("t_max", C.c_float),
                ("blend_start", C.c_float * 3), ("blend_stop", C.c_float * 3),
                ("idx_start", C.c_uint32), ("idx_stop", C.c_uint32),
                ("sample_start", C.c_uint32), ("sample_stop", C.c_uint32),
                ("chunk", C.c_uint32), ("max_px_value", C.c_uint32),
                ("tile_pixels", C.c_uint32), ("tile_count", C.c_uint32), ("tile_index", C.c_uint32),
                ("illumination", C.c_uint32), ("sampler", C.c_uint32), ("reserved_", C.c_uint32)]


SAMPLER_REJECTION, SAMPLER_DIRECT = 0, 1   # pt_oracle.h


class Image(C.Structure):
    _fields_ = [("aspect_ratio", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32),
                ("max_px_value", C.c_uint32), ("px_size", C.c_uint32), ("header_size", C.c_uint32),
                ("num_pixels", C.c_uint64), ("data_size", C.c_uint64), ("size", C.c_uint64),
                ("blend_start", C.c_float * 3), ("blend_stop", C.c_float * 3)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("not_unit", C.c_uint64)]


PRNG_XOROSHIRO128, PRNG_XOSHIRO256 = 0, 1   # pt_oracle.h: std.rand.DefaultPrng generations; the default is the demo-pinned one
PRNG_KINDS = {"xoroshiro128": PRNG_XOROSHIRO128, "xoshiro256": PRNG_XOSHIRO256, None: PRNG_XOROSHIRO128}


class ZigRng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4), ("kind", C.c_uint32), ("pad_", C.c_uint32)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("pt_oracle.c", "pt_oracle.h", "Makefile")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libptoracle.so"] + (["-B"] if force else []))
    return _SO


_lib = None


def select_native():
    """bench.py's cpu_baseline leg only: build the same source with -O3 -march=native -ffp-contract=off ON THE BOX THAT
    RUNS IT (oracle/Makefile target libptoracle_native.so; the same IEEE operations, so the same bits) and use it."""
    global _SO, _lib
    subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "libptoracle_native.so"])  # -B: never reuse a copy built on another host
    _SO, _lib = os.path.join(_HERE, "libptoracle_native.so"), None
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        L.orc_render_sum.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Camera), C.POINTER(Config), C.c_void_p,
                                     C.c_uint32, C.POINTER(Stats)]
        L.orc_render_sum_zig.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Camera), C.POINTER(Config),
                                         C.POINTER(ZigRng), C.c_void_p, C.POINTER(Stats)]
        L.orc_resolve_mean.argtypes = [C.POINTER(Config), C.c_void_p, C.c_void_p]
        L.orc_resolve_mean.restype = None
        L.orc_resolve_ppm.argtypes = [C.POINTER(Config), C.POINTER(Image), C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_render_ppm.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Camera), C.POINTER(Config), C.POINTER(Image),
                                     C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(Stats)]
        L.orc_sample_radiance.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Camera), C.POINTER(Config), C.c_uint32,
                                          C.c_uint32, C.c_int, fp]
        L.orc_setup_scene.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(ZigRng)]
        L.orc_setup_scene_prng.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(ZigRng)]
        L.orc_zig_seed_kind.argtypes = [C.POINTER(ZigRng), C.c_uint64, C.c_uint32]
        L.orc_zig_seed_kind.restype = None
        L.orc_scene_synthetic.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32]
        L.orc_camera_new.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]
        L.orc_camera_new.restype = None
        L.orc_camera_preset.argtypes = [C.c_int, C.c_float, C.POINTER(Camera)]
        L.orc_image_new.argtypes = [C.c_uint32, C.c_float, C.c_uint32, C.POINTER(Image)]
        L.orc_image_new.restype = None
        L.orc_image_header.argtypes = [C.POINTER(Image), C.c_char_p, C.c_size_t]
        L.orc_rgb_to_ascii.argtypes = [C.POINTER(Image), fp, C.c_char_p, C.c_size_t]
        L.orc_num_digits.argtypes = [C.c_uint64]
        L.orc_num_digits.restype = C.c_uint32
        L.orc_col_num_to_string.argtypes = [C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t]
        L.orc_degrees_to_radians.argtypes = [C.c_float]
        L.orc_degrees_to_radians.restype = C.c_float
        L.orc_chunk_size.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_chunk_size.restype = C.c_uint64
        L.orc_zig_seed.argtypes = [C.POINTER(ZigRng), C.c_uint64]
        L.orc_zig_seed.restype = None
        L.orc_zig_next.argtypes = [C.POINTER(ZigRng)]
        L.orc_zig_next.restype = C.c_uint64
        L.orc_zig_float.argtypes = [C.POINTER(ZigRng)]
        L.orc_zig_float.restype = C.c_float
        L.orc_splitmix64_next.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_splitmix64_next.restype = C.c_uint64
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_philox4x32_10.restype = None
        L.orc_u32_to_float.argtypes = [C.c_uint32]
        L.orc_u32_to_float.restype = C.c_float
        for name in ("orc_length", "orc_length_squared"):
            getattr(L, name).argtypes = [fp]
            getattr(L, name).restype = C.c_float
        L.orc_unit_vector.argtypes = [fp, fp]
        L.orc_unit_vector.restype = None
        L.orc_dot.argtypes = [fp, fp]
        L.orc_dot.restype = C.c_float
        L.orc_cross.argtypes = [fp, fp, fp]
        L.orc_cross.restype = None
        L.orc_point_at.argtypes = [fp, fp, C.c_float, fp]
        L.orc_point_at.restype = None
        L.orc_cast_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, fp, C.c_uint32, C.POINTER(C.c_uint32), fp, fp]
        L.orc_cast_ray.restype = None
        L.orc_sphere_hit.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, fp, fp, fp]
        L.orc_scene_hit.argtypes = [C.c_void_p, C.c_uint32, fp, fp, C.c_float, C.c_float, fp, fp, fp]
        L.orc_reflect.argtypes = [fp, fp, fp]
        L.orc_reflect.restype = None
        L.orc_refract.argtypes = [fp, fp, C.c_float, fp]
        L.orc_schlick.argtypes = [C.c_float, C.c_float]
        L.orc_schlick.restype = C.c_float
        L.orc_pow.argtypes = [C.c_float, C.c_float]
        L.orc_pow.restype = C.c_float
        L.orc_scatter.argtypes = [C.c_void_p, fp, fp, fp, fp, C.c_uint32, C.POINTER(C.c_uint32), fp, fp, fp]
        L.orc_scatter.restype = None
        L.orc_lerp.argtypes = [fp, fp, fp, fp]
        for name in ("orc_sample_disk", "orc_sample_ball"):
            getattr(L, name).argtypes = [C.c_uint32, fp, C.c_uint32, C.POINTER(C.c_uint32), fp]
            getattr(L, name).restype = None
        L.orc_scene_hit_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.orc_scene_hit_batch.restype = None
        L.orc_sample_batch.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_sample_batch.restype = None
        _lib = L
    return _lib


def f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---- convenience wrappers --------------------------------------------------------
def setup_scene(n_scene, seed=0, prng=None):
    """scene.zig:86-94 via main.zig:100-105; returns (spheres, zig rng state after).
    prng: "xoroshiro128" (default: the generator behind the reference's images/demo.png) or "xoshiro256"."""
    buf = np.zeros(512, dtype=SPHERE_DTYPE)
    rng = ZigRng()
    k = lib().orc_setup_scene_prng(int(n_scene), int(seed), PRNG_KINDS[prng], buf.ctypes.data, len(buf), C.byref(rng))
    if k < 0:
        raise ValueError("orc_setup_scene: %d" % k)
    return buf[:k].copy(), rng


def scene_synthetic(n, seed=1):
    buf = np.zeros(n, dtype=SPHERE_DTYPE)
    k = lib().orc_scene_synthetic(int(n), int(seed), buf.ctypes.data, len(buf))
    if k < 0:
        raise ValueError("orc_scene_synthetic: %d" % k)
    return buf[:k].copy()


ASPECT_16_9 = float(np.float32(16.0 / 9.0))


def camera_preset(n_scene, aspect=ASPECT_16_9):
    cam = Camera()
    rc = lib().orc_camera_preset(int(n_scene), float(aspect), C.byref(cam))
    if rc != 0:
        raise ValueError("orc_camera_preset: %d" % rc)
    return cam


def image_new(width, aspect=ASPECT_16_9, max_px_value=255):
    img = Image()
    lib().orc_image_new(int(width), float(aspect), int(max_px_value), C.byref(img))
    return img


def make_config(img, subpixels=4, rays_per_subsample=1, rebounds=6, mode=0, seed=0, t_min=0.1, t_max=1000.0,
                idx_start=0, idx_stop=None, sample_start=0, sample_stop=0, chunk=0, tile_pixels=0, tile_count=1,
                tile_index=0, illumination=0, sampler=0):
    """RayTracerConfig as main.zig:85-91 sets it, plus the work range."""
    cfg = Config()
    cfg.width, cfg.height = img.width, img.height
    cfg.subpixels, cfg.rays_per_subsample = subpixels, rays_per_subsample
    cfg.rebounds, cfg.mode, cfg.seed = rebounds, mode, seed
    cfg.t_min, cfg.t_max = t_min, t_max
    cfg.blend_start = img.blend_start
    cfg.blend_stop = img.blend_stop
    cfg.idx_start = idx_start
    cfg.idx_stop = img.width * img.height if idx_stop is None else idx_stop
    cfg.sample_start, cfg.sample_stop, cfg.chunk = sample_start, sample_stop, chunk
    cfg.max_px_value = img.max_px_value
    cfg.tile_pixels, cfg.tile_count, cfg.tile_index = tile_pixels, tile_count, tile_index
    cfg.illumination = illumination
    cfg.sampler = sampler
    return cfg


def render_sum(spheres, cam, cfg, nthreads=1):
    out = np.zeros((cfg.height, cfg.width, 3), dtype=np.float32)
    st = Stats()
    sp = np.ascontiguousarray(spheres)
    rc = lib().orc_render_sum(sp.ctypes.data, len(sp), C.byref(cam), C.byref(cfg), out.ctypes.data, nthreads, C.byref(st))
    return rc, out, st


def render_sum_zig(spheres, cam, cfg, rng):
    """The reference's own sampler: main.zig:166-177's single-thread render() drawing from the sequential std.rand
    stream that built the scene (`rng`: the state setup_scene returned; advanced in place) and material.zig:15's
    module-level DefaultPrng.init(0) for the scatters."""
    out = np.zeros((cfg.height, cfg.width, 3), dtype=np.float32)
    st = Stats()
    sp = np.ascontiguousarray(spheres)
    rc = lib().orc_render_sum_zig(sp.ctypes.data, len(sp), C.byref(cam), C.byref(cfg), C.byref(rng), out.ctypes.data,
                                  C.byref(st))
    return rc, out, st


def sample_batch(which, sampler, xi):
    """which: "disk" | "ball"; xi: (n, stride) float32 draws -> (n, 4) float32: the vector and the draws it consumed."""
    xi = np.ascontiguousarray(xi, dtype=np.float32)
    out = np.zeros((xi.shape[0], 4), dtype=np.float32)
    lib().orc_sample_batch(0 if which == "disk" else 1, int(sampler), xi.ctypes.data, xi.shape[0], xi.shape[1], out.ctypes.data)
    return out


def scene_hit_batch(spheres, rays, t_min=0.1, t_max=1000.0):
    """Scene.is_hit (scene.zig:96-110) for (n, 6) float32 rays -> (n, 8): index or -1, t, p, n — the record of PT_FN_SCENE_HIT."""
    sp = np.ascontiguousarray(spheres)
    r = np.ascontiguousarray(rays, dtype=np.float32)
    out = np.zeros((r.shape[0], 8), dtype=np.float32)
    lib().orc_scene_hit_batch(sp.ctypes.data, len(sp), r.ctypes.data, r.shape[0], float(t_min), float(t_max), out.ctypes.data)
    return out


def resolve_mean(cfg, rgb_sum):
    out = np.zeros_like(rgb_sum)
    lib().orc_resolve_mean(C.byref(cfg), rgb_sum.ctypes.data, out.ctypes.data)
    return out


def resolve_ppm(cfg, img, rgb_sum):
    """The full .ppm byte buffer as main.zig builds it: pixel records + header."""
    buf = np.zeros(img.size, dtype=np.uint8)
    rc = lib().orc_resolve_ppm(C.byref(cfg), C.byref(img), rgb_sum.ctypes.data, buf.ctypes.data, buf.size)
    hdr = C.create_string_buffer(64)
    n = lib().orc_image_header(C.byref(img), hdr, 64)
    buf[:n] = np.frombuffer(hdr.raw[:n], dtype=np.uint8)
    return rc, buf
