"""Python view of the C++ host mirror (libpthost.so, path_tracer_amd/host/pt_host.h).

Scene / Camera / Image / RayTracerConfig / render keep the reference's names and meaning
(scene.zig, camera.zig, render_targets/ppm_image.zig, raytracer.zig,
renderers/ppm_image.zig); all arithmetic happens in the C++ host library and on the GPU.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import SPHERE_DTYPE, PtGpuError, pt_camera, pt_config
from .gpu import GpuContext

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libpthost.so")

ASPECT_16_9 = float(np.float32(16.0 / 9.0))  # main.zig:78


class pthost_image(C.Structure):
    _fields_ = [("aspect_ratio", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32),
                ("max_px_value", C.c_uint32), ("px_size", C.c_uint32), ("header_size", C.c_uint32),
                ("num_pixels", C.c_uint64), ("data_size", C.c_uint64), ("size", C.c_uint64),
                ("blend_start", C.c_float * 3), ("blend_stop", C.c_float * 3)]


HOST_EXPORTS = ["pthost_setup_scene", "pthost_setup_scene_prng", "pthost_scene_synthetic", "pthost_camera_new", "pthost_camera_preset",
                "pthost_image_new", "pthost_image_header", "pthost_rgb_to_ascii", "pthost_num_digits",
                "pthost_col_num_to_string", "pthost_chunk_size", "pthost_render_scene_ppm", "pthost_render_scene_ppm_multi",
                "pthost_filepath"]

_host = None


def load_host():
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB_PATH):
        raise FileNotFoundError("%s is missing: build it with path_tracer_amd/host/build.sh" % HOST_LIB_PATH)
    _lib.load()  # libptgpu first (libpthost.so NEEDs it; same directory via $ORIGIN)
    H = C.CDLL(HOST_LIB_PATH)
    fp = C.POINTER(C.c_float)
    H.pthost_setup_scene.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_uint32]
    H.pthost_setup_scene_prng.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32]
    H.pthost_scene_synthetic.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32]
    H.pthost_camera_new.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(pt_camera)]
    H.pthost_camera_preset.argtypes = [C.c_int, C.c_float, C.POINTER(pt_camera)]
    H.pthost_image_new.argtypes = [C.c_uint32, C.c_float, C.c_uint32, C.POINTER(pthost_image)]
    H.pthost_image_header.argtypes = [C.POINTER(pthost_image), C.c_char_p, C.c_size_t]
    H.pthost_rgb_to_ascii.argtypes = [C.POINTER(pthost_image), fp, C.c_char_p, C.c_size_t]
    H.pthost_num_digits.argtypes = [C.c_uint64]
    H.pthost_num_digits.restype = C.c_uint32
    H.pthost_col_num_to_string.argtypes = [C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t]
    H.pthost_chunk_size.argtypes = [C.c_uint64, C.c_uint64]
    H.pthost_chunk_size.restype = C.c_uint64
    H.pthost_render_scene_ppm.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
    H.pthost_render_scene_ppm_multi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_uint64, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
    H.pthost_filepath.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_char_p,
                                  C.c_size_t]
    _host = H
    return H


def _f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


class UserError(Exception):
    """errors.zig: UserError.SceneNotAvailable (and the path's other error names)."""


def _raise(rc, what):
    names = {_lib.PT_ERR_SCENE_NOT_AVAILABLE: "SceneNotAvailable", _lib.PT_ERR_VALUE_EXCEEDS_MAX: "ValueExceedsMaxValue",
             _lib.PT_ERR_NO_SPACE: "NoSpaceLeft", _lib.PT_ERR_DIRECTION_NOT_UNIT: "DirectionIsNotUnitVector"}
    if rc in names:
        raise UserError(names[rc])
    raise PtGpuError(rc, what)


class Scene:
    """scene.zig:19-94.  `spheres` is a numpy array of pt_sphere records."""

    def __init__(self):
        self.spheres = np.zeros(0, dtype=SPHERE_DTYPE)

    PRNG = {"xoroshiro128": 0, "xoshiro256": 1, None: 0}

    def setupScene(self, seed, n, prng=None):
        """scene.setupScene(&prng.random(), n) with prng = DefaultPrng.init(seed) (main.zig:100-105).
        prng: which std.rand.DefaultPrng generation — "xoroshiro128" (default: the one that reproduces the reference's
        images/demo.png) or "xoshiro256" (Zig 0.9.1 as recalled); pt_host.h."""
        buf = np.zeros(512, dtype=SPHERE_DTYPE)
        k = load_host().pthost_setup_scene_prng(int(n), int(seed), self.PRNG[prng], buf.ctypes.data, len(buf))
        if k < 0:
            _raise(k, "setupScene")
        self.spheres = buf[:k].copy()
        return self

    def sceneSynthetic(self, n, seed=1):
        """BASELINE config 5: ground sphere + n-1 random spheres (not in the reference)."""
        buf = np.zeros(int(n), dtype=SPHERE_DTYPE)
        k = load_host().pthost_scene_synthetic(int(n), int(seed), buf.ctypes.data, len(buf))
        if k < 0:
            _raise(k, "sceneSynthetic")
        self.spheres = buf[:k].copy()
        return self


class Camera:
    """camera.zig:18-107; `pod` is the pt_camera that crosses the C ABI."""

    def __init__(self, pod):
        self.pod = pod

    @staticmethod
    def new(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist):
        pod = pt_camera()
        rc = load_host().pthost_camera_new(_f3(lookfrom), _f3(lookat), _f3(vup), float(vfov), float(aspect_ratio),
                                           float(aperture), float(focus_dist), C.byref(pod))
        if rc != 0:
            _raise(rc, "Camera.new")
        return Camera(pod)

    @staticmethod
    def preset(num_scene, aspect_ratio=ASPECT_16_9):
        """The per-scene camera set-up of main.zig:107-136."""
        pod = pt_camera()
        rc = load_host().pthost_camera_preset(int(num_scene), float(aspect_ratio), C.byref(pod))
        if rc != 0:
            _raise(rc, "Camera.preset")
        return Camera(pod)


class Image:
    """render_targets/ppm_image.zig:28-104."""

    def __init__(self, width, aspect_ratio=ASPECT_16_9, max_px_value=255):
        self.pod = pthost_image()
        rc = load_host().pthost_image_new(int(width), float(aspect_ratio), int(max_px_value), C.byref(self.pod))
        if rc != 0:
            _raise(rc, "Image.new")
        for k in ("aspect_ratio", "width", "height", "max_px_value", "px_size", "header_size", "num_pixels",
                  "data_size", "size"):
            setattr(self, k, getattr(self.pod, k))
        self.blend_start = tuple(self.pod.blend_start)
        self.blend_stop = tuple(self.pod.blend_stop)

    def header(self):
        buf = C.create_string_buffer(96)
        n = load_host().pthost_image_header(C.byref(self.pod), buf, 96)
        if n < 0:
            _raise(n, "Image.header")
        return buf.raw[:n]

    def rgbToAscii(self, col):
        buf = C.create_string_buffer(32)
        rc = load_host().pthost_rgb_to_ascii(C.byref(self.pod), _f3(col), buf, 32)
        if rc != 0:
            _raise(rc, "Image.rgbToAscii")
        return buf.value


class RayTracerConfig:
    """raytracer.zig:17-23; defaults are main.zig:85-91."""

    def __init__(self, subpixels=4, t_min=0.1, t_max=1000.0, rebounds=6, rays_per_subsample=8):
        self.subpixels, self.t_min, self.t_max = subpixels, t_min, t_max
        self.rebounds, self.rays_per_subsample = rebounds, rays_per_subsample


def numDigits(num):
    return load_host().pthost_num_digits(int(num))


def colNumToString(max_value, value):
    buf = C.create_string_buffer(8)
    rc = load_host().pthost_col_num_to_string(int(max_value), int(value), buf, 8)
    if rc != 0:
        _raise(rc, "colNumToString")
    return buf.value


def chunkSize(num_pixels, num_threads):
    return load_host().pthost_chunk_size(int(num_pixels), int(num_threads))


def filepath(num_scene, cfg, img, seed):
    buf = C.create_string_buffer(256)
    n = load_host().pthost_filepath(int(num_scene), cfg.subpixels, cfg.rays_per_subsample, cfg.rebounds, img.width,
                                    int(seed), buf, 256)
    if n < 0:
        _raise(n, "filepath")
    return buf.value.decode()


def make_pt_config(img, rt, seed=0, istart=0, istop=None, deterministic=False, sample_start=0, sample_stop=0, chunk=0,
                   tile_pixels=0, tile_count=1, tile_index=0, illumination=0, sampler=0):
    c = pt_config()
    c.width, c.height = img.width, img.height
    c.subpixels, c.rays_per_subsample, c.rebounds = rt.subpixels, rt.rays_per_subsample, rt.rebounds
    c.mode = 1 if deterministic else 0
    c.seed = int(seed)
    c.t_min, c.t_max = rt.t_min, rt.t_max
    c.blend_start = (C.c_float * 3)(*img.blend_start)
    c.blend_stop = (C.c_float * 3)(*img.blend_stop)
    c.idx_start = int(istart)
    c.idx_stop = int(img.num_pixels if istop is None else istop)
    c.sample_start, c.sample_stop, c.chunk = int(sample_start), int(sample_stop), int(chunk)
    c.max_px_value = img.max_px_value
    c.tile_pixels, c.tile_count, c.tile_index = int(tile_pixels), int(tile_count), int(tile_index)
    c.illumination = int(illumination)
    c.sampler = int(sampler)   # _lib.PT_SAMPLER_*
    return c


def render(gpu, scene, camera, img, ray_tracer_config, seed, slice_, istart=0, istop=None, deterministic=False, chunk=0, sampler=0):
    """render(cfg) of renderers/ppm_image.zig:30-87 on one GPU: writes the pixel records of
    [istart, istop) into `slice_` (uint8 array of img.size bytes).  The caller writes the
    header afterwards (main.zig:180-182)."""
    assert isinstance(gpu, GpuContext)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(camera.pod)
    cfg = make_pt_config(img, ray_tracer_config, seed, istart, istop, deterministic, chunk=chunk, sampler=sampler)
    gpu.render_ppm(cfg, slice_, img.header_size, img.px_size)
    return cfg


def render_to_file(gpu, num_scene, width, spp, seed, rebounds=6, path=None, deterministic=False):
    """main.zig:78-191 without the prompts; returns the path written."""
    img = Image(width)
    rt = RayTracerConfig(rays_per_subsample=spp, rebounds=rebounds)
    scene = Scene().setupScene(seed, num_scene)
    camera = Camera.preset(num_scene, img.aspect_ratio)
    slice_ = np.zeros(img.size, dtype=np.uint8)
    render(gpu, scene, camera, img, rt, seed, slice_, deterministic=deterministic)
    hdr = img.header()
    slice_[:len(hdr)] = np.frombuffer(hdr, dtype=np.uint8)
    path = path or filepath(num_scene, rt, img, seed)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    slice_.tofile(path)
    return path
