"""Host side above the C ABI (C++ mirror of scene.zig / camera.zig / render_targets/ppm_image.zig /
utils.zig) against the oracle, and the C-ABI library's exports.  No GPU needed: nothing here
launches a kernel."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libptgpu_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly what include/ptgpu.h declares."""
    from path_tracer_amd import _lib
    L = _lib.load()
    header = open(os.path.join(ROOT, "include", "ptgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(ptgpu_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_lib.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.ptgpu_abi_version() == 3
    assert _lib.strerror(_lib.PT_ERR_DIRECTION_NOT_UNIT) == "DirectionIsNotUnitVector"
    assert _lib.strerror(_lib.PT_ERR_SCENE_NOT_AVAILABLE) == "SceneNotAvailable"


def test_pod_layouts_match_header():
    from path_tracer_amd import _lib
    assert _lib.SPHERE_DTYPE.itemsize == 36
    assert C.sizeof(_lib.pt_camera) == 76
    assert C.sizeof(_lib.pt_config) == 112 and _lib.pt_config.sampler.offset == 104
    assert _lib.pt_config.seed.offset == 24 and _lib.pt_config.tile_pixels.offset == 88
    assert C.sizeof(_lib.pt_stats) == 136 and _lib.pt_stats.lists_ms.offset == 112 and _lib.pt_stats.primary_candidates.offset == 128 and _lib.pt_stats.segments.offset == 72 and _lib.pt_stats.primary_ms.offset == 96 and _lib.pt_stats.box_tests.offset == 104


def test_no_gpu_means_error_not_fallback():
    """Without a gfx950 device the product path fails loudly (there is no CPU fallback)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from path_tracer_amd.gpu import GpuContext\n"
            "from path_tracer_amd._lib import PtGpuError\n"
            "try:\n    GpuContext(0)\nexcept PtGpuError as e:\n    print('ERR', e.status)\nelse:\n    print('CREATED')\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120).stdout
    assert "ERR -2" in out, out  # PT_ERR_NO_DEVICE


def test_host_exports():
    from path_tracer_amd import host
    H = host.load_host()
    for name in host.HOST_EXPORTS:
        assert getattr(H, name) is not None


@pytest.mark.parametrize("n_scene,seed", [(18, 0), (19, 3), (20, 42), (21, 456), (21, 0), (21, 2 ** 40 + 7)])
def test_scene_matches_oracle(orc, n_scene, seed):
    """scene.zig:32-94 restated twice (C++ host, C oracle): identical sphere tables."""
    from path_tracer_amd import host
    got = host.Scene().setupScene(seed, n_scene).spheres
    ref, _ = orc.setup_scene(n_scene, seed)
    assert got.tobytes() == ref.tobytes()


def test_scene_not_available():
    from path_tracer_amd import host
    with pytest.raises(host.UserError, match="SceneNotAvailable"):
        host.Scene().setupScene(0, 17)
    with pytest.raises(host.UserError, match="SceneNotAvailable"):
        host.Camera.preset(22)


def test_synthetic_scene_matches_oracle(orc):
    from path_tracer_amd import host
    got = host.Scene().sceneSynthetic(2000, 1).spheres
    ref = orc.scene_synthetic(2000, 1)
    assert got.tobytes() == ref.tobytes()
    small = got[1:]
    assert got[0]["radius"] == 1000 and small["radius"].min() >= 0.05 and small["radius"].max() <= 0.3
    assert np.abs(small["cx"]).max() <= 50 and np.all(small["cy"] == small["radius"])


@pytest.mark.parametrize("n_scene", [18, 19, 20, 21])
def test_camera_preset_matches_oracle(orc, n_scene):
    from path_tracer_amd import host
    got = host.Camera.preset(n_scene).pod
    ref = orc.camera_preset(n_scene)
    assert bytes(got) == bytes(ref)


def test_camera_new_basis(orc):
    """camera.zig:48-97: orthonormal basis, lens radius = aperture/2, corner + spans reach the focus plane."""
    from path_tracer_amd import host
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, host.ASPECT_16_9, 0.1, 10.0).pod
    u, v = np.array(cam.u[:]), np.array(cam.v[:])
    assert abs(np.dot(u, v)) < 1e-6 and abs(np.linalg.norm(u) - 1) < 1e-6 and abs(np.linalg.norm(v) - 1) < 1e-6
    assert cam.lens_radius == np.float32(0.05)
    centre = np.array(cam.lower_left_corner[:]) + 0.5 * np.array(cam.horizontal[:]) + 0.5 * np.array(cam.vertical[:])
    assert abs(np.linalg.norm(centre - np.array([13, 2, 3])) - 10.0) < 1e-4
    ref = orc.Camera()
    orc.lib().orc_camera_new(orc.f3((13, 2, 3)), orc.f3((0, 0, 0)), orc.f3((0, 1, 0)), 20.0, host.ASPECT_16_9, 0.1, 10.0,
                             C.byref(ref))
    assert bytes(cam) == bytes(ref)


def test_image_kats():
    """render_targets/ppm_image.zig:106-175 on the host mirror."""
    from path_tracer_amd import host
    img = host.Image(640)
    assert img.px_size == 12 and img.height == 360 and img.header_size == 15
    assert img.header() == b"P3\n640 360\n255\n"
    assert host.Image(640, max_px_value=65535).px_size == 18
    assert img.rgbToAscii((0.5, 0.7, 1.0)) == b"  0   0   1\n"
    assert img.rgbToAscii((255, 10, 0)) == b"255  10   0\n"
    assert img.size == 15 + 12 * 640 * 360 and img.num_pixels == 640 * 360
    for w, h in ((256, 144), (400, 225), (1920, 1080), (3840, 2160)):
        assert host.Image(w).height == h


def test_utils_kats():
    from path_tracer_amd import host
    assert host.numDigits(123) == 3
    assert host.colNumToString(255, 42) == b" 42"
    with pytest.raises(host.UserError, match="ValueExceedsMaxValue"):
        host.colNumToString(128, 129)
    n = host.chunkSize(640 * 480, 4)
    assert 4 < n < 640 * 480


def test_filepath():
    from path_tracer_amd import host
    rt = host.RayTracerConfig(rays_per_subsample=150, rebounds=6)
    assert host.filepath(21, rt, host.Image(256), 456) == "images/scene21-w256-subpx4-spp150-depth6-seed456.ppm"


def test_bad_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any device work."""
    from path_tracer_amd import _lib
    L = _lib.load()
    assert L.ptgpu_create(0, None) == _lib.PT_ERR_BAD_ARG
    assert L.ptgpu_destroy(None) == _lib.PT_ERR_BAD_ARG
    assert L.ptgpu_set_scene(None, None, 0) == _lib.PT_ERR_BAD_ARG
    assert L.ptgpu_get_stats(None, None) == _lib.PT_ERR_BAD_ARG


def test_sphere_tree_builder_invariants():
    """The builder behind ptgpu_set_scene, on the host alone: every sphere sits in exactly one leaf, inside the boxes of all its
    ancestors (centre +- |radius|, the premise of the walk's conservativeness), leaves hold at most 4 spheres, depth within the
    traversal stack; median splits alone stay within the LDS capacity for every one-tile scene."""
    from path_tracer_amd import _lib, host
    L = _lib.load()

    def check(sp, median_only=0):
        nodes, depth, leaves, bad = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64(1)
        rc = L.ptgpu_debug_tree_check(sp.ctypes.data, len(sp), median_only, C.byref(nodes), C.byref(depth), C.byref(leaves), C.byref(bad))
        return rc, nodes.value, depth.value, leaves.value, bad.value

    rng = np.random.default_rng(3)
    scenes = [host.Scene().setupScene(456, k).spheres for k in (18, 19, 20, 21)] + [host.Scene().sceneSynthetic(n, 2).spheres for n in (1, 4, 5, 300, 3000, 100000)]
    for trial in range(30):
        n = int(rng.choice([2, 9, 77, 512, 2000]))
        sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
        spread = float(rng.choice([0.0, 1.0, 50.0]))
        sp["cx"], sp["cy"], sp["cz"] = 1.0e4 * (trial % 3) + rng.normal(0, spread, n), np.round(rng.normal(0, spread, n)), rng.normal(0, spread, n)
        sp["radius"] = rng.choice([1e-3, 0.2, 3.0, 5000.0, -0.7], n)
        scenes.append(sp)
    for sp in scenes:
        for median_only in (0, 1):
            rc, nodes, depth, leaves, bad = check(sp, median_only)
            assert rc == 0 and bad == 0
            assert depth <= 24 and leaves >= (len(sp) + 3) // 4 and (nodes == leaves - 1 or len(sp) <= 4)
            if median_only and len(sp) <= 512:
                assert nodes <= 127 and depth <= 8          # what the LDS copy (192 nodes, 13 levels) relies on
    bad_scene = scenes[7].copy()
    bad_scene["cy"][3] = np.inf
    assert check(bad_scene)[0] == _lib.PT_ERR_UNSUPPORTED
