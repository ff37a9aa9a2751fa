"""The primary pass (pt_primary_lists + pt_primary) against the megakernel's own camera rays.

With PT_LIMIT_PRIMARY_BYTES = 0 every camera ray is cast inside the megakernel and takes the scan over ALL spheres
(the reference's loop, scene.zig:96-110); with the primary pass it meets only the spheres on its pixel's candidate
list.  The two must agree bit for bit — frames, path and ray counts — which is exactly the statement "the closest hit
of every camera ray is on its pixel's list" (a sphere missing from a list changes a pixel only if it was that ray's
closest hit; any other omission is harmless by definition).  Every other GPU parity test runs with the primary pass on
and compares with the oracle, which has no such lists.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(gpu, cfg, expect_primary=True):
    """render with the primary pass off, with camera rays only, and with the second bounce too; frames and counters must
    be identical"""
    from path_tracer_amd import _lib
    try:
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BOUNCES, 2)
        on = gpu.render_sum(cfg, allow_not_unit=True)
        st_on = gpu.stats()
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BOUNCES, 1)
        on1 = gpu.render_sum(cfg, allow_not_unit=True)
        st_on1 = gpu.stats()
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 0)
        off = gpu.render_sum(cfg, allow_not_unit=True)
        st_off = gpu.stats()
    finally:
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BOUNCES, 2)
    assert st_off.primary_rays == 0 and st_off.list2_rays == 0 and st_on1.list2_rays == 0
    assert st_on1.primary_rays == st_on.primary_rays
    if expect_primary:
        assert st_on.primary_rays > 0
    for st in (st_on, st_on1):
        assert (st.paths, st.rays, st.not_unit) == (st_off.paths, st_off.rays, st_off.not_unit)
    np.testing.assert_array_equal(on1, off)
    np.testing.assert_array_equal(on, off)
    return on, st_on


@pytest.mark.parametrize("n_scene", [18, 19, 20, 21])
@pytest.mark.parametrize("deterministic", [False, True])
def test_reference_scenes_on_off(gpu, n_scene, deterministic):
    """The four scenes of scene.zig with their own cameras (main.zig:107-136: apertures 0.2 / 0.2 / 2.0 / 0.1)."""
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, n_scene)
    cam = host.Camera.preset(n_scene)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(160)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=8, rebounds=50)
    _, st = _both(gpu, host.make_pt_config(img, rt, seed=3, chunk=5, deterministic=deterministic))
    assert st.primary_rays == st.paths            # every pixel of these frames has a list: no camera ray takes the scan
    if not deterministic or n_scene == 21:
        assert st.list2_rays > 0


def test_ranges_tiles_samples_and_aovs_on_off(gpu):
    """Sub-ranges, a tile split with a cut last tile, a sample sub-range, one chunk (sums straight into the frame),
    first-hit colour functions, rebounds 0 and 1."""
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(100)   # 100 x 56
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=6, rebounds=50)
    for kw in (dict(chunk=0), dict(chunk=5, istart=777, istop=4001), dict(chunk=4, tile_pixels=256, tile_count=3, tile_index=2),
               dict(chunk=3, sample_start=2, sample_stop=11), dict(chunk=1), dict(chunk=4, illumination=1), dict(chunk=4, illumination=2)):
        _both(gpu, host.make_pt_config(img, rt, seed=9, **kw))
    for rebounds in (0, 1, 2):
        _both(gpu, host.make_pt_config(img, host.RayTracerConfig(subpixels=2, rays_per_subsample=6, rebounds=rebounds), seed=9, chunk=4))
    # every chunk size that gets per-item masks (it divides 64 and is <= 32), and some that do not; 70 samples: a ragged last chunk
    rt70 = host.RayTracerConfig(subpixels=2, rays_per_subsample=35, rebounds=50)
    for chunk in (1, 2, 8, 16, 32, 64, 7, 70, 33):
        _both(gpu, host.make_pt_config(img, rt70, seed=10, chunk=chunk, istart=500, istop=2500))


def test_segments_of_the_record_buffer_are_bit_identical(gpu):
    """PT_LIMIT_PRIMARY_BYTES bounds records + lists: a small limit cuts the render into segments, a limit too small for
    one wave of pixels switches the pass off for that render; neither changes a bit."""
    from path_tracer_amd import _lib, host
    scene = host.Scene().setupScene(456, 21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(host.Camera.preset(21).pod)
    img = host.Image(200)   # 22 400 pixels
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=8, rebounds=50)      # 32 samples: 1 KiB of records per pixel
    cfg = host.make_pt_config(img, rt, seed=5, chunk=8)
    one = gpu.render_sum(cfg)
    assert gpu.stats().segments == 1
    try:
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 5000 * (32 * 32 + 64))
        many = gpu.render_sum(cfg)
        st = gpu.stats()
        assert 5 <= st.segments <= 7 and st.primary_rays == st.paths
        np.testing.assert_array_equal(many, one)
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 4096)
        none = gpu.render_sum(cfg)
        st = gpu.stats()
        assert st.segments == 1 and st.primary_rays == 0
        np.testing.assert_array_equal(none, one)
    finally:
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)


def _random_camera(rng, host, aspect):
    lookfrom = rng.normal(0, 6, 3)
    lookat = lookfrom + rng.normal(0, 1, 3) * rng.choice([0.5, 3.0, 20.0])
    vup = np.array([0.0, 1.0, 0.0]) + rng.normal(0, 0.3, 3)
    vfov = float(rng.choice([5.0, 20.0, 60.0, 120.0]))
    aperture = float(rng.choice([0.0, 0.0, 0.05, 0.5, 4.0]))
    focus = float(rng.choice([0.3, 3.0, 10.0, 40.0]))
    return host.Camera.new(lookfrom, lookat, vup, vfov, aspect, aperture, focus)


def test_random_cameras_and_adversarial_scenes_on_off(gpu):
    """Pinhole to very wide lenses, focal planes in front of / inside / behind the spheres, narrow and wide fields
    of view, cameras inside spheres; radii from 1e-3 to 5000 and negative; far-away, coincident and overlapping
    spheres; dense scenes whose lists overflow (those pixels' camera rays take the scan)."""
    from path_tracer_amd import _lib, host
    rng = np.random.default_rng(11)
    img = host.Image(64)    # 64 x 36
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=16, t_min=0.001, t_max=1.0e6, rebounds=8)
    listed = second = total = 0
    for trial in range(96):
        n = int(rng.choice([3, 40, 300, 700]))
        sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
        spread = float(rng.choice([4.0, 15.0, 60.0]))
        sp["cx"], sp["cy"], sp["cz"] = rng.normal(0, spread, n), rng.normal(0, spread, n), rng.normal(0, spread, n)
        sp["radius"] = rng.choice([1e-3, 0.02, 0.2, 0.5, 3.0, 40.0, -0.7], n)
        sp["kind"] = rng.choice([0, 1, 2], n, p=[0.6, 0.25, 0.15])
        sp["ar"], sp["ag"], sp["ab"] = rng.random(n), rng.random(n), rng.random(n)
        sp["param"] = np.where(sp["kind"] == 2, 1.5, 0.3 * rng.random(n))
        sp[0] = (0, -5000, 0, 5000, 0, 0.5, 0.5, 0.5, 0)
        if n > 3:
            sp[1] = (2000, 0, 0, 1.0, 0, 1, 1, 1, 0)
            sp[2] = sp[3] = (1, 1, 1, 0.25, 0, 1, 0.5, 0.2, 0)     # coincident: the lower index must win
        cam = _random_camera(rng, host, img.aspect_ratio)
        gpu.set_scene(sp)
        gpu.set_camera(cam.pod)
        _, st = _both(gpu, host.make_pt_config(img, rt, seed=100 + trial, chunk=16), expect_primary=False)
        listed += st.primary_rays
        second += st.list2_rays
        total += st.paths
    assert 0.5 * total < listed < total     # both kinds of pixel were exercised: listed ones and overflowing ones
    assert second > 0.02 * total            # and second-bounce lists were used


def test_no_list_misses_a_sphere_the_exact_test_accepts(gpu):
    """The property the candidate lists rest on, sphere by sphere (ptgpu_debug_list_safety) instead of through frames, where a missing
    sphere only shows when it was the closest: pt_primary's own rays — its jitter, lens draws and scattered directions, under exactly the
    conditions it uses a list — also get the reference's exact test on EVERY sphere; each accepted sphere must be in the list, and the
    closest hit must be the full scan's.  Reference scenes, random cameras over adversarial scenes (as above), a many-tile scene whose
    lists are built by the tree walk."""
    from path_tracer_amd import _lib, host
    rng = np.random.default_rng(12)
    total = dict(accepted_a=0, missing_a=0, accepted_b=0, missing_b=0, closest_differs=0, rays_a=0, rays_b=0)

    def run(cfg, may_have_no_list=False):
        try:
            r = gpu.debug_list_safety(cfg)
        except _lib.PtGpuError as e:             # PT_ERR_UNSUPPORTED: no pixel of this camera got a list (every list overflowed)
            assert may_have_no_list and e.status == _lib.PT_ERR_UNSUPPORTED, e
            return
        assert r["missing_a"] == 0 and r["missing_b"] == 0 and r["closest_differs"] == 0, r
        for k in total:
            total[k] += r[k]

    for n_scene, width in ((18, 96), (19, 96), (20, 96), (21, 160)):
        gpu.set_scene(host.Scene().setupScene(456, n_scene).spheres)
        gpu.set_camera(host.Camera.preset(n_scene).pod)
        run(host.make_pt_config(host.Image(width), host.RayTracerConfig(subpixels=4, rays_per_subsample=8, rebounds=50), seed=3, chunk=16))
    img = host.Image(64)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=8, t_min=0.001, t_max=1.0e6, rebounds=8)
    for trial in range(48):
        n = int(rng.choice([3, 40, 300, 700]))
        sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
        spread = float(rng.choice([4.0, 15.0, 60.0]))
        sp["cx"], sp["cy"], sp["cz"] = rng.normal(0, spread, n), rng.normal(0, spread, n), rng.normal(0, spread, n)
        sp["radius"] = rng.choice([1e-3, 0.02, 0.2, 0.5, 3.0, 40.0, -0.7], n)
        sp["kind"] = rng.choice([0, 1, 2], n, p=[0.6, 0.25, 0.15])
        sp["ar"], sp["ag"], sp["ab"] = rng.random(n), rng.random(n), rng.random(n)
        sp["param"] = np.where(sp["kind"] == 2, 1.5, 0.3 * rng.random(n))
        sp[0] = (0, -5000, 0, 5000, 0, 0.5, 0.5, 0.5, 0)
        if n > 3:
            sp[1] = (2000, 0, 0, 1.0, 0, 1, 1, 1, 0)
            sp[2] = sp[3] = (1, 1, 1, 0.25, 0, 1, 0.5, 0.2, 0)
        gpu.set_scene(sp)
        gpu.set_camera(_random_camera(rng, host, img.aspect_ratio).pod)
        run(host.make_pt_config(img, rt, seed=300 + trial, chunk=16), may_have_no_list=True)
    gpu.set_scene(host.Scene().sceneSynthetic(3000, 7).spheres)         # lists by the tree walk (pt_primary_lists<true>), in walk order
    gpu.set_camera(host.Camera.preset(21).pod)
    run(host.make_pt_config(host.Image(96), host.RayTracerConfig(subpixels=2, rays_per_subsample=4, rebounds=20), seed=4, chunk=4))
    assert total["rays_a"] > 3_000_000 and total["rays_b"] > 300_000, total
    assert total["accepted_a"] > 1_000_000 and total["accepted_b"] > 100_000, total


def test_many_tile_scene_on_off(gpu):
    """More spheres than one LDS tile: the primary pass reads the exact geometry from global memory."""
    from path_tracer_amd import host
    scene = host.Scene().sceneSynthetic(3000, 7)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(host.Camera.preset(21).pod)
    img = host.Image(96)
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=4, rebounds=20)
    _, st = _both(gpu, host.make_pt_config(img, rt, seed=4, chunk=4))
    assert 0 < st.primary_rays <= st.paths


def test_negative_t_min_keeps_camera_rays_in_the_megakernel(orc, gpu):
    """The lists describe rays that leave the lens forwards; with t_min < 0 a ray may also hit behind its origin, so the
    pass is not used — and the result is still the oracle's."""
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 19)
    cam = host.Camera.preset(19)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(48)
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=2, t_min=-5.0, t_max=1000.0, rebounds=6)
    cfg = host.make_pt_config(img, rt, seed=2)
    got = gpu.render_sum(cfg)
    assert gpu.stats().primary_rays == 0
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam.pod), C.sizeof(ocam))
    rc, ref, _ = orc.render_sum(scene.spheres, ocam, ocfg, nthreads=4)
    assert rc == 0
    np.testing.assert_array_equal(got, ref)


def test_out_of_memory_falls_back_to_smaller_segments():
    """When the device cannot give the primary pass the buffer its limit allows, the render takes half of it (and so on,
    finally none) instead of failing: with all but ~6 GiB of the HBM held by someone else, a render whose records
    need 10.6 GB still completes, in more segments, with the same bits.  (A context of its own: the shared one may
    already own a buffer of that size from an earlier test.)"""
    import torch
    from path_tracer_amd import _lib, host
    from path_tracer_amd.gpu import GpuContext
    scene = host.Scene().setupScene(456, 21)
    img = host.Image(1920)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=40, rebounds=50)      # 160 samples: 5 KiB of records per pixel
    cfg = host.make_pt_config(img, rt, seed=5, chunk=16)
    frame = torch.zeros(img.num_pixels * 3, dtype=torch.float32, device="cuda")
    hog = None
    with GpuContext(0) as gpu:
        gpu.set_scene(scene.spheres)
        gpu.set_camera(host.Camera.preset(21).pod)
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 2 << 30)
        gpu.render_sum_device(cfg, frame.data_ptr())
        torch.cuda.synchronize()
        assert gpu.stats().segments == 6
        want = frame.clone()
        try:
            gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 1 << 40)
            torch.cuda.empty_cache()
            free, _ = torch.cuda.mem_get_info()
            hog = torch.empty(max(free - (6 << 30), 0), dtype=torch.uint8, device="cuda")
            frame.zero_()
            gpu.render_sum_device(cfg, frame.data_ptr())
            torch.cuda.synchronize()
            st = gpu.stats()
            assert 2 <= st.segments < 6 and st.primary_rays == st.paths
            assert torch.equal(frame, want)
        finally:
            del hog
            torch.cuda.empty_cache()
