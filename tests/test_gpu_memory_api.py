"""Device memory a context holds (include/ptgpu.h: ptgpu_work_bytes, ptgpu_trim, PT_LIMIT_PRIMARY_BYTES) and argument checks
that must come before any allocation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(gpu):
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(host.Camera.preset(21).pod)
    return host


def test_work_buffers_grow_are_kept_trimmed_and_follow_the_limit(gpu):
    from path_tracer_amd import _lib
    host = _setup(gpu)
    img = host.Image(320)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=8, rebounds=50)
    cfg = host.make_pt_config(img, rt, seed=3, chunk=16)
    gpu.trim()
    assert gpu.work_bytes() == 0
    a = gpu.render_sum(cfg)
    held = gpu.work_bytes()
    records = img.num_pixels * 32 * 32                      # 32 B per path of the launch
    assert held >= records and gpu.stats().segments == 1 and gpu.stats().primary_rays > 0
    b = gpu.render_sum(cfg)
    assert gpu.work_bytes() == held                         # kept for the next render, not grown
    np.testing.assert_array_equal(a, b)
    try:
        # a lowered limit gives the record buffer back at once, and the next render is cut into segments: same bits
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, records // 3)
        assert gpu.work_bytes() <= held - records
        c = gpu.render_sum(cfg)
        assert gpu.stats().segments >= 3 and gpu.work_bytes() < held
        np.testing.assert_array_equal(a, c)
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, 0)        # no primary pass at all
        d = gpu.render_sum(cfg)
        assert gpu.stats().primary_rays == 0
        np.testing.assert_array_equal(a, d)
    finally:
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)
    gpu.trim()
    assert gpu.work_bytes() == 0
    np.testing.assert_array_equal(gpu.render_sum(cfg), a)   # everything is allocated again on demand


def test_automatic_limit_leaves_room_when_the_device_is_nearly_full(gpu):
    """The default PT_LIMIT_PRIMARY_BYTES is half of what hipMemGetInfo reports free: with most of the device taken by somebody
    else the render still succeeds — in more segments or without the pass — and the frame is the same."""
    import torch
    host = _setup(gpu)
    img = host.Image(640)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=16, rebounds=50)
    cfg = host.make_pt_config(img, rt, seed=9, chunk=16)
    gpu.trim()
    want = gpu.render_sum(cfg)
    assert gpu.stats().segments == 1
    gpu.trim()
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info()
    need = img.num_pixels * 64 * 32
    hog = torch.empty(max(free - need // 2, 1 << 20), dtype=torch.uint8, device="cuda")   # leave a quarter of what one segment needs
    try:
        got = gpu.render_sum(cfg)
        st = gpu.stats()
        assert st.segments > 1 or st.primary_rays == 0
        np.testing.assert_array_equal(got, want)
    finally:
        del hog
        torch.cuda.empty_cache()
        gpu.trim()


def test_bad_configs_are_refused_before_anything_is_allocated(gpu):
    from path_tracer_amd import _lib
    host = _setup(gpu)
    gpu.trim()
    img = host.Image(64)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=1, rebounds=6)
    import ctypes as C
    L, h = gpu._L, gpu._h
    tiny = np.zeros(16, dtype=np.float32)                   # never touched: the library must refuse before it reads or writes anything
    buf = np.zeros(img.size, dtype=np.uint8)
    for mutate in (lambda c: setattr(c, "width", 1 << 30), lambda c: setattr(c, "height", 0), lambda c: setattr(c, "sampler", 2),
                   lambda c: setattr(c, "mode", 7), lambda c: setattr(c, "idx_stop", 64 * 36 + 1)):
        cfg = host.make_pt_config(img, rt, seed=1)
        mutate(cfg)
        assert L.ptgpu_render_sum_host(h, C.byref(cfg), tiny.ctypes.data) == _lib.PT_ERR_BAD_ARG
        assert L.ptgpu_render_mean_host(h, C.byref(cfg), tiny.ctypes.data) == _lib.PT_ERR_BAD_ARG
        rc = L.ptgpu_render_ppm(h, C.byref(cfg), buf.ctypes.data, buf.size, img.header_size, img.px_size)
        assert rc in (_lib.PT_ERR_BAD_ARG, _lib.PT_ERR_NO_SPACE), rc
    assert not tiny.any() and not buf.any()
    assert gpu.work_bytes() == 0                            # a bogus width never reached hipMalloc
