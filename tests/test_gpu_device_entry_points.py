"""The device-pointer entry points of the C ABI — the exact calls bench.py and the multi-GPU path make:
ptgpu_render_sum_device on a torch tensor with a NON-default stream and ptgpu_resolve_device (records and
d_rgb8 outputs) — against the host-pointer wrappers and the oracle; plus the bounded chunk-partial buffer
(PT_LIMIT_PARTIAL_BYTES: segmented renders are bit-identical) and tile splits whose last tile is cut.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(gpu, n_scene, seed=456):
    from path_tracer_amd import host
    scene = host.Scene().setupScene(seed, n_scene)
    cam = host.Camera.preset(n_scene)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    return host, scene, cam


def _oracle(orc, spheres, cam_pod, cfg, nthreads=16):
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam_pod), C.sizeof(ocam))
    rc, out, st = orc.render_sum(spheres, ocam, ocfg, nthreads=nthreads)
    assert rc == 0
    return out, st, ocfg


def test_render_sum_device_on_side_stream_and_resolve_device(orc, gpu):
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(160)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=8, rebounds=50)
    cfg = host.make_pt_config(img, rt, seed=456, chunk=4)
    want = gpu.render_sum(cfg)                                   # host-pointer wrapper
    ref, st, ocfg = _oracle(orc, scene.spheres, cam.pod, cfg)
    np.testing.assert_array_equal(want, ref)

    side = torch.cuda.Stream()
    frame = torch.full((img.height * img.width * 3,), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        assert torch.cuda.current_stream().cuda_stream == side.cuda_stream != 0
        gpu.render_sum_device(cfg, frame.data_ptr(), side.cuda_stream)
        gst = gpu.stats()
        assert gst.paths == st.paths and gst.rays == st.rays and gst.segments == 1
        # the resolve on the same stream: 12-byte ASCII records and packed u8 triples
        rec = torch.zeros(img.num_pixels * 12, dtype=torch.uint8, device="cuda")
        rgb8 = torch.zeros(img.num_pixels * 3, dtype=torch.uint8, device="cuda")
        gpu.resolve_device(cfg, frame.data_ptr(), rec.data_ptr(), rgb8.data_ptr(), side.cuda_stream)
        gpu.resolve_device(cfg, frame.data_ptr(), 0, rgb8.data_ptr(), side.cuda_stream)   # either output may be NULL
        gpu.resolve_device(cfg, frame.data_ptr(), rec.data_ptr(), 0, side.cuda_stream)
    side.synchronize()
    got = frame.cpu().numpy().reshape(img.height, img.width, 3)
    np.testing.assert_array_equal(got, want)

    rc, ref_ppm = orc.resolve_ppm(ocfg, orc.image_new(img.width), ref)
    assert rc == 0
    np.testing.assert_array_equal(rec.cpu().numpy(), ref_ppm[img.header_size:])
    buf = np.zeros(img.size, dtype=np.uint8)
    gpu.render_ppm(cfg, buf, img.header_size, img.px_size)      # the host wrapper writes the same bytes
    np.testing.assert_array_equal(buf[img.header_size:], ref_ppm[img.header_size:])
    # d_rgb8 = the numbers in the records
    txt = bytes(ref_ppm[img.header_size:]).decode().split()
    np.testing.assert_array_equal(rgb8.cpu().numpy(), np.array(txt, dtype=np.int64).astype(np.uint8))


def test_resolve_device_sub_range_offsets(orc, gpu):
    """Records of a sub-range land at (idx - idx_start) * 12 of d_records (include/ptgpu.h)."""
    host, scene, cam = _setup(gpu, 18, 1)
    img = host.Image(64)
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=2, rebounds=6)
    full = host.make_pt_config(img, rt, seed=3)
    frame = torch.zeros(img.num_pixels * 3, dtype=torch.float32, device="cuda")
    gpu.render_sum_device(full, frame.data_ptr(), 0)
    rec_all = torch.zeros(img.num_pixels * 12, dtype=torch.uint8, device="cuda")
    gpu.resolve_device(full, frame.data_ptr(), rec_all.data_ptr(), 0, 0)
    a, b = 5 * 64 + 7, 11 * 64 + 3
    sub = host.make_pt_config(img, rt, seed=3, istart=a, istop=b)
    rec = torch.zeros((b - a) * 12, dtype=torch.uint8, device="cuda")
    gpu.resolve_device(sub, frame.data_ptr(), rec.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rec.cpu().numpy(), rec_all.cpu().numpy()[a * 12:b * 12])


@pytest.mark.parametrize("chunk", [16, 64])
def test_partial_buffer_limit_segments_are_bit_identical(gpu, chunk):
    """PT_LIMIT_PARTIAL_BYTES bounds the chunk-partial buffer: the render is cut into segments (one launch each)
    and must not change a bit — full range, a sub-range, and a tile split with a cut last tile."""
    from path_tracer_amd import _lib
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(200)                       # 200 x 112 = 22 400 pixels
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=32, rebounds=50)   # 128 samples: 8 / 2 chunks
    cfgs = [host.make_pt_config(img, rt, seed=5, chunk=chunk),
            host.make_pt_config(img, rt, seed=5, chunk=chunk, istart=1000, istop=20001),
            host.make_pt_config(img, rt, seed=5, chunk=chunk, tile_pixels=256, tile_count=3, tile_index=1)]
    try:
        for cfg in cfgs:
            gpu.set_limit(_lib.PT_LIMIT_PARTIAL_BYTES, 2 << 30)
            one = gpu.render_sum(cfg)
            assert gpu.stats().segments == 1
            nchunks = 128 // chunk
            gpu.set_limit(_lib.PT_LIMIT_PARTIAL_BYTES, 3000 * nchunks * 12)       # ~3000 pixel slots per launch
            many = gpu.render_sum(cfg)
            st = gpu.stats()
            assert st.segments >= 3 and st.paths == (one != 0).any(axis=2).sum() * 128
            np.testing.assert_array_equal(many, one)
    finally:
        gpu.set_limit(_lib.PT_LIMIT_PARTIAL_BYTES, 2 << 30)


@pytest.mark.parametrize("width,tile_pixels,owners", [(100, 256, 3), (100, 64, 5), (33, 256, 2), (40, 4096, 2)])
def test_tile_split_with_cut_last_tile(orc, gpu, width, tile_pixels, owners):
    """(idx_stop - idx_start) % tile_pixels != 0: the last tile is cut by the range (Image(100) = 5600 pixels with
    256-pixel tiles; a range shorter than one tile).  Round 1's kernel never finished such a launch.  The owners'
    frames are disjoint and their union is the single-GPU frame, bit for bit — and the oracle's."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(width)
    assert img.num_pixels % tile_pixels != 0
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=3, rebounds=50)
    whole = gpu.render_sum(host.make_pt_config(img, rt, seed=11, chunk=2))
    ref, st, _ = _oracle(orc, scene.spheres, cam.pod, host.make_pt_config(img, rt, seed=11, chunk=2))
    np.testing.assert_array_equal(whole, ref)
    acc = np.zeros_like(whole)
    paths = 0
    for g in range(owners):
        cfg = host.make_pt_config(img, rt, seed=11, chunk=2, tile_pixels=tile_pixels, tile_count=owners, tile_index=g)
        part = gpu.render_sum(cfg)
        paths += gpu.stats().paths
        opart, _, _ = _oracle(orc, scene.spheres, cam.pod, cfg)
        np.testing.assert_array_equal(part, opart)
        assert not ((acc != 0) & (part != 0)).any()
        acc += part
    assert paths == st.paths == img.num_pixels * 6
    np.testing.assert_array_equal(acc, whole)


def test_colour_just_above_max_is_reported(orc, gpu):
    """A colour in (255, 256) truncates to 255 but violates the reference's assert col <= 255
    (renderers/ppm_image.zig:76): written as 255 AND reported, by the GPU resolve and by the oracle alike."""
    from path_tracer_amd import _lib
    from path_tracer_amd._lib import PtGpuError
    host, scene, cam = _setup(gpu, 18, 0)
    img = host.Image(32)
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=1)
    cfg = host.make_pt_config(img, rt, deterministic=True)
    for c in range(3):
        cfg.blend_start[c] = cfg.blend_stop[c] = 1.002       # sky = 1.002 -> 255.51
    buf = np.zeros(img.size, dtype=np.uint8)
    with pytest.raises(PtGpuError) as e:
        gpu.render_ppm(cfg, buf, img.header_size, img.px_size)
    assert e.value.status == _lib.PT_ERR_VALUE_EXCEEDS_MAX
    assert bytes(buf[img.header_size:img.header_size + 12]) == b"255 255 255\n"
    ref, _, ocfg = _oracle(orc, scene.spheres, cam.pod, cfg)
    rc, ref_ppm = orc.resolve_ppm(ocfg, orc.image_new(32), ref)
    assert rc == -6
    np.testing.assert_array_equal(buf[img.header_size:], ref_ppm[img.header_size:])
    for c in range(3):
        cfg.blend_start[c] = cfg.blend_stop[c] = 1.0         # exactly 255.0 is fine
    gpu.render_ppm(cfg, buf, img.header_size, img.px_size)
