"""PT_SAMPLER_DIRECT on the GPU (include/ptgpu.h; NOT in the reference): the direct samplers are the oracle's, operation by
operation, so in that mode too the GPU's float32 sums EQUAL the oracle's — per function on the committed vectors, and for
whole frames through every evaluator (scan, sphere tree, primary pass on and off).  That the mode draws from the reference's
distributions is shown on the CPU (tests/test_sampler_direct.py); here the GPU's own 4096-path frames in both modes agree
to the north star's 40 dB."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_direct.npz")


def test_sampler_functions_match_the_committed_vectors(gpu):
    from path_tracer_amd import _lib
    g = np.load(GOLD)
    for which, fn in (("disk", _lib.PT_FN_SAMPLE_DISK), ("ball", _lib.PT_FN_SAMPLE_BALL)):
        for name in ("rejection", "direct"):
            got = gpu.debug_eval(fn, g["sample_%s_%s_in" % (which, name)])
            np.testing.assert_array_equal(got, g["sample_%s_%s_out" % (which, name)])


def test_sampler_functions_match_the_oracle_on_fresh_draws(orc, gpu):
    from path_tracer_amd import _lib
    rng = np.random.default_rng(99)
    xi = rng.integers(0, 1 << 23, size=(50_000, 16)).astype(np.float32) / np.float32(1 << 23)
    for which, fn in (("disk", _lib.PT_FN_SAMPLE_DISK), ("ball", _lib.PT_FN_SAMPLE_BALL)):
        for smp in (0, 1):
            inp = np.concatenate([np.full((len(xi), 1), smp, np.float32), xi], axis=1)
            np.testing.assert_array_equal(gpu.debug_eval(fn, inp), orc.sample_batch(which, smp, xi))


def _both(orc, gpu, spheres, cam, img, **kw):
    cfg = orc.make_config(img, sampler=orc.SAMPLER_DIRECT, **kw)
    rc, ref, st = orc.render_sum(spheres, cam, cfg, nthreads=16)
    assert rc == 0
    got = gpu.render_sum(cfg)
    gst = gpu.stats()
    assert (gst.paths, gst.rays) == (st.paths, st.rays)
    np.testing.assert_array_equal(got, ref)
    return got


@pytest.mark.parametrize("n_scene,width,spp,rebounds,seed", [(18, 40, 3, 50, 42), (21, 64, 2, 50, 456), (21, 48, 3, 6, 7), (20, 48, 2, 50, 1)])
def test_direct_mode_frames_bit_exact(orc, gpu, n_scene, width, spp, rebounds, seed):
    from path_tracer_amd import _lib
    spheres, _ = orc.setup_scene(n_scene, 456)
    cam = orc.camera_preset(n_scene)
    img = orc.image_new(width)
    gpu.set_camera(cam)
    frames = []
    try:
        for tree_min, primary in ((_lib.PT_TREE_MIN_SPHERES_DEFAULT, _lib.PT_PRIMARY_BYTES_DEFAULT), (0, _lib.PT_PRIMARY_BYTES_DEFAULT),
                                  (_lib.PT_TREE_MIN_SPHERES_DEFAULT, 0), (0, 0), (1, _lib.PT_PRIMARY_BYTES_DEFAULT)):
            gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, tree_min)
            gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, primary)
            gpu.set_scene(spheres)
            frames.append(_both(orc, gpu, spheres, cam, img, subpixels=4, rays_per_subsample=spp, rebounds=rebounds, seed=seed, chunk=4))
    finally:
        gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, _lib.PT_TREE_MIN_SPHERES_DEFAULT)
        gpu.set_limit(_lib.PT_LIMIT_PRIMARY_BYTES, _lib.PT_PRIMARY_BYTES_DEFAULT)


def test_direct_mode_committed_frames_ranges_and_multi_tile_scene(orc, gpu):
    g = np.load(GOLD)
    for n_scene, width in ((18, 32), (21, 64)):
        spheres, _ = orc.setup_scene(n_scene, 456)
        cam = orc.camera_preset(n_scene)
        gpu.set_scene(spheres)
        gpu.set_camera(cam)
        cfg = orc.make_config(orc.image_new(width), subpixels=4, rays_per_subsample=2, rebounds=50, seed=456, sampler=1)
        np.testing.assert_array_equal(gpu.render_sum(cfg), g["direct_scene%d_w%d_d50_seed456" % (n_scene, width)])
        st = gpu.stats()
        assert [st.paths, st.rays] == list(g["direct_scene%d_w%d_d50_seed456_rays" % (n_scene, width)])
    # pixel range + sample range + chunks + tiles (what the multi-GPU split uses), direct mode
    spheres, _ = orc.setup_scene(21, 456)
    cam = orc.camera_preset(21)
    img = orc.image_new(64)
    _both(orc, gpu, spheres, cam, img, subpixels=4, rays_per_subsample=5, rebounds=50, seed=9, idx_start=128, idx_stop=64 * 36 - 11,
          sample_start=3, sample_stop=17, chunk=4, tile_pixels=64, tile_count=3, tile_index=1)
    # a scene of several LDS tiles: sphere tree in global memory, lists built by the tree walk
    big = orc.scene_synthetic(3000, 1)
    gpu.set_scene(big)
    _both(orc, gpu, big, cam, orc.image_new(48), subpixels=4, rays_per_subsample=2, rebounds=50, seed=3, chunk=4)
    gpu.set_scene(spheres)


def test_noalloc_entry_point_in_direct_mode(gpu):
    import torch
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(96)
    cfg = host.make_pt_config(img, host.RayTracerConfig(subpixels=4, rays_per_subsample=3, rebounds=50), seed=11, chunk=4, sampler=1)
    a = torch.zeros(img.height * img.width * 3, dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    gpu.render_sum_device(cfg, a.data_ptr())
    gpu.render_sum_device_noalloc(cfg, b.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(a, b) and float(a.abs().sum()) > 0


def test_direct_and_rejection_frames_converge_to_one_image(gpu):
    """North star's tolerance (>= 40 dB at 4096 paths per pixel) between the GPU's frame in direct mode and its frame with the
    reference's rejection loops (another seed): two estimates of one image; and the value rises with the sample count."""
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(96)
    psnr = {}
    for rps in (64, 1024):
        rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=rps, rebounds=50)
        rej = gpu.render_mean(host.make_pt_config(img, rt, seed=1, chunk=16)).astype(np.float64)
        rej2 = gpu.render_mean(host.make_pt_config(img, rt, seed=3, chunk=16)).astype(np.float64)
        dire = gpu.render_mean(host.make_pt_config(img, rt, seed=2, chunk=16, sampler=1)).astype(np.float64)
        psnr[rps] = (10.0 * np.log10(1.0 / float(np.mean((rej - dire) ** 2))), 10.0 * np.log10(1.0 / float(np.mean((rej - rej2) ** 2))))
        assert abs(psnr[rps][0] - psnr[rps][1]) < 1.0, psnr   # direct vs rejection looks like rejection vs rejection
    assert psnr[1024][0] >= 40.0, psnr
    assert psnr[1024][0] - psnr[64][0] > 10.0, psnr             # 12 dB per 16x samples, within noise


def test_direct_mode_converges_to_the_reference_sampler(orc, gpu):
    """North star N4 in direct mode: the GPU's 4096-path frame drawn with PT_SAMPLER_DIRECT against the REFERENCE'S OWN sampler
    (orc_render_sum_zig: main.zig:166-177's sequential std.rand streams and rejection loops): >= 40 dB at 1024 reference paths,
    rising with the reference's sample count as two unbiased estimates of one image do."""
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(48)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=1024, rebounds=50)
    g = gpu.render_mean(host.make_pt_config(img, rt, seed=1, chunk=16, sampler=1)).astype(np.float64)
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam.pod), C.sizeof(ocam))
    oimg = orc.image_new(48)
    psnr = {}
    for rps in (64, 256):
        spheres, stream = orc.setup_scene(21, 456)
        ocfg = orc.make_config(oimg, subpixels=4, rays_per_subsample=rps, rebounds=50, mode=2)
        rc, s, st = orc.render_sum_zig(spheres, ocam, ocfg, stream)
        assert rc == 0
        z = orc.resolve_mean(ocfg, s).astype(np.float64)
        psnr[4 * rps] = 10.0 * np.log10(1.0 / float(np.mean((g - z) ** 2)))
    assert psnr[1024] >= 40.0, psnr
    assert psnr[1024] >= psnr[256] + 3.0, psnr
