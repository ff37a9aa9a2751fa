"""The BASELINE.json configurations on the GPU.

configs[0] and configs[1] are small enough to compare the whole frame with the oracle, bit for bit.
configs[2..4] run at (or near) full size and are checked through size-independent properties —
pixel spot checks against the oracle at the full sample count, run-to-run determinism, exact
tile-split partition, sample-range additivity — plus a many-tile (N > one LDS tile) scene against the
oracle, and the north star's convergence criterion (PSNR >= 40 dB at 4096 spp between independent
estimates).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(gpu, n_scene, seed=456):
    from path_tracer_amd import host
    scene = host.Scene().setupScene(seed, n_scene)
    cam = host.Camera.preset(n_scene)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    return host, scene, cam


def _oracle_sum(orc, spheres, cam_pod, cfg, nthreads=16):
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam_pod), C.sizeof(ocam))
    rc, out, st = orc.render_sum(spheres, ocam, ocfg, nthreads=nthreads)
    assert rc == 0
    return out, st


def test_config0_readme_demo_full_frame(orc, gpu):
    """configs[0]: scene 21, 256x144, 150 spp (x4 subpixels = 600 paths/pixel), depth 6, seed 456."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(256)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=150, rebounds=6)
    cfg = host.make_pt_config(img, rt, seed=456)
    got = gpu.render_sum(cfg)
    ref, st = _oracle_sum(orc, scene.spheres, cam.pod, cfg)
    gst = gpu.stats()
    assert gst.paths == st.paths == 256 * 144 * 600 and gst.rays == st.rays
    np.testing.assert_array_equal(got, ref)


def test_config1_three_materials_full_frame(orc, gpu):
    """configs[1]: the 5-sphere Lambertian/Metal/Dielectric scene, 400x225, 100 spp, depth 50."""
    host, scene, cam = _setup(gpu, 18, 42)
    img = host.Image(400)
    assert img.height == 225
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=25, rebounds=50)
    cfg = host.make_pt_config(img, rt, seed=42)
    got = gpu.render_sum(cfg)
    ref, st = _oracle_sum(orc, scene.spheres, cam.pod, cfg)
    assert gpu.stats().rays == st.rays
    np.testing.assert_array_equal(got, ref)
    # and the bytes render() would have written
    buf = np.zeros(img.size, dtype=np.uint8)
    gpu.render_ppm(cfg, buf, img.header_size, img.px_size)
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
    rc, ref_ppm = orc.resolve_ppm(ocfg, orc.image_new(400), ref)
    assert rc == 0
    np.testing.assert_array_equal(buf[img.header_size:], ref_ppm[img.header_size:])


def test_config2_headline_full_size_properties(orc, gpu):
    """configs[2]: scene 21, 1920x1080, 1024 spp, depth 50 — full size, in bench.py's own configuration: chunk 16 (the
    primary pass's per-item masks and batched record reads), sphere tree, both candidate lists, one segment."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(1920)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=256, rebounds=50)
    full = gpu.render_sum(host.make_pt_config(img, rt, seed=456, chunk=16))
    st = gpu.stats()
    assert st.paths == 1920 * 1080 * 1024 and 1.8 < st.rays / st.paths < 2.6 and st.not_unit == 0
    assert st.primary_rays == st.paths and st.list2_rays > 0 and st.box_tests > 0   # lists for every camera ray, second-bounce lists, the tree: the shipped path
    mean = full / 1024.0
    assert 0.0 <= mean.min() and mean.max() <= 1.0 + 1e-6
    # run-to-run determinism (work stealing must not change a bit)
    again = gpu.render_sum(host.make_pt_config(img, rt, seed=456, chunk=16))
    np.testing.assert_array_equal(again, full)
    # pixel spot checks at the full sample count: oracle with the same chunking, 24 scattered pixels
    rng = np.random.default_rng(7)
    for idx in rng.integers(0, img.num_pixels, 24):
        cfg = host.make_pt_config(img, rt, seed=456, chunk=16, istart=int(idx), istop=int(idx) + 1)
        ref, _ = _oracle_sum(orc, scene.spheres, cam.pod, cfg, nthreads=1)
        np.testing.assert_array_equal(full.reshape(-1, 3)[idx], ref.reshape(-1, 3)[idx])
    # the same frame with the unmasked record path (chunk 64: every record is looked at) differs by fp32 add order only
    other = gpu.render_sum(host.make_pt_config(img, rt, seed=456, chunk=64, istart=500 * 1920, istop=502 * 1920))
    np.testing.assert_allclose(other[500:502], full[500:502], rtol=2e-5)
    # the reference's own summation order (one sequential fp32 sum of 1024 terms, chunk = 0) differs from
    # the chunked sum only by fp32 add order (a 1024-term sequential sum carries ~1e-5 relative rounding)
    seq = gpu.render_sum(host.make_pt_config(img, rt, seed=456, chunk=0, istart=1000 * 1920, istop=1001 * 1920))
    np.testing.assert_allclose(seq[1000], full[1000], rtol=5e-5)
    cfg = host.make_pt_config(img, rt, seed=456, chunk=0, istart=1000 * 1920 + 77, istop=1000 * 1920 + 78)
    ref, _ = _oracle_sum(orc, scene.spheres, cam.pod, cfg, nthreads=1)
    np.testing.assert_array_equal(seq[1000, 77], ref[1000, 77])  # and that order is the oracle's, bit for bit


def test_n4_convergence_at_config0_frame_size(orc, gpu):
    """North star N4 at configs[0]'s own frame (256x144): the GPU's 4096-path frame against the reference's own sequential
    sampler (orc_render_sum_zig) at 256 paths per pixel — mse = sigma^2 (1/4096 + 1/256), so the PSNR must sit where two
    unbiased estimates of one image put it (measured 48x27: 256 reference paths -> ~42 dB), and the GPU frame against an
    independent oracle Philox frame of 512 paths per pixel likewise."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(256)
    assert img.height == 144
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=1024, rebounds=50)
    g = gpu.render_mean(host.make_pt_config(img, rt, seed=1, chunk=16)).astype(np.float64)
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam.pod), C.sizeof(ocam))
    oimg = orc.image_new(256)
    spheres, stream = orc.setup_scene(21, 456)
    ocfg = orc.make_config(oimg, subpixels=4, rays_per_subsample=64, rebounds=50, mode=2)
    rc, s, stz = orc.render_sum_zig(spheres, ocam, ocfg, stream)      # single thread, as main.zig:166-177 (~10 s)
    assert rc == 0 and stz.paths == 256 * 144 * 256
    z = orc.resolve_mean(ocfg, s).astype(np.float64)
    psnr_ref = 10.0 * np.log10(1.0 / float(np.mean((g - z) ** 2)))
    pcfg = orc.make_config(oimg, subpixels=4, rays_per_subsample=128, rebounds=50, mode=0, seed=2)
    rc, s, _ = orc.render_sum(spheres, ocam, pcfg, nthreads=32)
    assert rc == 0
    p = orc.resolve_mean(pcfg, s).astype(np.float64)
    psnr_phi = 10.0 * np.log10(1.0 / float(np.mean((g - p) ** 2)))
    # two estimates with n1 and n2 paths: expected PSNR difference 10 log10((1/4096 + 1/256) / (1/4096 + 1/512)) = 2.7 dB
    assert psnr_ref >= 38.0 and psnr_phi >= 40.0, (psnr_ref, psnr_phi)
    assert 1.5 < psnr_phi - psnr_ref < 4.0, (psnr_ref, psnr_phi)


def test_config3_sample_range_split_4k(gpu):
    """configs[3] shape (3840x2160, sample-range split), at 64 of the 4096 spp: the two halves of the
    sample range add up to the whole frame, as the RCCL reduce will add them."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(3840)
    assert img.height == 2160
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=16, rebounds=50)
    whole = gpu.render_sum(host.make_pt_config(img, rt, seed=456, chunk=32))
    a = gpu.render_sum(host.make_pt_config(img, rt, seed=456, chunk=32, sample_start=0, sample_stop=32))
    b = gpu.render_sum(host.make_pt_config(img, rt, seed=456, chunk=32, sample_start=32, sample_stop=64))
    np.testing.assert_array_equal(a + b, whole)  # chunk = 32: the same partial sums, added in the same order


def test_tile_split_is_exact_partition_full_hd(gpu):
    """Tile split (configs[4]'s partitioning) on a full-HD frame: 8 owners, disjoint, union == single GPU."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(1920)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=2, rebounds=50)
    whole = gpu.render_sum(host.make_pt_config(img, rt, seed=1))
    acc = np.zeros_like(whole)
    for g in range(8):
        part = gpu.render_sum(host.make_pt_config(img, rt, seed=1, tile_pixels=256, tile_count=8, tile_index=g))
        assert not ((acc != 0) & (part != 0)).any()
        acc += part
    np.testing.assert_array_equal(acc, whole)


def test_config4_many_tiles_against_oracle(orc, gpu):
    """configs[4]'s scene shape: a synthetic random-sphere scene larger than one LDS tile (tiles are
    restaged every ray step), small frame, against the oracle."""
    from path_tracer_amd import host
    scene = host.Scene().sceneSynthetic(3000, 1)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(64)
    for deterministic, spp in ((True, 1), (False, 2)):
        rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=spp, rebounds=50)
        cfg = host.make_pt_config(img, rt, seed=1, deterministic=deterministic)
        got = gpu.render_sum(cfg)
        ref, st = _oracle_sum(orc, scene.spheres, cam.pod, cfg)
        assert gpu.stats().rays == st.rays
        np.testing.assert_array_equal(got, ref)


def test_config4_100k_spheres_runs_and_matches_spot_pixels(orc, gpu):
    """The full 100 000-sphere table (196 LDS tiles per ray step): a few pixels against the oracle."""
    from path_tracer_amd import host
    scene = host.Scene().sceneSynthetic(100000, 1)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(1920)
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=2, rebounds=50)
    row = 700
    cfg = host.make_pt_config(img, rt, seed=1, istart=row * 1920, istop=row * 1920 + 256)
    got = gpu.render_sum(cfg)
    ref, st = _oracle_sum(orc, scene.spheres, cam.pod, cfg)
    assert gpu.stats().rays == st.rays
    np.testing.assert_array_equal(got, ref)


def test_convergence_psnr_4096spp(orc, gpu):
    """North star: the stochastic image converges — two INDEPENDENT 4096-spp estimates (GPU with seed 1,
    oracle with seed 2) agree to >= 40 dB PSNR."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(48)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=1024, rebounds=50)
    g = gpu.render_mean(host.make_pt_config(img, rt, seed=1))
    ref_sum, _ = _oracle_sum(orc, scene.spheres, cam.pod, host.make_pt_config(img, rt, seed=2))
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(host.make_pt_config(img, rt, seed=2)), C.sizeof(ocfg))
    r = orc.resolve_mean(ocfg, ref_sum)
    mse = float(np.mean((g.astype(np.float64) - r.astype(np.float64)) ** 2))
    psnr = 10.0 * np.log10(1.0 / mse)
    assert psnr >= 40.0, psnr


def test_convergence_against_the_reference_sampler(orc, gpu):
    """North star: "the stochastic output must converge to the reference's image at high spp (>= 40 dB at 4096 spp)".
    The GPU's 4096-path frame (Philox streams) against the REFERENCE'S OWN SAMPLER — main.zig:166-177's single-thread
    render() drawing u, v and the lens offsets from the sequential std.rand stream that built the scene and the
    scatters from material.zig:15's module-level DefaultPrng.init(0), restated by orc_render_sum_zig.  Two unbiased
    estimators of one image: mse = sigma^2 (1/4096 + 1/n), so the PSNR must clear 40 dB and must RISE with n
    (measured on the CPU with the bit-identical oracle: 1024 / 2048 / 4096 reference paths -> 48.0 / 50.7 / 52.0 dB;
    a bias from the generator swap would show as a floor).  Tolerance: >= 40 dB at n = 1024, >= +3 dB from n = 256
    (expected 10 log10(17/5) = 5.3)."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(48)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=1024, rebounds=50)
    g = gpu.render_mean(host.make_pt_config(img, rt, seed=1)).astype(np.float64)
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam.pod), C.sizeof(ocam))
    oimg = orc.image_new(48)
    psnr = {}
    for rps in (64, 256):
        spheres, stream = orc.setup_scene(21, 456)      # the stream continues from the scene set-up (main.zig:100,173)
        np.testing.assert_array_equal(spheres, scene.spheres)
        ocfg = orc.make_config(oimg, subpixels=4, rays_per_subsample=rps, rebounds=50, mode=2)
        rc, s, st = orc.render_sum_zig(spheres, ocam, ocfg, stream)
        assert rc == 0 and st.paths == 48 * 27 * 4 * rps
        z = orc.resolve_mean(ocfg, s).astype(np.float64)
        psnr[4 * rps] = 10.0 * np.log10(1.0 / float(np.mean((g - z) ** 2)))
    assert psnr[1024] >= 40.0, psnr
    assert psnr[1024] >= psnr[256] + 3.0, psnr


@pytest.mark.parametrize("deterministic", [True, False])
def test_config2_full_hd_frame_against_oracle(orc, gpu, deterministic):
    """configs[2]'s frame (scene 21, 1920x1080, depth 50) at 4 paths per pixel: every one of the 2 073 600
    pixels equals the oracle's, bit for bit (8.3 M paths, ~17 M rays)."""
    host, scene, cam = _setup(gpu, 21, 456)
    img = host.Image(1920)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=1, rebounds=50)
    cfg = host.make_pt_config(img, rt, seed=456, chunk=2, deterministic=deterministic)
    got = gpu.render_sum(cfg)
    ref, st = _oracle_sum(orc, scene.spheres, cam.pod, cfg, nthreads=32)
    gst = gpu.stats()
    assert gst.paths == st.paths == 1920 * 1080 * 4 and gst.rays == st.rays
    np.testing.assert_array_equal(got, ref)
