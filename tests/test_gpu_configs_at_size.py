"""BASELINE configs[3] and configs[4] at size, one rank's share each (what one of 8 GPUs renders).

configs[3]: scene 21, 3840x2160, 4096 paths/pixel, depth 50, sample-range split: one rank's 512 samples of every
pixel (3.2 GB of chunk partials at chunk 16, 136 GB of primary-pass records -> two segments under the limits), spot pixels
against the oracle under the same chunking, and exact additivity of two adjacent 512-sample ranges.
configs[4]: 100 000 spheres, 1920x1080, 256 paths/pixel, depth 50, interleaved 256-pixel tiles, one of 8 owners,
spot pixels against the oracle; pixels of the other owners stay untouched.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_pixel(orc, spheres, cam_pod, cfg_of, idx):
    cfg = cfg_of(int(idx), int(idx) + 1)
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam_pod), C.sizeof(ocam))
    rc, out, _ = orc.render_sum(spheres, ocam, ocfg, nthreads=1)
    assert rc == 0
    return out.reshape(-1, 3)[int(idx)]


def test_config3_one_rank_share_4k_512_of_4096(orc, gpu):
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 21)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(3840)
    assert (img.width, img.height) == (3840, 2160)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=1024, rebounds=50)     # 4096 paths per pixel

    def share(i0=0, i1=None, s0=512, s1=1024, chunk=16):                             # rank 1 of 8
        return host.make_pt_config(img, rt, seed=456, chunk=chunk, sample_start=s0, sample_stop=s1, istart=i0, istop=i1)

    got = gpu.render_sum(share())
    st = gpu.stats()
    assert st.paths == 3840 * 2160 * 512 and st.not_unit == 0 and 1.8 < st.rays / st.paths < 2.6
    # 3.2 GB of chunk partials through the 2 GiB PT_LIMIT_PARTIAL_BYTES buffer: two segments (the primary pass's 136 GB
    # of path records, 32 B x 4.35e9 paths, would also need two under its 96 GiB PT_LIMIT_PRIMARY_BYTES)
    assert st.segments == 2 and 0.9 * st.paths < st.primary_rays <= st.paths
    rng = np.random.default_rng(33)
    flat = got.reshape(-1, 3)
    for idx in rng.integers(0, img.num_pixels, 16):
        np.testing.assert_array_equal(flat[idx], _oracle_pixel(orc, scene.spheres, cam.pod, share, idx))
    # additivity, exact: with chunk = 512 each range is ONE partial sum, and the 1024-sample range adds the two in order
    a = gpu.render_sum(share(s0=0, s1=512, chunk=512))
    b = gpu.render_sum(share(s0=512, s1=1024, chunk=512))
    whole = gpu.render_sum(share(s0=0, s1=1024, chunk=512))
    np.testing.assert_array_equal(a + b, whole)
    del a, whole
    # the same samples under another chunking differ by fp32 add order only
    np.testing.assert_allclose(b, got, rtol=1e-4)


def test_config4_one_rank_tile_share_100k_spheres(orc, gpu):
    from path_tracer_amd import host
    scene = host.Scene().sceneSynthetic(100000, 1)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(1920)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=64, rebounds=50)       # 256 paths per pixel
    owner, owners, tile = 3, 8, 256

    def share(i0=0, i1=None):
        return host.make_pt_config(img, rt, seed=1, chunk=16, tile_pixels=tile, tile_count=owners, tile_index=owner,
                                   istart=i0, istop=i1)

    got = gpu.render_sum(share())
    st = gpu.stats()
    flat = got.reshape(-1, 3)
    mine = (np.arange(img.num_pixels) // tile) % owners == owner
    assert st.paths == int(mine.sum()) * 256 and st.not_unit == 0
    assert st.primary_rays > 0 and st.box_tests > 0    # lists built by pt_primary_lists<true>'s tree walk, megakernel through the tree
    assert not flat[~mine].any() and flat[mine].any(axis=1).all()
    rng = np.random.default_rng(44)
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam.pod), C.sizeof(ocam))
    for idx in rng.choice(np.flatnonzero(mine), 6, replace=False):
        # one pixel as a plain range: the tile split only selects pixels, a pixel's samples do not depend on it
        one = host.make_pt_config(img, rt, seed=1, chunk=16, istart=int(idx), istop=int(idx) + 1)
        ocfg = orc.Config()
        C.memmove(C.byref(ocfg), C.byref(one), C.sizeof(ocfg))
        rc, ref, _ = orc.render_sum(scene.spheres, ocam, ocfg, nthreads=1)
        assert rc == 0
        np.testing.assert_array_equal(flat[idx], ref.reshape(-1, 3)[idx])


def test_config4_whole_frame_tile_shares_and_a_band_through_the_scan(gpu):
    """configs[4] at full size through the sphere tree: the eight owners' tile shares add up to the one-GPU frame exactly,
    and a band of 16 rows traced through the scan over all 100 000 spheres (PT_LIMIT_TREE_MIN_SPHERES = 0) has the same bits."""
    from path_tracer_amd import _lib, host
    scene = host.Scene().sceneSynthetic(100000, 1)
    gpu.set_scene(scene.spheres)
    assert gpu.tree_info()["nodes"] > 0
    gpu.set_camera(host.Camera.preset(21).pod)
    img = host.Image(1920)
    rt = host.RayTracerConfig(subpixels=4, rays_per_subsample=64, rebounds=50)       # 256 paths per pixel
    whole = gpu.render_sum(host.make_pt_config(img, rt, seed=1, chunk=16))
    st = gpu.stats()
    assert st.paths == img.num_pixels * 256 and st.not_unit == 0 and st.primary_rays > 0
    acc = np.zeros_like(whole)
    rays = 0
    for owner in range(8):
        acc += gpu.render_sum(host.make_pt_config(img, rt, seed=1, chunk=16, tile_pixels=256, tile_count=8, tile_index=owner))
        rays += gpu.stats().rays
    assert rays == st.rays
    np.testing.assert_array_equal(acc, whole)       # other owners' pixels are 0: x + 0 = x
    i0, i1 = 600 * img.width, 616 * img.width
    band_cfg = host.make_pt_config(img, rt, seed=1, chunk=16, istart=i0, istop=i1)
    band_tree = gpu.render_sum(band_cfg)
    rays_tree = gpu.stats().rays
    try:
        gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, 0)
        gpu.set_scene(scene.spheres)
        assert gpu.tree_info()["nodes"] == 0
        band_scan = gpu.render_sum(band_cfg)
        assert gpu.stats().rays == rays_tree
    finally:
        gpu.set_limit(_lib.PT_LIMIT_TREE_MIN_SPHERES, _lib.PT_TREE_MIN_SPHERES_DEFAULT)
    np.testing.assert_array_equal(band_tree, band_scan)
    np.testing.assert_array_equal(band_tree.reshape(-1, 3)[i0:i1], whole.reshape(-1, 3)[i0:i1])
