"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

The kernel reproduces the reference's unfused f32 arithmetic and consumes the same
Philox draws in the same order as the oracle, so the comparison is exact equality of
the float32 radiance sums — in deterministic mode (north-star tolerance: 1e-5 relative)
and in stochastic mode alike.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render_both(orc, gpu, n_scene, width, seed_scene=456, **kw):
    spheres, _ = orc.setup_scene(n_scene, seed_scene)
    cam = orc.camera_preset(n_scene)
    img = orc.image_new(width)
    cfg = orc.make_config(img, **kw)
    rc, ref, st = orc.render_sum(spheres, cam, cfg, nthreads=8)
    assert rc == 0
    gpu.set_scene(spheres)
    gpu.set_camera(cam)
    got = gpu.render_sum(cfg)
    return ref, got, st, gpu.stats(), cfg, img


@pytest.mark.parametrize("n_scene,width,rebounds", [(18, 32, 6), (18, 64, 50), (21, 64, 6), (21, 64, 50), (20, 48, 6)])
def test_deterministic_bit_exact(orc, gpu, n_scene, width, rebounds):
    ref, got, st, gst, cfg, img = _render_both(orc, gpu, n_scene, width, subpixels=1, rays_per_subsample=1,
                                               rebounds=rebounds, mode=1)
    assert gst.paths == st.paths and gst.rays == st.rays
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("n_scene,width,spp,rebounds,seed", [(18, 40, 3, 50, 42), (21, 48, 2, 6, 456), (21, 32, 4, 50, 7),
                                                              (19, 32, 2, 6, 1)])
def test_stochastic_bit_exact(orc, gpu, n_scene, width, spp, rebounds, seed):
    ref, got, st, gst, cfg, img = _render_both(orc, gpu, n_scene, width, subpixels=4, rays_per_subsample=spp,
                                               rebounds=rebounds, mode=0, seed=seed)
    assert gst.paths == st.paths and gst.rays == st.rays
    np.testing.assert_array_equal(got, ref)


def test_chunked_and_ranges(orc, gpu):
    # sample-range split + chunked partial sums + partial pixel range, as the multi-GPU path uses them
    spheres, _ = orc.setup_scene(21, 456)
    cam = orc.camera_preset(21)
    img = orc.image_new(40)
    gpu.set_scene(spheres)
    gpu.set_camera(cam)
    npx = img.width * img.height
    cfg = orc.make_config(img, subpixels=4, rays_per_subsample=5, rebounds=50, seed=9, idx_start=37, idx_stop=npx - 11,
                          sample_start=3, sample_stop=17, chunk=4)
    rc, ref, st = orc.render_sum(spheres, cam, cfg, nthreads=8)
    assert rc == 0
    got = gpu.render_sum(cfg)
    np.testing.assert_array_equal(got, ref)
    assert not got.reshape(-1, 3)[:37].any() and not got.reshape(-1, 3)[npx - 11:].any()


def test_ppm_bytes(orc, gpu):
    # render() replacement: the pixel records of the .ppm buffer, byte for byte
    spheres, _ = orc.setup_scene(21, 456)
    cam = orc.camera_preset(21)
    img = orc.image_new(64)
    cfg = orc.make_config(img, subpixels=4, rays_per_subsample=2, rebounds=6, seed=456)
    rc, ref_sum, _ = orc.render_sum(spheres, cam, cfg, nthreads=8)
    rc2, ref_ppm = orc.resolve_ppm(cfg, img, ref_sum)
    assert rc == 0 and rc2 == 0
    gpu.set_scene(spheres)
    gpu.set_camera(cam)
    buf = np.zeros(img.size, dtype=np.uint8)
    gpu.render_ppm(cfg, buf, img.header_size, img.px_size)
    np.testing.assert_array_equal(buf[img.header_size:], ref_ppm[img.header_size:])
    assert not buf[:img.header_size].any()  # the caller writes the header (main.zig:180-182)
