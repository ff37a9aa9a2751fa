"""The reference's first-hit colour functions colorNormal / colorAlbedo (illumination.zig:42-73), which
render() keeps as commented-out alternatives to radiance (renderers/ppm_image.zig:64-66): oracle
known answers on the CPU, GPU == oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest


def _oracle_frame(orc, n_scene, width, illumination, mode=1, spp=1, seed=3):
    spheres, _ = orc.setup_scene(n_scene, 456)
    cam = orc.camera_preset(n_scene)
    img = orc.image_new(width)
    cfg = orc.make_config(img, subpixels=1, rays_per_subsample=spp, rebounds=6, mode=mode, seed=seed, illumination=illumination)
    rc, out, st = orc.render_sum(spheres, cam, cfg, nthreads=4)
    assert rc == 0
    return spheres, cam, img, cfg, out, st


def test_oracle_color_albedo_known_answers(orc):
    """Scene 18 (scene.zig:40-49), distant view: the central ray hits the blue Lambertian (0.1,0.2,0.5); every
    hit pixel shows one of the scene's albedos or (1,1,1) for the glass sphere; sky pixels equal radiance's."""
    spheres, cam, img, cfg, alb, st = _oracle_frame(orc, 18, 64, 2)
    h, w = img.height, img.width
    assert st.rays == st.paths == h * w                      # one ray per sample, no bounces
    np.testing.assert_array_equal(alb[h // 2, w // 2], np.float32([0.1, 0.2, 0.5]))
    palette = {(np.float32(0.8), np.float32(0.8), np.float32(0.0)), (np.float32(0.1), np.float32(0.2), np.float32(0.5)),
               (np.float32(0.8), np.float32(0.6), np.float32(0.2)), (np.float32(1.0), np.float32(1.0), np.float32(1.0))}
    _, _, _, _, rad, _ = _oracle_frame(orc, 18, 64, 0)
    sky = np.zeros((h, w), dtype=bool)
    for y in range(h):
        for x in range(w):
            px = tuple(alb[y, x])
            if px not in palette:                              # not an albedo: must be the sky lerp, as radiance sees it
                np.testing.assert_array_equal(alb[y, x], rad[y, x])
                sky[y, x] = True
    assert sky[0].all() and 0.2 < sky.mean() < 0.95


def test_oracle_color_normal_known_answers(orc):
    """colorNormal = (unit(p - (0,0,-1)) + 1)/2 about the CONSTANT centre (0,0,-1) (illumination.zig:46): on the
    central sphere (whose centre that is) it is the true normal, facing the camera at (-2,2,1)."""
    spheres, cam, img, cfg, nrm, _ = _oracle_frame(orc, 18, 640, 1)   # fine pixels: the +1 row quirk shifts p by one pixel
    h, w = img.height, img.width
    to_cam = np.array([-2.0, 2.0, 2.0]) / np.sqrt(12.0)        # (lookfrom - centre) normalised
    np.testing.assert_allclose(nrm[h // 2, w // 2], (to_cam + 1.0) * 0.5, atol=0.06)
    assert nrm.min() >= 0.0 and nrm.max() <= 1.0
    _, _, _, _, rad, _ = _oracle_frame(orc, 18, 640, 0)
    np.testing.assert_array_equal(nrm[0], rad[0])            # top row is sky in both


@pytest.mark.gpu
@pytest.mark.parametrize("n_scene,width", [(18, 96), (21, 128)])
@pytest.mark.parametrize("illumination", [1, 2])
@pytest.mark.parametrize("mode,spp", [(1, 1), (0, 5)])
def test_gpu_aov_bit_exact(orc, gpu, n_scene, width, illumination, mode, spp):
    spheres, cam, img, cfg, ref, st = _oracle_frame(orc, n_scene, width, illumination, mode=mode, spp=spp, seed=11)
    from path_tracer_amd._lib import pt_config
    gcfg = pt_config()
    C.memmove(C.byref(gcfg), C.byref(cfg), C.sizeof(gcfg))
    gpu.set_scene(spheres)
    gpu.set_camera(cam)
    got = gpu.render_sum(gcfg)
    np.testing.assert_array_equal(got, ref)
    assert gpu.stats().rays == st.rays == st.paths


@pytest.mark.gpu
def test_gpu_rejects_unknown_illumination(gpu):
    from path_tracer_amd import _lib, host
    scene = host.Scene().setupScene(0, 18)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(host.Camera.preset(18).pod)
    with pytest.raises(_lib.PtGpuError) as e:
        gpu.render_sum(host.make_pt_config(host.Image(32), host.RayTracerConfig(), illumination=3))
    assert e.value.status == _lib.PT_ERR_BAD_ARG
