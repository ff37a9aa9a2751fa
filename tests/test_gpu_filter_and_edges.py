"""The conservative LDS filter never hides a hit, and the edge cases of the boundary.

PT_FN_SCENE_HIT_FILTERED runs the megakernel's own Scene.is_hit (filter scan + exact pass on the
survivors); PT_FN_SCENE_HIT is the plain exact scan over every sphere (the reference's loop).  The two
must agree bit for bit on every ray — including rays built to graze spheres, rays starting on
surfaces, far-away origins (large |o|^2 + |c|^2, where the filter's rounding margin matters) and scenes
with huge / tiny / negative radii.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(F)


def _grazing_rays(rng, spheres, n, origin_scale):
    """Rays aimed at the silhouette of randomly chosen spheres (distance to the centre = r*(1+-tiny))."""
    sp = spheres[rng.integers(0, len(spheres), n)]
    c = np.stack([sp["cx"], sp["cy"], sp["cz"]], axis=1).astype(np.float64)
    r = np.abs(sp["radius"].astype(np.float64))
    org = rng.normal(0, origin_scale, (n, 3))
    to_c = c - org
    dist = np.linalg.norm(to_c, axis=1, keepdims=True)
    axis = to_c / dist
    perp = np.cross(axis, rng.normal(0, 1, (n, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    eps = rng.choice([0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-3, -1e-3], (n, 1))
    target = c + perp * r[:, None] * (1.0 + eps)
    d = _unit(target - org)
    return np.concatenate([org.astype(F), d], axis=1).astype(F)


def _compare(gpu, rays, t_min=0.1, t_max=1000.0):
    from path_tracer_amd import _lib
    exact = gpu.debug_eval(_lib.PT_FN_SCENE_HIT, rays, t_min, t_max)
    filt = gpu.debug_eval(_lib.PT_FN_SCENE_HIT_FILTERED, rays, t_min, t_max)
    np.testing.assert_array_equal(filt, exact)
    return exact


@pytest.mark.parametrize("n_scene", [18, 21])
def test_filter_matches_exact_scan_reference_scenes(gpu, n_scene):
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, n_scene)
    gpu.set_scene(scene.spheres)
    rng = np.random.default_rng(11 + n_scene)
    rays = np.concatenate([
        _grazing_rays(rng, scene.spheres, 20000, 5.0),
        _grazing_rays(rng, scene.spheres, 20000, 300.0),      # far origins: big |o|^2
        np.concatenate([rng.normal(0, 8, (20000, 3)).astype(F), _unit(rng.normal(0, 1, (20000, 3)))], axis=1),
    ])
    out = _compare(gpu, rays)
    assert (out[:, 0] >= 0).sum() > 1000 and (out[:, 0] < 0).sum() > 1000   # both outcomes exercised


def test_filter_rays_starting_on_surfaces(gpu):
    """Bounce rays start on the sphere they just hit (disc = hb^2 exactly on that sphere)."""
    from path_tracer_amd import _lib, host
    scene = host.Scene().setupScene(456, 21)
    gpu.set_scene(scene.spheres)
    rng = np.random.default_rng(5)
    primary = np.concatenate([np.tile(np.array([[13, 2, 3]], F), (30000, 1)),
                              _unit(np.stack([rng.uniform(-14, -4, 30000), rng.uniform(-2.5, 0.3, 30000),
                                              rng.uniform(-6, 3, 30000)], axis=1))], axis=1)
    first = gpu.debug_eval(_lib.PT_FN_SCENE_HIT, primary)
    hit = first[:, 0] >= 0
    p, n = first[hit, 2:5], first[hit, 5:8]
    d2 = _unit(n + rng.random((hit.sum(), 3)))
    out = _compare(gpu, np.concatenate([p, d2], axis=1).astype(F))
    assert (out[:, 0] >= 0).sum() > 100


def test_filter_adversarial_scene(gpu):
    """Huge, tiny and negative radii, far-away and coincident centres, more than one LDS tile."""
    from path_tracer_amd import _lib
    rng = np.random.default_rng(3)
    n = 1500
    sp = np.zeros(n, dtype=_lib.SPHERE_DTYPE)
    sp["cx"], sp["cy"], sp["cz"] = rng.normal(0, 30, n), rng.normal(0, 30, n), rng.normal(0, 30, n)
    sp["radius"] = rng.choice([1e-3, 0.05, 0.5, 3.0, 40.0, -0.7], n)
    sp["kind"] = 0
    sp[0] = (0, -5000, 0, 5000, 0, 0.5, 0.5, 0.5, 0)          # ground-like giant
    sp[1] = (2000, 0, 0, 1.0, 0, 1, 1, 1, 0)                  # far away
    sp[2] = sp[3] = (1, 1, 1, 0.25, 0, 1, 1, 1, 0)            # coincident: the first must win ties
    gpu.set_scene(sp)
    rays = np.concatenate([
        _grazing_rays(rng, sp, 30000, 20.0),
        _grazing_rays(rng, sp[2:4], 2000, 3.0),
        np.concatenate([rng.normal(0, 50, (20000, 3)).astype(F), _unit(rng.normal(0, 1, (20000, 3)))], axis=1),
    ])
    out = _compare(gpu, rays, t_min=0.001, t_max=1.0e6)
    assert not (out[:, 0] == 3).any()                         # scene.zig:102 strict '<': index 2 wins over 3


# ---- edge cases of the boundary --------------------------------------------------------------------------
def _cmp_render(orc, gpu, spheres, cam_pod, cfg):
    ocfg = orc.Config()
    C.memmove(C.byref(ocfg), C.byref(cfg), C.sizeof(ocfg))
    ocam = orc.Camera()
    C.memmove(C.byref(ocam), C.byref(cam_pod), C.sizeof(ocam))
    rc, ref, st = orc.render_sum(spheres, ocam, ocfg, nthreads=8)
    assert rc == 0
    got = gpu.render_sum(cfg)
    np.testing.assert_array_equal(got, ref)
    assert gpu.stats().rays == st.rays
    return got


@pytest.mark.parametrize("rebounds", [0, 1, 255])
def test_rebounds_extremes(orc, gpu, rebounds):
    """rebounds = 0: every hit is black at once (illumination.zig:80-81); 255 = the reference's u8 maximum."""
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 20)
    cam = host.Camera.preset(20)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(48)
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=3, rebounds=rebounds)
    _cmp_render(orc, gpu, scene.spheres, cam.pod, host.make_pt_config(img, rt, seed=8))


@pytest.mark.parametrize("nspheres", [1, 7, 8, 9, 33, 511, 512, 513])
def test_ragged_sphere_counts(orc, gpu, nspheres):
    """Counts around the padding (8), mask-word (32) and LDS-tile (512) boundaries."""
    from path_tracer_amd import host
    scene = host.Scene().sceneSynthetic(nspheres, 5)
    cam = host.Camera.preset(21)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(40)
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=2, rebounds=12)
    _cmp_render(orc, gpu, scene.spheres, cam.pod, host.make_pt_config(img, rt, seed=2))


def test_empty_and_single_pixel_ranges(orc, gpu):
    from path_tracer_amd import host
    scene = host.Scene().setupScene(456, 18)
    cam = host.Camera.preset(18)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(cam.pod)
    img = host.Image(33)   # odd width, height 18
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=1)
    empty = gpu.render_sum(host.make_pt_config(img, rt, istart=10, istop=10))
    assert not empty.any() and gpu.stats().paths == 0
    last = img.num_pixels - 1
    got = _cmp_render(orc, gpu, scene.spheres, cam.pod, host.make_pt_config(img, rt, seed=3, istart=last, istop=last + 1))
    assert got.reshape(-1, 3)[last].any() and not got.reshape(-1, 3)[:last].any()


def test_bad_configs_are_rejected(gpu):
    from path_tracer_amd import _lib, host
    from path_tracer_amd._lib import PtGpuError
    scene = host.Scene().setupScene(0, 18)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(host.Camera.preset(18).pod)
    img = host.Image(32)
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=1)
    for kw in (dict(istart=5, istop=4), dict(istop=img.num_pixels + 1), dict(sample_start=2, sample_stop=1),
               dict(sample_stop=9), dict(tile_pixels=16, tile_count=0), dict(tile_pixels=16, tile_count=2, tile_index=2),
               dict(tile_pixels=16, tile_count=2, istart=8)):
        with pytest.raises(PtGpuError) as e:
            gpu.render_sum(host.make_pt_config(img, rt, **kw))
        assert e.value.status == _lib.PT_ERR_BAD_ARG
    bad = scene.spheres.copy()
    bad["kind"][0] = 3
    with pytest.raises(PtGpuError):
        gpu.set_scene(bad)
    with pytest.raises(PtGpuError):
        gpu.set_scene(scene.spheres[:0])
    gpu.set_scene(scene.spheres)
    cfg = host.make_pt_config(img, rt)
    cfg.max_px_value = 65535            # the reference's colNumToString only formats u8
    with pytest.raises(PtGpuError) as e:
        gpu.render_ppm(cfg, np.zeros(img.size, dtype=np.uint8), img.header_size, img.px_size)
    assert e.value.status == _lib.PT_ERR_UNSUPPORTED
    with pytest.raises(PtGpuError) as e:   # NoSpaceLeft: slice too small
        gpu.render_ppm(host.make_pt_config(img, rt), np.zeros(100, dtype=np.uint8), img.header_size, img.px_size)
    assert e.value.status == _lib.PT_ERR_NO_SPACE


def test_colour_above_max_is_clamped_and_reported(gpu):
    """renderers/ppm_image.zig:75-77 asserts col <= 255: a sky brighter than 1.0 is written as 255 and the
    call returns ValueExceedsMaxValue after writing every record."""
    from path_tracer_amd import _lib, host
    from path_tracer_amd._lib import PtGpuError
    scene = host.Scene().setupScene(0, 18)
    gpu.set_scene(scene.spheres)
    gpu.set_camera(host.Camera.preset(18).pod)
    img = host.Image(32)
    rt = host.RayTracerConfig(subpixels=1, rays_per_subsample=1)
    cfg = host.make_pt_config(img, rt, deterministic=True)
    cfg.blend_start[0] = cfg.blend_stop[0] = 3.0      # red sky = 3.0 -> 765 > 255
    buf = np.zeros(img.size, dtype=np.uint8)
    with pytest.raises(PtGpuError) as e:
        gpu.render_ppm(cfg, buf, img.header_size, img.px_size)
    assert e.value.status == _lib.PT_ERR_VALUE_EXCEEDS_MAX
    first = bytes(buf[img.header_size:img.header_size + 12])
    assert first[:3] == b"255" and first[11:12] == b"\n"        # top-left pixel is sky


# ---- the filter invariant itself ---------------------------------------------------------------------------
def _pairs(rng, n, o_scale, c_scale):
    """(ray, sphere) pairs: random, aimed at the silhouette (tangent +- tiny), and origins ON the sphere."""
    r = 10.0 ** rng.uniform(-3, 4, n) * rng.choice([1.0, 1.0, 1.0, -1.0], n)      # radii 1e-3 .. 1e4, some negative
    c = rng.normal(0, 1, (n, 3)) * c_scale * rng.uniform(0, 1, (n, 1))
    o = rng.normal(0, 1, (n, 3)) * o_scale * rng.uniform(0, 1, (n, 1))
    kind = rng.integers(0, 4, n)
    on = kind == 3                                                                 # bounce rays: origin on the surface
    u = rng.normal(0, 1, (n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    o[on] = c[on] + u[on] * np.abs(r[on, None])
    to_c = c - o
    dist = np.linalg.norm(to_c, axis=1, keepdims=True) + 1e-30
    axis = to_c / dist
    perp = np.cross(axis, rng.normal(0, 1, (n, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True) + 1e-30
    eps = rng.choice([0.0, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4], (n, 1))
    d = c + perp * np.abs(r[:, None]) * (1.0 + eps) - o                            # through the silhouette
    rnd = rng.normal(0, 1, (n, 3))
    d[kind == 0] = rnd[kind == 0]                                                  # anywhere
    d[on] = rnd[on]
    d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-30
    d = d.astype(F)
    # the megakernel's directions are unit vectors to ~1e-7 (vunit): renormalise in f32 as vec3.zig:61-64 does
    inv = (F(1.0) / np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])).astype(F)
    d = (d * inv[:, None]).astype(F)
    return np.concatenate([o.astype(F), d, c.astype(F), r.astype(F)[:, None]], axis=1).astype(F)


def test_filter_invariant_discriminant_nonneg_implies_candidate(gpu):
    """The exact pass only ever sees the filter's candidates, and the parity tests compare final hits — a
    dropped candidate that is not the closest hit would be invisible to them.  PT_FN_FILTER_BIT returns the
    filter's verdict for a (ray, sphere) pair beside the reference's discriminant (sphere.zig:64-74): on
    1.2e6 adversarial pairs (|o|, |c| up to the t_max = 1000 scale, radii 1e-3 .. 1e4, tangent rays, origins on
    the surface) discriminant >= 0 must imply "candidate"."""
    from path_tracer_amd import _lib
    rng = np.random.default_rng(2024)
    total = nonneg = cand = 0
    for o_scale, c_scale in ((5.0, 12.0), (30.0, 30.0), (1000.0, 1000.0), (1000.0, 5.0), (0.0, 1000.0), (2000.0, 2000.0)):
        pairs = _pairs(rng, 200000, o_scale, c_scale)
        out = gpu.debug_eval(_lib.PT_FN_FILTER_BIT, pairs)
        bit, disc = (out[:, 0] == 1.0) & (out[:, 2] == 1.0), out[:, 1]   # general and shared-ordinate forms of the test
        # the device's discriminant is the reference's f32 arithmetic: cross-check against numpy float32
        o, d, c, r = pairs[:, 0:3], pairs[:, 3:6], pairs[:, 6:9], pairs[:, 9]
        oc = (o - c).astype(F)
        a = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(F)
        half_b = (oc[:, 0] * d[:, 0] + oc[:, 1] * d[:, 1] + oc[:, 2] * d[:, 2]).astype(F)
        cc = ((oc[:, 0] * oc[:, 0] + oc[:, 1] * oc[:, 1] + oc[:, 2] * oc[:, 2]).astype(F) - r * r).astype(F)
        np.testing.assert_array_equal(disc, (half_b * half_b - a * cc).astype(F))
        bad = (disc >= 0) & ~bit
        assert not bad.any(), (o_scale, c_scale, pairs[bad][:3], disc[bad][:3])
        assert ((out[:, 0] == 1.0) == (out[:, 2] == 1.0)).mean() > 0.999          # the two forms agree except at the margin
        total += len(pairs); nonneg += int((disc >= 0).sum()); cand += int(bit.sum())
    assert total >= 1_000_000 and nonneg > 100_000          # the invariant was exercised, not vacuous
    assert cand < 0.9 * total                               # and the filter does reject


# ---- shared-ordinate mask words ----------------------------------------------------------------------------
@pytest.mark.parametrize("nspheres,pattern", [(700, "all"), (1500, "runs"), (96, "all"), (100, "alternate"), (40, "negzero")])
def test_shared_ordinate_words_against_oracle(orc, gpu, nspheres, pattern):
    """Mask words (32 consecutive spheres) whose spheres share one centre ordinate take the filter's short test.
    Scenes built to mix both kinds of word in one tile and across tiles — all words shared; runs of shared and
    general words; a shared word followed by a partial one; no word shared because two values alternate; +0.0 vs
    -0.0 (different bits: not shared) — must still equal the oracle bit for bit, and the filtered Scene.is_hit the
    plain exact scan."""
    from path_tracer_amd import _lib, host
    sp = host.Scene().sceneSynthetic(nspheres, 9).spheres.copy()
    idx = np.arange(nspheres)
    if pattern == "all":
        sp["cy"] = 0.25
    elif pattern == "runs":
        run = (idx // 32) % 3 != 1                 # words 0, 2, 3, 5, 6, ... shared, words 1, 4, 7, ... general
        sp["cy"][run] = (0.2 + 0.01 * ((idx // 32) % 5))[run].astype(F)
    elif pattern == "alternate":
        sp["cy"] = np.where(idx % 2 == 0, 0.25, 0.3).astype(F)
    elif pattern == "negzero":
        sp["cy"] = np.where(idx % 2 == 0, 0.0, -0.0).astype(F)
    sp["cy"][0] = -1000.0                          # the ground sphere keeps its own centre: word 0 is general
    gpu.set_scene(sp)
    info = gpu.scene_info()
    expect = {"all": (nspheres // 32 - 1) * 32, "alternate": 0, "negzero": 0}.get(pattern)
    if expect is not None:
        assert info["shared_ordinate_tests"] == expect
    else:
        assert 0 < info["shared_ordinate_tests"] < nspheres
    cam = host.Camera.preset(21)
    gpu.set_camera(cam.pod)
    img = host.Image(48)
    rt = host.RayTracerConfig(subpixels=2, rays_per_subsample=2, rebounds=12)
    _cmp_render(orc, gpu, sp, cam.pod, host.make_pt_config(img, rt, seed=4))
    rng = np.random.default_rng(nspheres)
    rays = np.concatenate([_grazing_rays(rng, sp, 20000, 15.0),
                           np.concatenate([rng.normal(0, 20, (10000, 3)).astype(F), _unit(rng.normal(0, 1, (10000, 3)))], axis=1)])
    _compare(gpu, rays)
