"""Pins the CPU oracle against every known-answer test the reference's own `test` blocks hold
for this path (bold entries), plus published KATs of the RNG algorithms.

The reference (Zig) cannot be run here, so these KATs — and the restatement's line-by-line
correspondence — are what pins the oracle.  Reference files are cited per test.
"""
import ctypes as C
import math

import numpy as np
import pytest

F = np.float32


def f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


# ---- vec3.zig:97-160 ------------------------------------------------------------------------
def test_vec3_length(orc):  # "length(3,4,5)=sqrt(50)"
    assert orc.lib().orc_length(f3((3, 4, 5))) == F(math.sqrt(50.0))


def test_vec3_unit_vector(orc):  # vec3.zig:131-134
    out = f3((0, 0, 0))
    orc.lib().orc_unit_vector(f3((1, 2, 3)), out)
    assert abs(orc.lib().orc_length(out) - 1.0) < 1e-5


def test_vec3_dot_cross(orc):
    assert orc.lib().orc_dot(f3((2, 3, 4)), f3((5, 6, 7))) == 56.0  # vec3.zig:136-142
    out = f3((0, 0, 0))
    orc.lib().orc_cross(f3((1, 0, 2)), f3((2, 1, 2)), out)  # cross((1,0,2),(2,1,2)) = (-2,2,1)
    assert tuple(out) == (-2.0, 2.0, 1.0)


# ---- ray.zig:26-32 --------------------------------------------------------------------------
def test_ray_point_at(orc):
    out = f3((0, 0, 0))
    orc.lib().orc_point_at(f3((0, 0, 0)), f3((1, 1, 1)), 1.0, out)
    assert tuple(out) == (1.0, 1.0, 1.0)


# ---- sphere.zig:117-141 ---------------------------------------------------------------------
def test_sphere_is_hit_miss(orc):
    sp = np.zeros(1, dtype=orc.SPHERE_DTYPE)
    sp[0] = (0, 0, -1, 0.5, 0, 1.0, 2.0, 3.0, 0.0)
    t = C.c_float()
    p, n = f3((0, 0, 0)), f3((0, 0, 0))
    hit = orc.lib().orc_sphere_hit(sp.ctypes.data, f3((0, 0, 0)), f3((0, 1, 0)), 0.001, 10000.0, C.byref(t), p, n)
    assert hit == 0
    # and the ray that looks at the sphere does hit it, at t = 0.5, normal (0,0,1)
    hit = orc.lib().orc_sphere_hit(sp.ctypes.data, f3((0, 0, 0)), f3((0, 0, -1)), 0.001, 10000.0, C.byref(t), p, n)
    assert hit == 1 and t.value == 0.5 and tuple(n) == (0.0, 0.0, 1.0)


def test_sphere_near_root_only(orc):  # sphere.zig:81-85: a ray starting inside never sees the far side
    sp = np.zeros(1, dtype=orc.SPHERE_DTYPE)
    sp[0] = (0, 0, 0, 1.0, 0, 1, 1, 1, 0)
    t = C.c_float()
    p, n = f3((0, 0, 0)), f3((0, 0, 0))
    assert orc.lib().orc_sphere_hit(sp.ctypes.data, f3((0, 0, 0)), f3((0, 0, 1)), 0.1, 1000.0, C.byref(t), p, n) == 0


# ---- material.zig:210-216 -------------------------------------------------------------------
def test_refract_unit_length(orc):
    d, n = f3((0, 0, 0)), f3((0, 0, 0))
    orc.lib().orc_unit_vector(f3((5, 2, 0)), d)
    orc.lib().orc_unit_vector(f3((-3, 2, 0)), n)
    r = f3((0, 0, 0))
    assert orc.lib().orc_refract(d, n, 0.3, r) == 1
    assert abs(orc.lib().orc_length(r) - 1.0) < 1e-5


def test_refract_total_internal_reflection(orc):  # material.zig:106-108 null branch
    d = f3((0, 0, 0))
    orc.lib().orc_unit_vector(f3((1, -0.1, 0)), d)
    r = f3((0, 0, 0))
    assert orc.lib().orc_refract(d, f3((0, 1, 0)), 1.5, r) == 0


def test_schlick_and_pow(orc):  # material.zig:120-124 with Zig's pow(x,5) as square-and-multiply
    x = F(0.37)
    x2 = F(x * x)
    x4 = F(x2 * x2)
    assert orc.lib().orc_pow(float(x), 5.0) == F(x * x4)
    r0 = F((F(1.0) - F(1.5)) / (F(1.0) + F(1.5)))
    r0 = F(r0 * r0)
    cos = F(0.25)
    y = F(F(1.0) - cos)
    y5 = F(y * F(F(y * y) * F(y * y)))
    assert orc.lib().orc_schlick(float(cos), 1.5) == F(r0 + F(F(F(1.0) - r0) * y5))


# ---- illumination.zig:92-123 ----------------------------------------------------------------
def test_lerp_rejects_non_unit(orc):
    out = f3((0, 0, 0))
    rc = orc.lib().orc_lerp(f3((1, 2, 3)), f3((0.6, 0, 0)), f3((0.2, 0.8, 0)), out)
    assert rc == -5  # error.DirectionIsNotUnitVector


def test_lerp_in_range(orc):
    d = f3((0, 0, 0))
    orc.lib().orc_unit_vector(f3((3, 4, 5)), d)
    out = f3((0, 0, 0))
    assert orc.lib().orc_lerp(d, f3((1, 1, 1)), f3((0.5, 0.7, 1.0)), out) == 0
    assert 0.0 < out[0] < 1.0 and 0.0 < out[1] < 1.0


# ---- camera.zig:122-129 ---------------------------------------------------------------------
def test_random_vector_in_unit_disk_via_cast_ray(orc):
    cam = orc.camera_preset(20)
    xi = np.array([0.99, 0.99, 0.1, 0.7, 0.5, 0.5], dtype=np.float32)  # first try rejected: (0.98,0.98)
    used = C.c_uint32()
    o, d = f3((0, 0, 0)), f3((0, 0, 0))
    orc.lib().orc_cast_ray(C.byref(cam), 0.5, 0.5, orc.fptr(xi), len(xi), C.byref(used), o, d)
    assert used.value == 4
    assert abs(orc.lib().orc_length(d) - 1.0) < 1e-5


# ---- scene.zig:113-125, scene sizes -----------------------------------------------------------
def test_scene_sizes(orc):
    assert len(orc.setup_scene(18)[0]) == 5
    sp, _ = orc.setup_scene(21, 456)
    assert len(sp) == 1 + 22 * 22 + 3 == 488
    assert sp[0]["radius"] == 1000 and sp[-1]["kind"] == 1 and sp[-3]["kind"] == 2
    with pytest.raises(ValueError):
        orc.setup_scene(17)  # UserError.SceneNotAvailable
    kinds = np.bincount(sp["kind"][1:485], minlength=3) / 484.0
    assert 0.7 < kinds[0] < 0.9 and 0.08 < kinds[1] < 0.22 and 0.01 < kinds[2] < 0.1


# ---- multithreading.zig:46-51 ---------------------------------------------------------------
def test_chunk_size(orc):
    n = orc.lib().orc_chunk_size(640 * 480, 4)
    assert 4 < n < 640 * 480 and n == 76800
    assert orc.lib().orc_chunk_size(10, 4) == 3


# ---- utils.zig:22-111 -----------------------------------------------------------------------
def test_num_digits(orc):
    assert orc.lib().orc_num_digits(123) == 3


def test_col_num_to_string(orc):
    buf = C.create_string_buffer(8)
    assert orc.lib().orc_col_num_to_string(128, 129, buf, 8) == -6  # ValueExceedsMaxValue
    assert orc.lib().orc_col_num_to_string(255, 42, buf, 8) == 0 and buf.value == b" 42"


# ---- render_targets/ppm_image.zig:106-175 ---------------------------------------------------
def test_image_px_size_and_header(orc):
    img = orc.image_new(640)
    assert img.px_size == 12 and img.height == 360
    assert orc.image_new(640, max_px_value=65535).px_size == 18
    buf = C.create_string_buffer(64)
    n = orc.lib().orc_image_header(C.byref(img), buf, 64)
    assert n == 15 and buf.raw[:n] == b"P3\n640 360\n255\n" and img.header_size == 15


@pytest.mark.parametrize("width,height", [(256, 144), (400, 225), (640, 360), (1920, 1080), (3840, 2160)])
def test_image_heights(orc, width, height):  # f32 division + truncation
    assert orc.image_new(width).height == height


def test_rgb_to_ascii(orc):
    img = orc.image_new(640)
    buf = C.create_string_buffer(32)
    assert orc.lib().orc_rgb_to_ascii(C.byref(img), f3((0.5, 0.7, 1.0)), buf, 32) == 0
    assert buf.value == b"  0   0   1\n"
    assert orc.lib().orc_rgb_to_ascii(C.byref(img), f3((255, 10, 0)), buf, 32) == 0
    assert buf.value == b"255  10   0\n"


# ---- RNG algorithms: published known answers ---------------------------------------------------
def test_philox4x32_10_kat(orc):
    """Random123 kat_vectors for philox4x32-10."""
    def run(ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        orc.lib().orc_philox4x32_10(c, k, o)
        return tuple(o)
    assert run((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert run((0xffffffff,) * 4, (0xffffffff,) * 2) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert run((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_xoshiro256pp_kat(orc):
    """xoshiro256++ from state (1,2,3,4): the published reference sequence (also the vector in
    Zig's std/rand/Xoshiro256.zig test and in rand_xoshiro)."""
    r = orc.ZigRng()
    r.kind = orc.PRNG_XOSHIRO256
    r.s[0], r.s[1], r.s[2], r.s[3] = 1, 2, 3, 4
    got = [orc.lib().orc_zig_next(C.byref(r)) for _ in range(10)]
    assert got == [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205,
                   9973669472204895162, 14011001112246962877, 12406186145184390807, 15849039046786891736,
                   10450023813501588000]


def test_xoroshiro128p_kat(orc):
    """Xoroshiro128+ (rotations 55/14/36): the known-answer sequence of Zig std's own test "xoroshiro sequence"
    (state 0xaeecf86f7878dd75, 0x01cd153642e72622) — the DefaultPrng generation behind images/demo.png."""
    r = orc.ZigRng()
    r.kind = orc.PRNG_XOROSHIRO128
    r.s[0], r.s[1] = 0xaeecf86f7878dd75, 0x01cd153642e72622
    got = [orc.lib().orc_zig_next(C.byref(r)) for _ in range(6)]
    assert got == [0xb0ba0da5bb600397, 0x18a08afde614dccc, 0xa2635b956a31b929, 0xabe633c971efa045,
                   0x9ac19f9706ca3cac, 0xf62b426578c1e3fb]


def test_default_prng_generations(orc):
    """DefaultPrng.init(seed): Xoroshiro128 takes two SplitMix64 words, Xoshiro256 four; the default is the
    demo-pinned Xoroshiro128+ and the two build different scene-21 tables from the same seed."""
    s = C.c_uint64(456)
    words = [orc.lib().orc_splitmix64_next(C.byref(s)) for _ in range(4)]
    a, b, d = orc.ZigRng(), orc.ZigRng(), orc.ZigRng()
    orc.lib().orc_zig_seed_kind(C.byref(a), 456, orc.PRNG_XOROSHIRO128)
    orc.lib().orc_zig_seed_kind(C.byref(b), 456, orc.PRNG_XOSHIRO256)
    orc.lib().orc_zig_seed(C.byref(d), 456)
    assert list(a.s) == words[:2] + [0, 0] and list(b.s) == words and (list(d.s), d.kind) == (list(a.s), a.kind)
    t0, t1, td = (orc.setup_scene(21, 456, prng=k)[0] for k in ("xoroshiro128", "xoshiro256", None))
    assert len(t0) == len(t1) == 488 and t0.tobytes() == td.tobytes() and t0.tobytes() != t1.tobytes()


def test_splitmix64_kat(orc):
    """SplitMix64 (Vigna's reference) from seed 1234567."""
    s = C.c_uint64(1234567)
    got = [orc.lib().orc_splitmix64_next(C.byref(s)) for _ in range(5)]
    assert got == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431,
                   16408922859458223821]


def test_zig_float_bits(orc):
    # Random.float(f32): [1,2) from the top 23 bits, minus 1 -> multiples of 2^-23 in [0,1)
    assert orc.lib().orc_u32_to_float(0) == 0.0
    assert orc.lib().orc_u32_to_float(0xffffffff) == 1.0 - 2.0 ** -23
    assert orc.lib().orc_u32_to_float(0x80000000) == 0.5


# ---- structure of the restatement --------------------------------------------------------------
@pytest.mark.parametrize("n_scene,mode", [(18, 0), (21, 0), (21, 1), (20, 0)])
def test_iterative_equals_recursive(orc, n_scene, mode):
    """The GPU's bounce loop + innermost-first unwinding is the reference's recursion, bit for bit."""
    sp, _ = orc.setup_scene(n_scene, 456)
    cam = orc.camera_preset(n_scene)
    img = orc.image_new(48)
    cfg = orc.make_config(img, subpixels=2, rays_per_subsample=2, rebounds=50, mode=mode, seed=3)
    a, b = f3((0, 0, 0)), f3((0, 0, 0))
    for idx in range(0, img.width * img.height, 7):
        for s in range(4):
            ra = orc.lib().orc_sample_radiance(sp.ctypes.data, len(sp), C.byref(cam), C.byref(cfg), idx, s, 0, a)
            rb = orc.lib().orc_sample_radiance(sp.ctypes.data, len(sp), C.byref(cam), C.byref(cfg), idx, s, 1, b)
            assert ra == rb == 0 and tuple(a) == tuple(b)


def test_thread_split_matches_single_thread(orc):
    """renderMultiThread's ceil(P/T) contiguous split changes nothing in Philox mode (streams are per path)."""
    sp, _ = orc.setup_scene(21, 456)
    cam = orc.camera_preset(21)
    img = orc.image_new(40)
    cfg = orc.make_config(img, subpixels=4, rays_per_subsample=2, rebounds=6, seed=11)
    rc1, one, st1 = orc.render_sum(sp, cam, cfg, nthreads=1)
    rc7, many, st7 = orc.render_sum(sp, cam, cfg, nthreads=7)
    assert rc1 == rc7 == 0 and st1.rays == st7.rays and st1.paths == st7.paths == 40 * 22 * 8
    np.testing.assert_array_equal(one, many)


def test_chunked_sum_semantics(orc):
    """chunk = 0 is the reference's single sequential sum; chunked sums differ only by fp32 add order."""
    sp, _ = orc.setup_scene(18)
    cam = orc.camera_preset(18)
    img = orc.image_new(32)
    full = orc.render_sum(sp, cam, orc.make_config(img, subpixels=4, rays_per_subsample=4, seed=5), 4)[1]
    chunked = orc.render_sum(sp, cam, orc.make_config(img, subpixels=4, rays_per_subsample=4, seed=5, chunk=3), 4)[1]
    np.testing.assert_allclose(chunked, full, rtol=2e-6)
    # the two halves of a sample-range split add up to the whole (again up to add order)
    a = orc.render_sum(sp, cam, orc.make_config(img, subpixels=4, rays_per_subsample=4, seed=5, sample_start=0, sample_stop=8), 4)[1]
    b = orc.render_sum(sp, cam, orc.make_config(img, subpixels=4, rays_per_subsample=4, seed=5, sample_start=8, sample_stop=16), 4)[1]
    np.testing.assert_allclose(a + b, full, rtol=2e-6)


def test_tile_split_is_a_partition(orc):
    sp, _ = orc.setup_scene(18)
    cam = orc.camera_preset(18)
    img = orc.image_new(40)
    full = orc.render_sum(sp, cam, orc.make_config(img, subpixels=1, rays_per_subsample=2, seed=5), 4)[1]
    acc = np.zeros_like(full)
    for g in range(3):
        part = orc.render_sum(sp, cam, orc.make_config(img, subpixels=1, rays_per_subsample=2, seed=5, tile_pixels=16,
                                                       tile_count=3, tile_index=g), 4)[1]
        assert not (acc != 0)[part != 0].any()  # disjoint supports
        acc += part
    np.testing.assert_array_equal(acc, full)


def test_zig_mode_render_runs(orc):
    """main.zig's single-thread path with the Zig RNG emulation (unpinned): it must at least produce
    a plausible image and consume the scene stream first."""
    sp, rng = orc.setup_scene(21, 456)
    cam = orc.camera_preset(21)
    img = orc.image_new(24)
    cfg = orc.make_config(img, subpixels=2, rays_per_subsample=1, mode=2)
    out = np.zeros((img.height, img.width, 3), dtype=np.float32)
    st = orc.Stats()
    rc = orc.lib().orc_render_sum_zig(sp.ctypes.data, len(sp), C.byref(cam), C.byref(cfg), C.byref(rng), out.ctypes.data,
                                      C.byref(st))
    assert rc == 0 and st.paths == 24 * 13 * 2
    mean = out / 2.0
    assert 0.0 <= mean.min() and mean.max() <= 1.0 and mean[0].mean() > mean[-1].mean() * 0.8


def test_philox_streams_converge_to_the_reference_sampler(orc):
    """The generator swap is unbiased: the Philox-stream estimate (what the GPU computes, bit for bit) and the
    reference's own sequential std.rand sampler (orc_render_sum_zig: main.zig:166-177 + material.zig:15) agree
    better the more paths either takes.  24x13 pixels of scene 21, depth 50; the GPU-side test at 48x27 / 4096 paths
    is tests/test_gpu_baseline_configs.py::test_convergence_against_the_reference_sampler."""
    cam = orc.camera_preset(21)
    img = orc.image_new(24)
    sp, _ = orc.setup_scene(21, 456)
    cfgp = orc.make_config(img, subpixels=4, rays_per_subsample=256, rebounds=50, seed=1, chunk=16)
    rc, s, _ = orc.render_sum(sp, cam, cfgp, 8)
    assert rc == 0
    p = orc.resolve_mean(cfgp, s).astype(np.float64)
    psnr = {}
    for rps in (16, 256):
        sp2, stream = orc.setup_scene(21, 456)
        cfg = orc.make_config(img, subpixels=4, rays_per_subsample=rps, rebounds=50, mode=2)
        rc, z, st = orc.render_sum_zig(sp2, cam, cfg, stream)
        assert rc == 0 and st.paths == 24 * 13 * 4 * rps
        z = orc.resolve_mean(cfg, z).astype(np.float64)
        psnr[4 * rps] = 10.0 * np.log10(1.0 / float(np.mean((p - z) ** 2)))
    assert psnr[1024] >= 36.0 and psnr[1024] >= psnr[64] + 6.0, psnr  # expected gain 10 log10(17/2) = 9.3 dB
