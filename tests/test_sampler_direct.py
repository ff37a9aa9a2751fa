"""PT_SAMPLER_DIRECT / ORC_SAMPLER_DIRECT on the CPU: the opt-in samplers without rejection loops (NOT in the reference;
SURVEY.md 9.4-9.5 names the equivalents) must draw from the SAME distributions as the reference's loops
(camera.zig:113-120: the unit disk; material.zig:27-34: the positive octant of the unit ball).

Shown three ways, all on the oracle: (1) both samplers' outputs are uniform in coordinates in which the target
distribution is a product of uniforms (chi-square on a grid), with the moments of the closed forms; (2) frames rendered
with either sampler are two estimates of one image — their PSNR equals that of two independent rejection-sampler
estimates and rises with the sample count like theirs; (3) the committed vectors still come out of the oracle.
The GPU side (bit-equality with the oracle in direct mode) is tests/test_gpu_sampler_direct.py.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_direct.npz")


def _draws(n, k, seed):
    """values Random.float(f32) can take: multiples of 2^-23 in [0, 1)"""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 1 << 23, size=(n, k)).astype(np.float32) / np.float32(1 << 23)


def _chi2(counts):
    e = counts.sum() / counts.size
    return float(((counts - e) ** 2 / e).sum()), counts.size - 1


def _uniformity(u, bins):
    """chi-square statistic of points u in [0,1)^d on a bins^d grid, as a z-score (dof large: ~ normal)"""
    idx = np.minimum((u * bins).astype(np.int64), bins - 1)
    flat = np.ravel_multi_index(tuple(idx.T), (bins,) * u.shape[1])
    chi, dof = _chi2(np.bincount(flat, minlength=bins ** u.shape[1]).astype(np.float64))
    return (chi - dof) / np.sqrt(2.0 * dof)


@pytest.mark.parametrize("sampler", [0, 1])
def test_disk_is_uniform(orc, sampler):
    n = 400_000
    out = orc.sample_batch("disk", sampler, _draws(n, 48, 11 + sampler))
    p = out[:, :2].astype(np.float64)
    assert np.all(out[:, 2] == 0.0)
    r2 = (p ** 2).sum(1)
    assert r2.max() < 1.0 + 1e-6
    # uniform in the disk <=> (r^2, angle / 2 pi) uniform in the unit square
    u = np.stack([np.minimum(r2, 1.0 - 1e-12), (np.arctan2(p[:, 1], p[:, 0]) / (2 * np.pi)) % 1.0], axis=1)
    assert abs(_uniformity(u, 24)) < 4.0
    np.testing.assert_allclose(p.mean(0), [0, 0], atol=4e-3)
    np.testing.assert_allclose((p ** 2).mean(0), [0.25, 0.25], atol=2e-3)
    np.testing.assert_allclose(np.sqrt(r2).mean(), 2.0 / 3.0, atol=2e-3)
    if sampler == 1:
        assert np.all(out[:, 3] == 2.0)          # one try's worth of draws, always
    else:
        np.testing.assert_allclose(out[:, 3].mean(), 2.0 / (np.pi / 4.0), rtol=0.01)  # 2 draws per try, acceptance pi/4


@pytest.mark.parametrize("sampler", [0, 1])
def test_ball_octant_is_uniform(orc, sampler):
    n = 400_000
    out = orc.sample_batch("ball", sampler, _draws(n, 96, 21 + sampler))
    p = out[:, :3].astype(np.float64)
    assert p.min() >= 0.0                        # material.zig:29: components in [0, 1): the POSITIVE octant
    rho = np.sqrt((p ** 2).sum(1))
    assert rho.max() < 1.0 + 1e-6
    # uniform in the octant of the ball <=> (rho^3, cos(polar), azimuth / (pi/2)) uniform in the unit cube
    keep = rho > 1e-6
    u = np.stack([np.minimum(rho ** 3, 1.0 - 1e-12), np.minimum(p[:, 2] / np.maximum(rho, 1e-30), 1.0 - 1e-12),
                  np.minimum(np.arctan2(p[:, 1], p[:, 0]) / (np.pi / 2), 1.0 - 1e-12)], axis=1)[keep]
    assert abs(_uniformity(u, 10)) < 4.0
    np.testing.assert_allclose(p.mean(0), [0.375] * 3, atol=2e-3)        # 3/8
    np.testing.assert_allclose((p ** 2).mean(0), [0.2] * 3, atol=2e-3)   # 1/5
    np.testing.assert_allclose((p[:, 0] * p[:, 1]).mean(), 2.0 / (5.0 * np.pi), atol=2e-3)
    if sampler == 1:
        assert np.all(out[:, 3] == 3.0)
    else:
        np.testing.assert_allclose(out[:, 3].mean(), 3.0 / (np.pi / 6.0), rtol=0.01)  # 3 draws per try, acceptance pi/6


def test_two_sample_agreement_of_the_samplers(orc):
    """the two samplers against EACH OTHER: histograms over the same grid differ by sampling noise only"""
    n = 300_000
    for which, k, bins in (("disk", 48, 16), ("ball", 96, 8)):
        a = orc.sample_batch(which, 0, _draws(n, k, 5))[:, :3].astype(np.float64)
        b = orc.sample_batch(which, 1, _draws(n, k, 6))[:, :3].astype(np.float64)
        d = 2 if which == "disk" else 3
        lo = -1.0 if which == "disk" else 0.0
        def hist(p):
            idx = np.clip(((p[:, :d] - lo) / (1.0 - lo) * bins).astype(np.int64), 0, bins - 1)
            return np.bincount(np.ravel_multi_index(tuple(idx.T), (bins,) * d), minlength=bins ** d).astype(np.float64)
        ha, hb = hist(a), hist(b)
        m = (ha + hb) > 20
        chi = float(((ha - hb)[m] ** 2 / (ha + hb)[m]).sum())
        dof = int(m.sum()) - 1
        assert abs((chi - dof) / np.sqrt(2.0 * dof)) < 4.0, (which, chi, dof)


def test_direct_mode_frames_are_estimates_of_the_same_image(orc):
    """The PSNR test: oracle(rejection) vs oracle(direct) behaves like oracle(rejection, seed a) vs oracle(rejection, seed b):
    the same value within noise at every sample count, +3 dB per doubling — a biased sampler would show as a floor."""
    spheres, _ = orc.setup_scene(21, 456)
    cam = orc.camera_preset(21)
    img = orc.image_new(48)

    def mean(rps, seed, sampler):
        cfg = orc.make_config(img, subpixels=4, rays_per_subsample=rps, rebounds=50, seed=seed, sampler=sampler)
        rc, s, _ = orc.render_sum(spheres, cam, cfg, nthreads=8)
        assert rc == 0
        return orc.resolve_mean(cfg, s).astype(np.float64)

    def psnr(a, b):
        return 10.0 * np.log10(1.0 / float(np.mean((a - b) ** 2)))

    cross, same = {}, {}
    for rps in (32, 128, 512):
        rej_a, rej_b, dir_c = mean(rps, 1, 0), mean(rps, 2, 0), mean(rps, 3, 1)
        same[rps] = psnr(rej_a, rej_b)
        cross[rps] = 0.5 * (psnr(rej_a, dir_c) + psnr(rej_b, dir_c))
        assert abs(cross[rps] - same[rps]) < 1.0, (rps, cross, same)
    assert cross[512] >= 40.0, cross                                   # 2048 paths per pixel each
    assert cross[512] - cross[128] > 4.5 and cross[128] - cross[32] > 4.5, cross   # 6 dB per 4x, within noise


def test_direct_mode_is_not_the_rejection_stream_and_ignores_deterministic_mode(orc):
    spheres, _ = orc.setup_scene(18, 0)
    cam = orc.camera_preset(18)
    img = orc.image_new(32)
    a = orc.render_sum(spheres, cam, orc.make_config(img, rays_per_subsample=2, rebounds=50, seed=4, sampler=0))[1]
    b = orc.render_sum(spheres, cam, orc.make_config(img, rays_per_subsample=2, rebounds=50, seed=4, sampler=1))[1]
    assert not np.array_equal(a, b)
    da = orc.render_sum(spheres, cam, orc.make_config(img, subpixels=1, rebounds=50, mode=1, sampler=0))[1]
    db = orc.render_sum(spheres, cam, orc.make_config(img, subpixels=1, rebounds=50, mode=1, sampler=1))[1]
    np.testing.assert_array_equal(da, db)       # every draw is 0.5 and goes through the reference's expressions


def test_oracle_reproduces_the_committed_vectors(orc):
    g = np.load(GOLD)
    for which in ("disk", "ball"):
        for name, smp in (("rejection", 0), ("direct", 1)):
            inp = g["sample_%s_%s_in" % (which, name)]
            assert np.all(inp[:, 0] == smp)
            np.testing.assert_array_equal(orc.sample_batch(which, smp, inp[:, 1:]), g["sample_%s_%s_out" % (which, name)])
    for n_scene, width in ((18, 32), (21, 64)):
        spheres, _ = orc.setup_scene(n_scene, 456)
        cfg = orc.make_config(orc.image_new(width), subpixels=4, rays_per_subsample=2, rebounds=50, seed=456, sampler=1)
        rc, s, st = orc.render_sum(spheres, orc.camera_preset(n_scene), cfg, nthreads=4)
        assert rc == 0
        np.testing.assert_array_equal(s, g["direct_scene%d_w%d_d50_seed456" % (n_scene, width)])
        assert [st.paths, st.rays] == list(g["direct_scene%d_w%d_d50_seed456_rays" % (n_scene, width)])


def test_direct_sampler_edge_draws(orc):
    """the corner values a draw can take: 0 and 1 - 2^-23"""
    hi = np.float32(1.0) - np.float32(2.0 ** -23)
    xi = np.array([[0, 0, 0], [hi, hi, hi], [0, hi, 0], [hi, 0, hi], [0.5, 0.5, 0.5], [0.25, 0.75, 0.125]], dtype=np.float32)
    d = orc.sample_batch("disk", 1, xi)
    b = orc.sample_batch("ball", 1, xi)
    assert np.all(np.isfinite(d)) and np.all(np.isfinite(b))
    np.testing.assert_array_equal(d[0, :3], [0, 0, 0])                  # radius sqrt(0)
    np.testing.assert_array_equal(b[0, :3], [0, 0, 0])                  # radius 0^(1/3), guarded
    assert (d[:, :2].astype(np.float64) ** 2).sum(1).max() < 1.0 + 1e-6
    assert (b[:, :3].astype(np.float64) ** 2).sum(1).max() < 1.0 + 1e-6 and b[:, :3].min() >= 0.0
    # sqrt(1/4) * (cos, sin)(2 pi 3/4) = (0, -1/2) up to the polynomial's 1e-7
    np.testing.assert_allclose(d[5, :2], [0.0, -0.5], atol=2e-7)
