"""The library's lean correctly rounded sequences (csrc/pt_device.h: sqrt_rn, rcp_rn, inv_sqrt_rn, not_unit_len2) equal the
compiler's IEEE-correct expressions on EVERY f32 bit pattern — proof by exhaustion, re-run with the test suite — and equal
numpy's correctly rounded float32 sqrt / division (the CPU side of the same claim) on a few million patterns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which,name", [(0, "sqrtf(x)"), (1, "1.0f / x"), (2, "1.0f / sqrtf(x)"), (3, "|sqrt(a) - 1| > 1e-5 on a")])
def test_lean_sequence_equals_the_compilers_on_all_inputs(gpu, which, name):
    assert gpu.debug_exhaustive(which) == 0, name


def test_unit_vector_and_sphere_root_against_numpy(gpu):
    """vec3.zig:61-64 (unitVector) and sphere.zig:75-80 (the near root) as the device evaluates them, against numpy float32
    (IEEE sqrt and division) on random and edge inputs."""
    from path_tracer_amd import _lib
    rng = np.random.default_rng(3)
    n = 200_000
    f32 = np.float32
    o = rng.normal(0, 3, (n, 3)).astype(f32)
    d = rng.normal(0, 1, (n, 3)).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(f32)
    scale = (10.0 ** rng.uniform(-3, 3, (n, 1))).astype(f32)
    d[: n // 4] *= scale[: n // 4]                       # non-unit directions too
    c = (o + d * rng.uniform(0.2, 30, (n, 1)).astype(f32) + rng.normal(0, 0.3, (n, 3)).astype(f32)).astype(f32)
    r = (10.0 ** rng.uniform(-2, 1, (n, 1))).astype(f32)
    inp = np.concatenate([o, d, c, r, np.full((n, 1), 0.1, f32), np.full((n, 1), 1000.0, f32)], axis=1).astype(f32)
    out = gpu.debug_eval(_lib.PT_FN_SPHERE_HIT, inp)
    oc = o - c
    a = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
    hb = oc[:, 0] * d[:, 0] + oc[:, 1] * d[:, 1] + oc[:, 2] * d[:, 2]
    cc = (oc[:, 0] * oc[:, 0] + oc[:, 1] * oc[:, 1] + oc[:, 2] * oc[:, 2]) - r[:, 0] * r[:, 0]
    disc = hb * hb - a * cc
    with np.errstate(invalid="ignore"):
        t = (-hb - np.sqrt(disc)) / a
    hit = (disc >= 0) & (t < f32(1000.0)) & (t > f32(0.1))
    np.testing.assert_array_equal(out[:, 0] != 0, hit)
    assert hit.sum() > n // 10
    np.testing.assert_array_equal(out[hit, 1], t[hit])
    p = o + d * t[:, None]
    v = (p - c) * (f32(1.0) / r)
    inv = f32(1.0) / np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2])
    nn = v * inv[:, None]
    np.testing.assert_array_equal(out[hit, 2:5], p[hit])
    np.testing.assert_array_equal(out[hit, 5:8], nn[hit])
