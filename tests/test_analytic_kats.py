"""Closed-form known answers for the functions of the path — cases whose results follow from the reference's formulas
by hand (exactly representable, or one obvious f32 expression), independent of any restatement: they pin the ORACLE
(CPU) and, through ptgpu_debug_eval, the DEVICE functions (GPU) against the same numbers.  The reference's own test
blocks pin little here (a miss for Sphere.is_hit, a unit-length property for refract, nothing for scatter)."""
import ctypes as C

import numpy as np
import pytest

F = np.float32


def f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


def _sphere(orc, c, r, kind=0, albedo=(1, 1, 1), param=0.0):
    sp = np.zeros(1, dtype=orc.SPHERE_DTYPE)
    sp[0] = (c[0], c[1], c[2], r, kind, albedo[0], albedo[1], albedo[2], param)
    return sp


INV_SQRT_2_75 = F(1.0) / np.sqrt(F(2.75))        # vec3.zig:61-64: multiply by the reciprocal of the norm

# (d, n) -> reflect: d - n*(2*d.n), material.zig:59-65
REFLECT = [((1, -1, 0), (0, 1, 0), (1, 1, 0)), ((0, 0, -1), (0, 0, 1), (0, 0, 1)), ((3, 4, 0), (1, 0, 0), (-3, 4, 0))]
# (d, n, eta) -> refract: straight through an interface, any eta; material.zig:98-116
REFRACT = [((0, -1, 0), (0, 1, 0), 1.0, (0, -1, 0)), ((0, -1, 0), (0, 1, 0), 0.5, (0, -1, 0)), ((0, 0, 1), (0, 0, -1), 1.5, (0, 0, 1))]
# sphere.zig:63-103: (o, d, c, r) -> t, p, n ; a negative radius flips the normal (scene.zig:48)
SPHERE_HIT = [((0, 0, 0), (0, 0, -1), (0, 0, -1), 0.5, 0.5, (0, 0, -0.5), (0, 0, 1)),
              ((0, 0, 0), (0, 0, -1), (0, 0, -1), -0.5, 0.5, (0, 0, -0.5), (0, 0, -1)),
              ((0, 0, 4), (0, 0, -2), (0, 0, 0), 1.0, 1.5, (0, 0, 1), (0, 0, 1)),          # |d| = 2: t = (-hb - sqrt(disc)) / a
              ((-3, 0, 0), (1, 0, 0), (0, 0, 0), 2.0, 1.0, (-2, 0, 0), (-1, 0, 0))]


def _r0(ri):
    r0 = (F(1.0) - F(ri)) / (F(1.0) + F(ri))
    return F(r0 * r0)


# ---- the oracle -------------------------------------------------------------------------------------------
def test_oracle_reflect_refract_schlick_lerp(orc):
    L = orc.lib()
    for d, n, want in REFLECT:
        out = f3((0, 0, 0))
        L.orc_reflect(f3(d), f3(n), out)
        assert tuple(out) == tuple(map(float, want))
    for d, n, eta, want in REFRACT:
        out = f3((9, 9, 9))
        assert L.orc_refract(f3(d), f3(n), eta, out) == 1
        assert tuple(abs(v) for v in out) == tuple(abs(float(v)) for v in want) and sum(a * b for a, b in zip(out, want)) == 1.0
    # grazing incidence from the dense side: total internal reflection (null)
    assert L.orc_refract(f3((0.8, -0.6, 0)), f3((0, 1, 0)), 1.5, f3((0, 0, 0))) == 0
    # material.zig:120-124: schlick(1, ri) = r0, schlick(0, ri) = r0 + (1 - r0)
    for ri in (1.5, 2.0, 1.0):
        assert L.orc_schlick(1.0, ri) == _r0(ri)
        assert L.orc_schlick(0.0, ri) == F(_r0(ri) + (F(1.0) - _r0(ri)))
    # illumination.zig:27-34: straight up = blend_stop, straight down = blend_start, the horizon their mean
    white, blue = (1.0, 1.0, 1.0), (0.5, 0.75, 1.0)
    out = f3((0, 0, 0))
    assert L.orc_lerp(f3((0, 1, 0)), f3(white), f3(blue), out) == 0 and tuple(out) == blue
    assert L.orc_lerp(f3((0, -1, 0)), f3(white), f3(blue), out) == 0 and tuple(out) == white
    assert L.orc_lerp(f3((1, 0, 0)), f3(white), f3(blue), out) == 0 and tuple(out) == (0.75, 0.875, 1.0)
    assert L.orc_lerp(f3((0, 1.001, 0)), f3(white), f3(blue), out) != 0       # DirectionIsNotUnitVector


def test_oracle_sphere_and_scene_hits(orc):
    L = orc.lib()
    for o, d, c, r, t_want, p_want, n_want in SPHERE_HIT:
        t = C.c_float()
        p, n = f3((0, 0, 0)), f3((0, 0, 0))
        assert L.orc_sphere_hit(_sphere(orc, c, r).ctypes.data, f3(o), f3(d), 0.001, 1000.0, C.byref(t), p, n) == 1
        assert t.value == t_want and tuple(p) == tuple(map(float, p_want)) and tuple(n) == tuple(map(float, n_want))
    # t outside (t_min, t_max) is no hit; the interval is open (sphere.zig:86: t < t_max and t > t_min)
    t = C.c_float()
    sp = _sphere(orc, (0, 0, -1), 0.5)
    for t_min, t_max, want in ((0.5, 1000.0, 0), (0.001, 0.5, 0), (0.25, 0.75, 1)):
        assert L.orc_sphere_hit(sp.ctypes.data, f3((0, 0, 0)), f3((0, 0, -1)), t_min, t_max, C.byref(t), f3((0, 0, 0)), f3((0, 0, 0))) == want
    # scene.zig:102 strict '<': of two coincident spheres the first wins; a nearer one later in the list wins over both
    two = np.concatenate([_sphere(orc, (0, 0, -3), 1.0), _sphere(orc, (0, 0, -3), 1.0), _sphere(orc, (0, 0, -1), 0.25)])
    p, n = f3((0, 0, 0)), f3((0, 0, 0))
    assert L.orc_scene_hit(two.ctypes.data, 2, f3((0, 0, 0)), f3((0, 0, -1)), 0.001, 1000.0, C.byref(t), p, n) == 0 and t.value == 2.0
    assert L.orc_scene_hit(two.ctypes.data, 3, f3((0, 0, 0)), f3((0, 0, -1)), 0.001, 1000.0, C.byref(t), p, n) == 2 and t.value == 0.75


def _scatter(orc, sp, d, p, n, xi):
    used = C.c_uint32()
    att, so, sd = f3((0, 0, 0)), f3((0, 0, 0)), f3((0, 0, 0))
    xi = np.asarray(xi, dtype=np.float32)
    orc.lib().orc_scatter(sp.ctypes.data, f3(d), f3(p), f3(n), xi.ctypes.data_as(C.POINTER(C.c_float)), len(xi), C.byref(used), att, so, sd)
    return tuple(att), tuple(so), tuple(sd), used.value


def test_oracle_scatter(orc):
    p, n = (0, 0, -1), (0, 0, 1)
    # Lambertian (material.zig:43-51): unit((p + (n + rnd)) - p), attenuation = albedo, origin = p; the first draw triple with
    # |rnd|^2 < 1 is taken (material.zig:27-34): (0.9, 0.9, 0.9) is rejected, (0.5, 0.5, 0.5) accepted
    att, so, sd, used = _scatter(orc, _sphere(orc, (0, 0, -2), 1.0, 0, (0.25, 0.5, 0.75)), (0, 0, -1), p, n, [0.9, 0.9, 0.9, 0.5, 0.5, 0.5])
    want = tuple(float(INV_SQRT_2_75 * F(v)) for v in (0.5, 0.5, 1.5))
    assert att == (0.25, 0.5, 0.75) and so == (0.0, 0.0, -1.0) and sd == want and used == 6
    # Metal, fuzz 0 (material.zig:79-88): the mirror direction; the random vector is drawn all the same
    att, so, sd, used = _scatter(orc, _sphere(orc, (0, 0, -2), 1.0, 1, (0.5, 0.5, 0.5), 0.0), (0, 0, -1), p, n, [0.5, 0.5, 0.5])
    assert sd == (0.0, 0.0, 1.0) and used == 3 and att == (0.5, 0.5, 0.5)
    # Dielectric, head-on from outside (material.zig:129-167): refracted straight through unless xi < schlick = r0
    glass = _sphere(orc, (0, 0, -2), 1.0, 2, (1, 1, 1), 1.5)
    att, so, sd, used = _scatter(orc, glass, (0, 0, -1), p, n, [0.5])
    assert att == (1.0, 1.0, 1.0) and tuple(abs(v) for v in sd) == (0.0, 0.0, 1.0) and sd[2] == -1.0 and used == 1
    att, so, sd, used = _scatter(orc, glass, (0, 0, -1), p, n, [0.03])           # 0.03 < r0 = 0.04: reflected about hit.n
    assert sd == (0.0, 0.0, 1.0) and used == 1


# ---- the device functions ------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_device_functions_closed_form(gpu):
    from path_tracer_amd import _lib
    a = np.array([list(d) + list(n) for d, n, _ in REFLECT], dtype=F)
    np.testing.assert_array_equal(gpu.debug_eval(_lib.PT_FN_REFLECT, a), np.array([w for _, _, w in REFLECT], dtype=F))
    a = np.array([list(d) + list(n) + [eta] for d, n, eta, _ in REFRACT] + [[0.8, -0.6, 0, 0, 1, 0, 1.5]], dtype=F)
    out = gpu.debug_eval(_lib.PT_FN_REFRACT, a)
    assert list(out[:, 0]) == [1.0, 1.0, 1.0, 0.0]
    np.testing.assert_array_equal(np.abs(out[:3, 1:]), np.abs(np.array([w for *_, w in REFRACT], dtype=F)))
    a = np.array([[1.0, 1.5], [0.0, 1.5], [1.0, 2.0], [0.0, 1.0]], dtype=F)
    want = [_r0(1.5), F(_r0(1.5) + (F(1.0) - _r0(1.5))), _r0(2.0), F(1.0)]
    np.testing.assert_array_equal(gpu.debug_eval(_lib.PT_FN_SCHLICK, a)[:, 0], np.array(want, dtype=F))
    white, blue = [1.0, 1.0, 1.0], [0.5, 0.75, 1.0]
    a = np.array([[0, 1, 0] + white + blue, [0, -1, 0] + white + blue, [1, 0, 0] + white + blue, [0, 1.001, 0] + white + blue], dtype=F)
    out = gpu.debug_eval(_lib.PT_FN_LERP, a)
    assert list(out[:, 0]) == [0.0, 0.0, 0.0, 1.0]
    np.testing.assert_array_equal(out[:3, 1:], np.array([blue, white, [0.75, 0.875, 1.0]], dtype=F))
    a = np.array([list(o) + list(d) + list(c) + [r, 0.001, 1000.0] for o, d, c, r, *_ in SPHERE_HIT], dtype=F)
    out = gpu.debug_eval(_lib.PT_FN_SPHERE_HIT, a)
    want = np.array([[1.0, t] + list(p) + list(n) for *_, t, p, n in SPHERE_HIT], dtype=F)
    np.testing.assert_array_equal(out, want)
    # scatter: in d, p, n, kind, param, xi[16]  out: scattered d, n_used
    xi = lambda v: list(v) + [0.5] * (16 - len(v))  # noqa: E731
    a = np.array([[0, 0, -1, 0, 0, -1, 0, 0, 1, 0, 0.0] + xi([0.9, 0.9, 0.9, 0.5, 0.5, 0.5]),
                  [0, 0, -1, 0, 0, -1, 0, 0, 1, 1, 0.0] + xi([0.5, 0.5, 0.5]),
                  [0, 0, -1, 0, 0, -1, 0, 0, 1, 2, 1.5] + xi([0.5]),
                  [0, 0, -1, 0, 0, -1, 0, 0, 1, 2, 1.5] + xi([0.03])], dtype=F)
    out = gpu.debug_eval(_lib.PT_FN_SCATTER, a)
    lam = [float(INV_SQRT_2_75 * F(v)) for v in (0.5, 0.5, 1.5)]
    np.testing.assert_array_equal(out[:, 3], np.array([6, 3, 1, 1], dtype=F))
    np.testing.assert_array_equal(out[0, :3], np.array(lam, dtype=F))
    np.testing.assert_array_equal(out[1, :3], np.array([0, 0, 1], dtype=F))
    assert out[2, 2] == -1.0 and out[3, 2] == 1.0 and not out[2:, :2].any()
    # scene.zig:102 through both scans: coincident spheres, the first wins
    sp = np.zeros(3, dtype=_lib.SPHERE_DTYPE)
    sp[0] = sp[1] = (0, 0, -3, 1.0, 0, 1, 1, 1, 0)
    sp[2] = (0, 5, 0, 0.25, 0, 1, 1, 1, 0)
    gpu.set_scene(sp)
    rays = np.array([[0, 0, 0, 0, 0, -1]], dtype=F)
    for fn in (_lib.PT_FN_SCENE_HIT, _lib.PT_FN_SCENE_HIT_FILTERED):
        out = gpu.debug_eval(fn, rays, 0.001, 1000.0)
        assert out[0, 0] == 0.0 and out[0, 1] == 2.0
