"""A second, independent reading of the reference: the per-function hot-path pieces restated in numpy
float32 scalars straight from the Zig text (each function cites its lines), compared BIT FOR BIT with the
golden per-function vectors that the C oracle produced (tests/golden/per_function.npz) and that the GPU
tests check the HIP device functions against.  Two restatements written separately (C, then numpy)
agreeing on 400 random inputs per function is the strongest pin available while the Zig toolchain is
absent; numpy float32 arithmetic is IEEE single without fused multiply-add, like the reference's build.
"""
import os

import numpy as np
import pytest

f32 = np.float32
ONE, TWO, HALF = f32(1.0), f32(2.0), f32(0.5)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "per_function.npz")


@pytest.fixture(scope="module")
def vec():
    return np.load(GOLDEN)


# ---- vec3.zig ------------------------------------------------------------------------------------------
def v(x, y, z):
    return (f32(x), f32(y), f32(z))


def add(a, b):                                   # vec3.zig:41-43
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub(a, b):                                   # vec3.zig:46-48
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def mul(a, k):                                   # vec3.zig:51-57: k * component
    k = f32(k)
    return (k * a[0], k * a[1], k * a[2])


def length_squared(a):                           # vec3.zig:37-39: pow(x,2) + pow(y,2) + pow(z,2), left to right
    return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]


def norm(a):                                     # vec3.zig:30-32
    return np.sqrt(length_squared(a))


def unit_vector(a):                              # vec3.zig:60-63: multiply by the reciprocal, not divide
    inv_n = ONE / norm(a)
    return (inv_n * a[0], inv_n * a[1], inv_n * a[2])


def dot(a, b):                                   # vec3.zig:66-68
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


# ---- sphere.zig:63-103 ---------------------------------------------------------------------------------
def sphere_is_hit(center, radius, o, d, t_min, t_max):
    oc = sub(o, center)
    a = length_squared(d)
    half_b = dot(oc, d)
    c = length_squared(oc) - radius * radius
    discriminant = half_b * half_b - a * c
    if discriminant < f32(0.0):
        return None
    sqrtd = np.sqrt(discriminant)
    t = (-half_b - sqrtd) / a                    # near root only
    if t < t_max and t > t_min:
        p = add(o, mul(d, t))                    # ray.zig:18 pointAt
        n = unit_vector(mul(sub(p, center), ONE / radius))
        return t, p, n
    return None


# ---- material.zig ---------------------------------------------------------------------------------------
def reflect(d, n):                               # :59-65
    return sub(d, mul(n, TWO * dot(d, n)))


def refract(d, n, n1_over_n2):                   # :98-116
    cos_theta = min(dot(mul(d, -1), n), ONE)
    r_perp = mul(add(d, mul(n, cos_theta)), n1_over_n2)
    discriminant = ONE - length_squared(r_perp)
    if discriminant < 0:
        return None
    r_par = mul(n, f32(-1) * np.sqrt(discriminant))
    return add(r_perp, r_par)


def pow5(x):                                     # std.math.pow(f32, x, 5.0): square-and-multiply on the mantissa, i = 101b
    return x * ((x * x) * (x * x))


def schlick(cosine, ri):                         # :120-124
    r0 = (ONE - ri) / (ONE + ri)
    r0 = r0 * r0
    return r0 + (ONE - r0) * pow5(ONE - cosine)


class Draws:
    """The uniform draws a function consumes, in order; past the supplied ones every draw is 0.5 (the
    convention of the per-function entries: include/ptgpu.h, oracle xi arrays)."""

    def __init__(self, xi):
        self.xi, self.n = xi, 0

    def next(self):
        x = f32(self.xi[self.n]) if self.n < len(self.xi) else HALF
        self.n += 1
        return x


def random_vector_in_unit_sphere(r):             # :27-34 — components in [0,1): the positive octant
    while True:
        p = (r.next(), r.next(), r.next())
        if length_squared(p) < ONE:
            return p


def scatter(kind, param, d, p, n, r):            # :43-51, :79-88, :129-167
    if kind == 0:
        target = add(p, add(n, random_vector_in_unit_sphere(r)))
        return unit_vector(sub(target, p))
    if kind == 1:
        sc = reflect(d, n)
        fc = mul(random_vector_in_unit_sphere(r), param)
        return unit_vector(add(sc, fc))
    if dot(d, n) > f32(0.0):
        outward = unit_vector(v(-n[0], -n[1], -n[2]))
        n1_over_n2 = param
        cosine = param * dot(d, n) / norm(d)
    else:
        outward = n
        n1_over_n2 = ONE / param
        cosine = -dot(d, n) / norm(d)
    refracted = refract(d, outward, n1_over_n2)
    if refracted is not None:
        if r.next() < schlick(cosine, param):
            return reflect(d, n)                 # about hit.n, not outward_normal
        return refracted
    return reflect(d, n)


# ---- illumination.zig:27-34 -----------------------------------------------------------------------------
def lerp(u, p0, p1):
    if abs(norm(u) - ONE) > f32(0.00001):
        return None
    t = (u[1] + ONE) * HALF
    return add(mul(p0, ONE - t), mul(p1, t))


# ---- camera.zig:99-120 ----------------------------------------------------------------------------------
def random_vector_in_unit_disk(r):
    while True:
        p = (TWO * r.next() - ONE, TWO * r.next() - ONE, f32(0.0))
        if length_squared(p) < ONE:
            return p


def cast_ray(cam, s, t, r):
    rd = mul(random_vector_in_unit_disk(r), cam["lens_radius"])
    offset = add(mul(cam["u"], rd[0]), mul(cam["v"], rd[1]))
    origin = add(cam["lookfrom"], offset)
    direction = unit_vector(sub(sub(add(add(cam["llc"], mul(cam["horizontal"], s)), mul(cam["vertical"], t)), cam["lookfrom"]), offset))
    return origin, direction


def _eq(got, want):
    np.testing.assert_array_equal(np.asarray(got, dtype=f32).view(np.uint32), np.asarray(want, dtype=f32).view(np.uint32))


# ---- the comparisons --------------------------------------------------------------------------------------
def test_sphere_is_hit(vec):
    hits = 0
    for row, out in zip(vec["sphere_hit_in"], vec["sphere_hit_out"]):
        r = sphere_is_hit(tuple(row[6:9]), row[9], tuple(row[0:3]), tuple(row[3:6]), row[10], row[11])
        assert (r is not None) == bool(out[0])
        if r is not None:
            hits += 1
            _eq([r[0], *r[1], *r[2]], out[1:8])
    assert 50 < hits < 350


def test_reflect_refract_schlick_lerp(vec):
    for row, out in zip(vec["reflect_in"], vec["reflect_out"]):
        _eq(reflect(tuple(row[0:3]), tuple(row[3:6])), out)
    for row, out in zip(vec["refract_in"], vec["refract_out"]):
        r = refract(tuple(row[0:3]), tuple(row[3:6]), row[6])
        assert (r is not None) == bool(out[0])
        if r is not None:
            _eq(r, out[1:4])
    for row, out in zip(vec["schlick_in"], vec["schlick_out"]):
        _eq([schlick(row[0], row[1])], out)
    for row, out in zip(vec["lerp_in"], vec["lerp_out"]):
        r = lerp(tuple(row[0:3]), tuple(row[3:6]), tuple(row[6:9]))
        assert (r is None) == bool(out[0])
        if r is not None:
            _eq(r, out[1:4])


def test_scatter_all_three_materials(vec):
    kinds = set()
    for row, out in zip(vec["scatter_in"], vec["scatter_out"]):
        r = Draws(row[11:])
        kind = int(row[9])
        kinds.add(kind)
        _eq(scatter(kind, row[10], tuple(row[0:3]), tuple(row[3:6]), tuple(row[6:9]), r), out[0:3])
        assert r.n == int(out[3])               # the same number of draws consumed
    assert kinds == {0, 1, 2}


def test_cast_ray(vec, orc):
    c = orc.camera_preset(21)                    # main.zig:127-131 via camera.zig:48-96 (the oracle's Camera.new)
    cam = {"lookfrom": tuple(f32(x) for x in c.lookfrom), "u": tuple(f32(x) for x in c.u), "v": tuple(f32(x) for x in c.v),
           "llc": tuple(f32(x) for x in c.lower_left_corner), "horizontal": tuple(f32(x) for x in c.horizontal),
           "vertical": tuple(f32(x) for x in c.vertical), "lens_radius": f32(c.lens_radius)}
    for row, out in zip(vec["cast_ray_in"], vec["cast_ray_out"]):
        r = Draws(row[2:])
        o, d = cast_ray(cam, row[0], row[1], r)
        _eq([*o, *d], out[0:6])
        assert r.n == int(out[6])


def test_scene_is_hit_first_sphere_wins(vec, orc):
    """scene.zig:96-110: every sphere against the full (t_min, t_max), strict '<' keeps the first on ties."""
    spheres, _ = orc.setup_scene(21, 456)
    centers = [(f32(s["cx"]), f32(s["cy"]), f32(s["cz"])) for s in spheres]
    radii = [f32(s["radius"]) for s in spheres]
    for row, out in list(zip(vec["scene_hit_in"], vec["scene_hit_out"]))[:60]:   # 60 rays x 488 spheres in Python
        o, d = tuple(row[0:3]), tuple(row[3:6])
        best, closest = -1, f32(1000.0)
        rec = None
        for j, (c, rad) in enumerate(zip(centers, radii)):
            h = sphere_is_hit(c, rad, o, d, f32(0.1), f32(1000.0))
            if h is not None and h[0] < closest:
                best, closest, rec = j, h[0], h
        assert best == int(out[0])
        if best >= 0:
            _eq([rec[0], *rec[1], *rec[2]], out[1:8])


# ---- image level: render() + radiance, recursive as the reference (deterministic mode) -----------------------
class Half:
    """Deterministic mode: every draw is 0.5."""

    def next(self):
        return HALF


def radiance(spheres, ray, t_min, t_max, blend_start, blend_stop, num_rebounds):   # illumination.zig:76-90
    o, d = ray
    best, closest = None, t_max
    for s in spheres:                                                               # scene.zig:96-110
        h = sphere_is_hit(s["c"], s["r"], o, d, t_min, t_max)
        if h is not None and h[0] < closest:
            best, closest = (s, h), h[0]
    if best is None:
        return lerp(d, blend_start, blend_stop)
    if num_rebounds <= 0:
        return v(0, 0, 0)
    s, (t, p, n) = best
    nd = scatter(s["kind"], s["param"], d, p, n, Half())
    atten = v(1, 1, 1) if s["kind"] == 2 else s["albedo"]
    col = radiance(spheres, (p, nd), t_min, t_max, blend_start, blend_stop, num_rebounds - 1)
    return (col[0] * atten[0], col[1] * atten[1], col[2] * atten[2])              # v.elementwiseMul(s.attenuation)


@pytest.mark.parametrize("depth", [6, 50])
def test_deterministic_frame_scene18(orc, depth):
    """renderers/ppm_image.zig:49-69 with one sample per pixel, in numpy, against the golden frame the oracle
    produced and the GPU reproduces (tests/golden/images.npz: det_scene18_w32_d*)."""
    want = np.load(os.path.join(os.path.dirname(GOLDEN), "images.npz"))["det_scene18_w32_d%d" % depth]
    sp, _ = orc.setup_scene(18, 456)
    spheres = [{"c": v(s["cx"], s["cy"], s["cz"]), "r": f32(s["radius"]), "kind": int(s["kind"]), "param": f32(s["param"]),
                "albedo": v(s["ar"], s["ag"], s["ab"])} for s in sp]
    c = orc.camera_preset(18)
    cam = {"lookfrom": tuple(f32(x) for x in c.lookfrom), "u": tuple(f32(x) for x in c.u), "v": tuple(f32(x) for x in c.v),
           "llc": tuple(f32(x) for x in c.lower_left_corner), "horizontal": tuple(f32(x) for x in c.horizontal),
           "vertical": tuple(f32(x) for x in c.vertical), "lens_radius": f32(c.lens_radius)}
    H, W = want.shape[:2]
    white, blue = v(1, 1, 1), v(0.5, 0.7, 1.0)                                     # render_targets/ppm_image.zig:69-70
    got = np.zeros_like(want)
    for idx in range(H * W):
        i_col, i_row = idx % W, idx // W
        u = (f32(i_col) + HALF) / f32(W)
        vv = (f32(H - i_row + 1) + HALF) / f32(H)                                  # sic: + 1
        ray = cast_ray(cam, u, vv, Half())
        col = radiance(spheres, ray, f32(0.1), f32(1000.0), white, blue, depth)
        got[i_row, i_col] = col                                                    # one sample: 0 + col
    _eq(got, want)
