/*
 * pt_oracle.c — CPU oracle: plain-C restatement of jackdbd/path-tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off: the reference is Zig/LLVM without FP contraction, so
 * every a*b+c below is two roundings, as in the reference).
 *
 * Citations are file:line under /root/reference/src/.
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* vec3.zig:7-91                                                             */
/* ------------------------------------------------------------------------ */
typedef struct { float x, y, z; } vec3;

static inline vec3 v3(float x, float y, float z) { vec3 r = {x, y, z}; return r; } /* vec3.zig:21 */
static inline vec3 vadd(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); } /* vec3.zig:42 */
static inline vec3 vsub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); } /* vec3.zig:47 */
static inline vec3 vmul(vec3 a, float k) { return v3(k * a.x, k * a.y, k * a.z); }      /* vec3.zig:52 */
static inline float vdot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  /* vec3.zig:67 */
static inline vec3 vcross(vec3 a, vec3 b) {                                            /* vec3.zig:72 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline vec3 vemul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); } /* vec3.zig:81 */

/* Zig 0.9.1 std.math.pow(f32, x, y) for the two call shapes on the path:
 * y == 2 (vec3.zig:38) and y == 5 (material.zig:123).  The std implementation
 * (a port of Go's math.Pow) separates mantissa and exponent with frexp, then runs
 * square-and-multiply over the bits of the integer exponent on the mantissas and
 * rescales once.  Scaling by powers of two is exact, so away from
 * overflow/underflow the result equals the same square-and-multiply on the plain
 * floats: y=2 -> x*x; y=5 -> x * ((x*x)*(x*x)).  Special cases of std pow that can
 * occur here (x == 0 -> 0, x == 1 -> 1) give the same values as the products. */
float orc_pow(float x, float y) {
    if (y == 2.0f) return x * x;
    if (y == 5.0f) { float x2 = x * x; float x4 = x2 * x2; return x * x4; }
    return powf(x, y); /* not on the path */
}
static inline float zpow2(float x) { return x * x; }

static inline float vlen2(vec3 a) { return zpow2(a.x) + zpow2(a.y) + zpow2(a.z); } /* vec3.zig:37-39 */
static inline float vlen(vec3 a) { return sqrtf(vlen2(a)); }                       /* vec3.zig:30-32 */
static inline vec3 vunit(vec3 a) {                                                 /* vec3.zig:61-64 */
    const float inv_n = 1.0f / vlen(a);
    return v3(inv_n * a.x, inv_n * a.y, inv_n * a.z);
}

float orc_length(const float v[3]) { return vlen(v3(v[0], v[1], v[2])); }
float orc_length_squared(const float v[3]) { return vlen2(v3(v[0], v[1], v[2])); }
void orc_unit_vector(const float v[3], float out[3]) { vec3 u = vunit(v3(v[0], v[1], v[2])); out[0] = u.x; out[1] = u.y; out[2] = u.z; }
float orc_dot(const float a[3], const float b[3]) { return vdot(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); }
void orc_cross(const float a[3], const float b[3], float out[3]) { vec3 c = vcross(v3(a[0], a[1], a[2]), v3(b[0], b[1], b[2])); out[0] = c.x; out[1] = c.y; out[2] = c.z; }

/* ray.zig:5-21 */
typedef struct { vec3 origin, direction; } ray_t;
static inline vec3 point_at(ray_t r, float t) { return vadd(r.origin, vmul(r.direction, t)); } /* ray.zig:18 */
void orc_point_at(const float o[3], const float d[3], float t, float out[3]) {
    ray_t r = {v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    vec3 p = point_at(r, t); out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

/* ------------------------------------------------------------------------ */
/* RNG                                                                        */
/* ------------------------------------------------------------------------ */
/* Zig 0.9.1 std/rand/SplitMix64 (recalled). */
uint64_t orc_splitmix64_next(uint64_t *state) {
    *state += 0x9e3779b97f4a7c15ULL;
    uint64_t z = *state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
/* std.rand.DefaultPrng.init(seed), two generations of Zig's std:
 *   ORC_PRNG_XOROSHIRO128 — Xoroshiro128+ (rotations 55, 14, 36), seeded with two SplitMix64 outputs: the DefaultPrng
 *     of the older std.  PINNED twice: by std's own "xoroshiro sequence" known-answer test (tests/test_oracle_kats.py)
 *     and by the reference's images/demo.png — scene 21 built from this generator with seed 456 reproduces the
 *     picture's 484 small spheres (tests/test_demo_pin.py); no other candidate does.
 *   ORC_PRNG_XOSHIRO256 — Xoshiro256++ seeded with four SplitMix64 outputs: Zig 0.9.1's DefaultPrng as recalled
 *     (pinned by the published xoshiro256++ vector only; no reference artefact made with it exists). */
void orc_zig_seed_kind(orc_zigrng *r, uint64_t seed, uint32_t kind) {
    uint64_t s = seed;
    memset(r, 0, sizeof *r);
    r->kind = kind;
    const int words = (kind == ORC_PRNG_XOSHIRO256) ? 4 : 2;
    for (int i = 0; i < words; i++) r->s[i] = orc_splitmix64_next(&s);
}
void orc_zig_seed(orc_zigrng *r, uint64_t seed) { orc_zig_seed_kind(r, seed, ORC_PRNG_DEFAULT); }
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t orc_zig_next(orc_zigrng *r) {
    uint64_t *s = r->s;
    if (r->kind == ORC_PRNG_XOSHIRO256) { /* xoshiro256++ (Blackman & Vigna) */
        const uint64_t res = rotl64(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl64(s[3], 45);
        return res;
    }
    /* xoroshiro128+ (2016 constants), std/rand Xoroshiro128.next() */
    const uint64_t s0 = s[0];
    uint64_t s1 = s[1];
    const uint64_t res = s0 + s1;
    s1 ^= s0;
    s[0] = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
    s[1] = rotl64(s1, 36);
    return res;
}
/* Random.float(f32) of Zig 0.9.1: int(u32) = the low 4 bytes of one next();
 * bits 0x3F800000 | (s >> 9) give [1,2); subtract 1 (recalled). */
float orc_u32_to_float(uint32_t s) {
    uint32_t repr = (0x7fu << 23) | (s >> 9);
    float f;
    memcpy(&f, &repr, 4);
    return f - 1.0f;
}
float orc_zig_float(orc_zigrng *r) { return orc_u32_to_float((uint32_t)orc_zig_next(r)); }

/* Philox4x32-10 (Salmon et al., SC'11), the counter-based generator the GPU path
 * keys by (seed, pixel, sample). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { RNG_PHILOX = 0, RNG_CONST = 1, RNG_ZIG = 2, RNG_ARRAY = 3 };
typedef struct rng_t {
    int kind;
    uint32_t key[2], ctr[4], buf[4], n; /* philox path stream: draw n = word n&3 of block n>>2 */
    orc_zigrng *z;
    const float *xi; uint32_t n_xi;     /* array */
    int direct;                         /* orc_config.sampler == ORC_SAMPLER_DIRECT (never in deterministic mode) */
} rng_t;

static float rng_float(rng_t *r) {
    switch (r->kind) {
    case RNG_CONST: r->n++; return 0.5f;
    case RNG_ZIG: r->n++; return orc_zig_float(r->z);
    case RNG_ARRAY: { float v = (r->n < r->n_xi) ? r->xi[r->n] : 0.5f; r->n++; return v; }
    default: {
        if ((r->n & 3u) == 0) { r->ctr[2] = r->n >> 2; orc_philox4x32_10(r->ctr, r->key, r->buf); }
        float v = orc_u32_to_float(r->buf[r->n & 3u]);
        r->n++;
        return v;
    }
    }
}
/* Philox path streams only: k consecutive draws never straddle a block — if the current
 * block has fewer than k words left, skip them.  Applied to the (u,v) jitter pair, each disk
 * try (k = 2) and each unit-sphere try (k = 3), so that a try costs the GPU one Philox call at
 * one program point.  Other generators (Zig emulation, constant, array) are plain sequences. */
static void rng_reserve(rng_t *r, uint32_t k) {
    if (r->kind == RNG_PHILOX && (r->n & 3u) + k > 4u) r->n = (r->n + 3u) & ~3u;
}
static void rng_philox_path(rng_t *r, uint64_t seed, uint32_t pixel, uint32_t sample) {
    memset(r, 0, sizeof *r);
    r->kind = RNG_PHILOX;
    r->key[0] = (uint32_t)seed; r->key[1] = (uint32_t)(seed >> 32);
    r->ctr[0] = pixel; r->ctr[1] = sample; r->ctr[2] = 0; r->ctr[3] = 0;
}

/* The rejection loops of camera.zig:113-120 and material.zig:27-34 are unbounded in
 * the reference.  Oracle and GPU both stop after this many tries and keep the last
 * draw (probability < 1e-80 for the ball, never observed): a bound every wave
 * reaches, not a change of distribution. */
#define ORC_MAX_TRIES 256

/* ---- ORC_SAMPLER_DIRECT: the same two distributions without the rejection loops --------------
 * NOT in the reference: an opt-in of this build (orc_config.sampler), restated here so that the GPU's direct samplers
 * have an oracle.  Plain f32 multiplies, adds, one sqrtf and integer bit operations only — no libm sin/cos/cbrt — so
 * that gcc and the GPU evaluate the same IEEE operations and agree bit for bit.  One "try" worth of draws each. */
/* sin and cos of f * pi/2, f in [0,1): folded to [0, pi/4], Taylor polynomials (errors < 3e-8 before rounding) */
static void sincos_quarter(float f, float *s_out, float *c_out) {
    const int swap = f > 0.5f;
    const float g = swap ? 1.0f - f : f;            /* exact */
    const float th = g * 1.57079632679489661923f;
    const float t2 = th * th;
    float sp = 2.75573192239858906526e-6f;          /* 1/9! */
    sp = sp * t2 + -1.98412698412698412698e-4f;     /* -1/7! */
    sp = sp * t2 + 8.33333333333333333333e-3f;      /* 1/5! */
    sp = sp * t2 + -1.66666666666666666667e-1f;     /* -1/3! */
    sp = sp * t2 + 1.0f;
    const float sn = th * sp;
    float cp = 2.48015873015873015873e-5f;          /* 1/8! */
    cp = cp * t2 + -1.38888888888888888889e-3f;     /* -1/6! */
    cp = cp * t2 + 4.16666666666666666667e-2f;      /* 1/4! */
    cp = cp * t2 + -0.5f;
    cp = cp * t2 + 1.0f;
    *s_out = swap ? cp : sn;
    *c_out = swap ? sn : cp;
}
/* x^(1/3), x in [0,1): x * r^2 with r = x^(-1/3) from the exponent trick and three Newton steps r <- r (4 - x r^3) / 3
 * (relative error < 5e-7) */
static float cbrt_unit(float x) {
    uint32_t i;
    memcpy(&i, &x, 4);
    i = 0x54a21d2au - i / 3u;
    float r;
    memcpy(&r, &i, 4);
    const float third = 1.0f / 3.0f;
    for (int k = 0; k < 3; k++) {
        const float r3 = (r * r) * r;
        r = (r * (4.0f - x * r3)) * third;
    }
    return (x == 0.0f) ? 0.0f : x * (r * r);
}
/* uniform in the unit disk (camera.zig:113-120's distribution): sqrt(xi0) * (cos, sin)(2 pi xi1) */
static vec3 direct_vector_in_unit_disk(float xi0, float xi1) {
    const float q = xi1 * 4.0f;                      /* exact */
    const uint32_t k = (uint32_t)q;                  /* quadrant 0..3 */
    float s, c;
    sincos_quarter(q - (float)k, &s, &c);
    const float cx = (k == 0u) ? c : (k == 1u) ? -s : (k == 2u) ? -c : s;
    const float sy = (k == 0u) ? s : (k == 1u) ? c : (k == 2u) ? -s : -c;
    const float rr = sqrtf(xi0);
    return v3(rr * cx, rr * sy, 0.0f);
}
/* uniform in the positive octant of the unit ball (material.zig:27-34's distribution, [0,1)^3 kept inside the ball):
 * direction with z = xi0 uniform (equal areas), azimuth xi1 * pi/2, radius xi2^(1/3) */
static vec3 direct_vector_in_unit_sphere(float xi0, float xi1, float xi2) {
    float s, c;
    sincos_quarter(xi1, &s, &c);
    const float sxy = sqrtf(1.0f - xi0 * xi0);
    const float rho = cbrt_unit(xi2);
    return v3(rho * (sxy * c), rho * (sxy * s), rho * xi0);
}

/* ------------------------------------------------------------------------ */
/* sphere.zig / scene.zig / material.zig / camera.zig / illumination.zig      */
/* ------------------------------------------------------------------------ */
typedef struct { vec3 p, n; float t; int front_face; const orc_sphere *sphere; int index; } hit_t; /* sphere.zig:21-27 */

/* sphere.zig:63-103 */
static int sphere_is_hit(const orc_sphere *self, ray_t ray, float t_min, float t_max, hit_t *hit) {
    const vec3 center = v3(self->cx, self->cy, self->cz);
    const vec3 oc = vsub(ray.origin, center);
    const float a = vlen2(ray.direction);
    const float half_b = vdot(oc, ray.direction);
    const float c = vlen2(oc) - self->radius * self->radius;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return 0;
    const float sqrtd = sqrtf(discriminant);
    const float x1 = (-half_b - sqrtd) / a;
    const float t = x1; /* the far root is ignored: sphere.zig:82-85 */
    if (t < t_max && t > t_min) {
        const vec3 p = point_at(ray, t);
        const vec3 outward_normal = vunit(vmul(vsub(p, center), 1.0f / self->radius));
        hit->p = p;
        hit->n = outward_normal;
        hit->t = t;
        hit->front_face = vdot(ray.direction, outward_normal) < 0; /* sphere.zig:29-34 */
        hit->sphere = self;
        return 1;
    }
    return 0;
}

/* scene.zig:96-110 */
static int scene_is_hit(const orc_sphere *sp, uint32_t n, ray_t ray, float t_min, float t_max, hit_t *out) {
    int found = 0;
    float closest_so_far = t_max;
    for (uint32_t i = 0; i < n; i++) {
        hit_t h;
        if (sphere_is_hit(&sp[i], ray, t_min, t_max, &h)) {
            if (h.t < closest_so_far) {
                h.index = (int)i;
                *out = h;
                found = 1;
                closest_so_far = h.t;
            }
        }
    }
    return found;
}

/* material.zig:27-34 — note [0,1)^3: positive octant of the unit ball. */
static vec3 random_vector_in_unit_sphere(rng_t *r) {
    vec3 p = v3(0, 0, 0);
    if (r->direct) {
        rng_reserve(r, 3);
        const float x = rng_float(r), y = rng_float(r), z = rng_float(r);
        return direct_vector_in_unit_sphere(x, y, z);
    }
    for (int tries = 0; tries < ORC_MAX_TRIES; tries++) {
        rng_reserve(r, 3);
        const float x = rng_float(r), y = rng_float(r), z = rng_float(r);
        p = v3(x, y, z);
        if (vlen2(p) < 1.0f) return p;
    }
    return p;
}
/* camera.zig:113-120 */
static vec3 random_vector_in_unit_disk(rng_t *r) {
    vec3 p = v3(0, 0, 0);
    if (r->direct) {
        rng_reserve(r, 2);
        const float x = rng_float(r), y = rng_float(r);
        return direct_vector_in_unit_disk(x, y);
    }
    for (int tries = 0; tries < ORC_MAX_TRIES; tries++) {
        rng_reserve(r, 2);
        const float x = 2.0f * rng_float(r) - 1.0f;
        const float y = 2.0f * rng_float(r) - 1.0f;
        p = v3(x, y, 0.0f);
        if (vlen2(p) < 1.0f) return p;
    }
    return p;
}

/* material.zig:59-65 */
static vec3 reflect(vec3 d, vec3 n) { return vsub(d, vmul(n, 2.0f * vdot(d, n))); }

/* material.zig:98-116; returns 0 for null (total internal reflection) */
static int refract(vec3 d, vec3 n, float n1_over_n2, vec3 *out) {
    const float dn = vdot(vmul(d, -1.0f), n);
    const float cos_theta = (dn < 1.0f) ? dn : 1.0f; /* math.min */
    const vec3 r_perpendicular = vmul(vadd(d, vmul(n, cos_theta)), n1_over_n2);
    const float discriminant = 1.0f - vlen2(r_perpendicular);
    if (discriminant < 0) return 0;
    const vec3 r_parallel = vmul(n, -1.0f * sqrtf(discriminant));
    *out = vadd(r_perpendicular, r_parallel);
    return 1;
}

/* material.zig:120-124 */
static float schlick(float cosine, float refraction_index) {
    float r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * orc_pow(1.0f - cosine, 5.0f);
}

typedef struct { vec3 attenuation; ray_t ray; } scatter_t; /* material.zig:19-22 */

/* material.zig:43-51 */
static scatter_t lambertian_scatter(const orc_sphere *m, const hit_t *hit, rng_t *r) {
    const vec3 target = vadd(hit->p, vadd(hit->n, random_vector_in_unit_sphere(r)));
    const vec3 d = vunit(vsub(target, hit->p));
    scatter_t s = {v3(m->ar, m->ag, m->ab), {hit->p, d}};
    return s;
}
/* material.zig:79-88 */
static scatter_t metal_scatter(const orc_sphere *m, ray_t ray, const hit_t *hit, rng_t *r) {
    const vec3 sc = reflect(ray.direction, hit->n);
    const vec3 fc = vmul(random_vector_in_unit_sphere(r), m->param);
    const vec3 d = vunit(vadd(sc, fc));
    scatter_t s = {v3(m->ar, m->ag, m->ab), {hit->p, d}};
    return s;
}
/* material.zig:129-167 */
static scatter_t dielectric_scatter(const orc_sphere *m, ray_t ray, const hit_t *hit, rng_t *r) {
    vec3 outward_normal;
    float n1_over_n2, cosine;
    const float ri = m->param;
    const vec3 attenuation = v3(1.0f, 1.0f, 1.0f);
    if (vdot(ray.direction, hit->n) > 0.0f) {
        outward_normal = vunit(v3(-hit->n.x, -hit->n.y, -hit->n.z));
        n1_over_n2 = ri;
        cosine = ri * vdot(ray.direction, hit->n) / vlen(ray.direction);
    } else {
        outward_normal = hit->n;
        n1_over_n2 = 1.0f / ri;
        cosine = -vdot(ray.direction, hit->n) / vlen(ray.direction);
    }
    vec3 refracted_dir;
    scatter_t s;
    s.attenuation = attenuation;
    s.ray.origin = hit->p;
    if (refract(ray.direction, outward_normal, n1_over_n2, &refracted_dir)) {
        const float reflection_prob = schlick(cosine, ri);
        if (rng_float(r) < reflection_prob) s.ray.direction = reflect(ray.direction, hit->n);
        else s.ray.direction = refracted_dir;
    } else {
        s.ray.direction = reflect(ray.direction, hit->n);
    }
    return s;
}
/* material.zig:191-198 (the reference draws from its file-global prng here) */
static scatter_t material_scatter(ray_t ray, const hit_t *hit, rng_t *r) {
    const orc_sphere *m = hit->sphere;
    switch (m->kind) {
    case 0: return lambertian_scatter(m, hit, r);
    case 1: return metal_scatter(m, ray, hit, r);
    default: return dielectric_scatter(m, ray, hit, r);
    }
}

/* camera.zig:99-106 */
static ray_t cast_ray(const orc_camera *c, float s, float t, rng_t *r) {
    const vec3 lookfrom = v3(c->lookfrom[0], c->lookfrom[1], c->lookfrom[2]);
    const vec3 cu = v3(c->u[0], c->u[1], c->u[2]);
    const vec3 cv = v3(c->v[0], c->v[1], c->v[2]);
    const vec3 llc = v3(c->lower_left_corner[0], c->lower_left_corner[1], c->lower_left_corner[2]);
    const vec3 horizontal = v3(c->horizontal[0], c->horizontal[1], c->horizontal[2]);
    const vec3 vertical = v3(c->vertical[0], c->vertical[1], c->vertical[2]);
    const vec3 rd = vmul(random_vector_in_unit_disk(r), c->lens_radius);
    const vec3 offset = vadd(vmul(cu, rd.x), vmul(cv, rd.y));
    const vec3 origin = vadd(lookfrom, offset);
    const vec3 direction =
        vunit(vsub(vsub(vadd(vadd(llc, vmul(horizontal, s)), vmul(vertical, t)), lookfrom), offset));
    ray_t ray = {origin, direction};
    return ray;
}

#define ORC_EPSILON 0.00001f /* constants.zig:3 */

/* illumination.zig:27-34 */
static int lerp(vec3 u, vec3 p0, vec3 p1, vec3 *out) {
    if (fabsf(vlen(u) - 1.0f) > ORC_EPSILON) return ORC_ERR_DIRECTION_NOT_UNIT;
    const float t = (u.y + 1.0f) * 0.5f;
    *out = vadd(vmul(p0, 1.0f - t), vmul(p1, t));
    return ORC_OK;
}

typedef struct {
    const orc_sphere *sp; uint32_t n;
    float t_min, t_max;
    vec3 blend_start, blend_stop;
    uint64_t rays; uint64_t not_unit;
} trace_ctx;

/* illumination.zig:76-90, recursive as the reference */
static int radiance(trace_ctx *tc, rng_t *r, ray_t ray, uint32_t num_rebounds, vec3 *out) {
    hit_t hit;
    tc->rays++;
    if (scene_is_hit(tc->sp, tc->n, ray, tc->t_min, tc->t_max, &hit)) {
        if (num_rebounds <= 0) { *out = v3(0.0f, 0.0f, 0.0f); return ORC_OK; }
        const scatter_t s = material_scatter(ray, &hit, r);
        vec3 v;
        const int rc = radiance(tc, r, s.ray, num_rebounds - 1, &v);
        if (rc != ORC_OK) return rc;
        *out = vemul(v, s.attenuation);
        return ORC_OK;
    }
    const int rc = lerp(ray.direction, tc->blend_start, tc->blend_stop, out);
    if (rc != ORC_OK) tc->not_unit++;
    return rc;
}

/* illumination.zig:42-51.  Sic: the "normal" is taken about the constant centre (0,0,-1). */
static int color_normal(trace_ctx *tc, ray_t ray, vec3 *out) {
    hit_t hit;
    tc->rays++;
    if (scene_is_hit(tc->sp, tc->n, ray, tc->t_min, tc->t_max, &hit)) {
        const vec3 p = vadd(ray.origin, vmul(ray.direction, hit.t)); /* ray.zig:18 pointAt */
        const vec3 n = vunit(vsub(p, v3(0.0f, 0.0f, -1.0f)));
        *out = vmul(vadd(n, v3(1.0f, 1.0f, 1.0f)), 0.5f);
        return ORC_OK;
    }
    const int rc = lerp(ray.direction, tc->blend_start, tc->blend_stop, out);
    if (rc != ORC_OK) tc->not_unit++;
    return rc;
}

/* illumination.zig:60-73 */
static int color_albedo(trace_ctx *tc, ray_t ray, vec3 *out) {
    hit_t hit;
    tc->rays++;
    if (scene_is_hit(tc->sp, tc->n, ray, tc->t_min, tc->t_max, &hit)) {
        const orc_sphere *m = &tc->sp[hit.index];
        *out = (m->kind == 2) ? v3(1.0f, 1.0f, 1.0f) : v3(m->ar, m->ag, m->ab);
        return ORC_OK;
    }
    const int rc = lerp(ray.direction, tc->blend_start, tc->blend_stop, out);
    if (rc != ORC_OK) tc->not_unit++;
    return rc;
}

/* The GPU kernel's shape of the same function: iterative bounce loop that records
 * the hit sphere per bounce, then applies the attenuations innermost-first.  Must
 * equal radiance() bit for bit (tests/test_oracle_kats.py::test_iterative_equals_recursive). */
static int radiance_iterative(trace_ctx *tc, rng_t *r, ray_t ray, uint32_t num_rebounds, vec3 *out) {
    int *stack = (int *)malloc(sizeof(int) * (num_rebounds + 1));
    uint32_t depth = 0;
    vec3 v;
    int rc = ORC_OK;
    for (;;) {
        hit_t hit;
        tc->rays++;
        if (scene_is_hit(tc->sp, tc->n, ray, tc->t_min, tc->t_max, &hit)) {
            if (depth == num_rebounds) { v = v3(0.0f, 0.0f, 0.0f); break; }
            const scatter_t s = material_scatter(ray, &hit, r);
            stack[depth++] = hit.index;
            ray = s.ray;
        } else {
            rc = lerp(ray.direction, tc->blend_start, tc->blend_stop, &v);
            if (rc != ORC_OK) tc->not_unit++;
            break;
        }
    }
    if (rc == ORC_OK) {
        while (depth > 0) {
            const orc_sphere *m = &tc->sp[stack[--depth]];
            const vec3 att = (m->kind == 2) ? v3(1.0f, 1.0f, 1.0f) : v3(m->ar, m->ag, m->ab);
            v = vemul(v, att);
        }
        *out = v;
    }
    free(stack);
    return rc;
}

static void trace_ctx_init(trace_ctx *tc, const orc_sphere *sp, uint32_t n, const orc_config *cfg) {
    tc->sp = sp; tc->n = n; tc->t_min = cfg->t_min; tc->t_max = cfg->t_max;
    tc->blend_start = v3(cfg->blend_start[0], cfg->blend_start[1], cfg->blend_start[2]);
    tc->blend_stop = v3(cfg->blend_stop[0], cfg->blend_stop[1], cfg->blend_stop[2]);
    tc->rays = 0; tc->not_unit = 0;
}

/* One camera sample of pixel idx: renderers/ppm_image.zig:50-51,61-66 */
static int trace_sample(trace_ctx *tc, const orc_camera *cam, const orc_config *cfg, uint32_t idx,
                        rng_t *r_cam, rng_t *r_mat, int iterative, vec3 *out) {
    const uint32_t i_col = idx % cfg->width;
    const uint32_t i_row = idx / cfg->width;
    rng_reserve(r_cam, 2);
    const float u = ((float)i_col + rng_float(r_cam)) / (float)cfg->width;
    const float v = ((float)(cfg->height - i_row + 1) + rng_float(r_cam)) / (float)cfg->height; /* sic: +1 */
    const ray_t ray = cast_ray(cam, u, v, r_cam);
    if (cfg->illumination == 1) return color_normal(tc, ray, out); /* renderers/ppm_image.zig:64 */
    if (cfg->illumination == 2) return color_albedo(tc, ray, out); /* renderers/ppm_image.zig:65 */
    return iterative ? radiance_iterative(tc, r_mat, ray, cfg->rebounds, out)
                     : radiance(tc, r_mat, ray, cfg->rebounds, out);
}

static void sample_range(const orc_config *cfg, uint32_t *s0, uint32_t *s1, uint32_t *chunk) {
    const uint32_t S = cfg->subpixels * cfg->rays_per_subsample;
    uint32_t a = cfg->sample_start, b = cfg->sample_stop;
    if (a == 0 && b == 0) { b = S; }
    *s0 = a; *s1 = b;
    uint32_t c = cfg->chunk;
    if (c == 0 || c > b - a) c = (b > a) ? b - a : 1;
    *chunk = c;
}

/* The sample loops of render() for pixels [i0,i1): renderers/ppm_image.zig:49-69.
 * Stochastic/deterministic modes: each (pixel, sample) owns a Philox stream /
 * the constant stream, consumed in the reference's call order. */
static int render_range(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg,
                        uint32_t i0, uint32_t i1, float *rgb_sum, orc_stats *st) {
    trace_ctx tc;
    trace_ctx_init(&tc, sp, n, cfg);
    uint32_t s0, s1, chunk;
    sample_range(cfg, &s0, &s1, &chunk);
    uint64_t paths = 0;
    int rc_all = ORC_OK;
    for (uint32_t idx = i0; idx < i1; idx++) {
        if (cfg->tile_pixels) { /* multi-GPU tile split: only this owner's runs */
            const uint32_t t = (idx - cfg->idx_start) / cfg->tile_pixels;
            if (t % cfg->tile_count != cfg->tile_index) continue;
        }
        vec3 total = v3(0.0f, 0.0f, 0.0f);
        int first = 1;
        for (uint32_t c0 = s0; c0 < s1; c0 += chunk) {
            const uint32_t c1 = (c0 + chunk < s1) ? c0 + chunk : s1;
            vec3 color_accum = v3(0.0f, 0.0f, 0.0f);
            for (uint32_t s = c0; s < c1; s++) {
                rng_t r;
                if (cfg->mode == ORC_MODE_DETERMINISTIC) { memset(&r, 0, sizeof r); r.kind = RNG_CONST; }
                else { rng_philox_path(&r, cfg->seed, idx, s); r.direct = cfg->sampler == ORC_SAMPLER_DIRECT; }
                vec3 color_sample;
                const int rc = trace_sample(&tc, cam, cfg, idx, &r, &r, 0, &color_sample);
                paths++;
                if (rc != ORC_OK) { rc_all = rc; continue; }
                color_accum = vadd(color_accum, color_sample);
            }
            if (first) { total = color_accum; first = 0; }
            else total = vadd(total, color_accum);
        }
        rgb_sum[3 * (size_t)idx + 0] = total.x;
        rgb_sum[3 * (size_t)idx + 1] = total.y;
        rgb_sum[3 * (size_t)idx + 2] = total.z;
    }
    if (st) { st->paths += paths; st->rays += tc.rays; st->sphere_tests += tc.rays * n; st->not_unit += tc.not_unit; }
    return rc_all;
}

/* multithreading.zig:40-44 */
uint64_t orc_chunk_size(uint64_t num_pixels, uint64_t num_threads) {
    const uint64_t n = num_pixels / num_threads;
    const uint64_t rem = num_pixels % num_threads;
    return rem > 0 ? n + 1 : n;
}

typedef struct {
    const orc_sphere *sp; uint32_t n; const orc_camera *cam; const orc_config *cfg;
    uint32_t i0, i1; float *rgb_sum; orc_stats st; int rc;
} thread_arg;

static void *thread_main(void *p) {
    thread_arg *a = (thread_arg *)p;
    memset(&a->st, 0, sizeof a->st);
    a->rc = render_range(a->sp, a->n, a->cam, a->cfg, a->i0, a->i1, a->rgb_sum, &a->st);
    return NULL;
}

int orc_render_sum(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg,
                   float *rgb_sum, uint32_t nthreads, orc_stats *stats) {
    if (!sp || !cam || !cfg || !rgb_sum) return ORC_ERR_BAD_ARG;
    if (cfg->idx_stop > cfg->width * cfg->height || cfg->idx_start > cfg->idx_stop) return ORC_ERR_BAD_ARG;
    if (cfg->mode == ORC_MODE_ZIG) return ORC_ERR_BAD_ARG; /* use orc_render_sum_zig */
    if (stats) memset(stats, 0, sizeof *stats);
    if (nthreads <= 1) return render_range(sp, n, cam, cfg, cfg->idx_start, cfg->idx_stop, rgb_sum, stats);
    /* renderers/ppm_image.zig:100-108 over the requested pixel range */
    const uint64_t num_pixels = cfg->idx_stop - cfg->idx_start;
    const uint64_t ppt = orc_chunk_size(num_pixels, nthreads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    thread_arg *args = (thread_arg *)malloc(sizeof(thread_arg) * nthreads);
    int rc_all = ORC_OK;
    for (uint32_t i = 0; i < nthreads; i++) {
        uint64_t istart = (uint64_t)i * ppt;
        uint64_t istop = (istart + ppt <= num_pixels) ? istart + ppt : num_pixels;
        if (istart > num_pixels) istart = num_pixels;
        thread_arg a = {sp, n, cam, cfg, (uint32_t)(cfg->idx_start + istart), (uint32_t)(cfg->idx_start + istop), rgb_sum, {0, 0, 0, 0}, 0};
        args[i] = a;
        pthread_create(&th[i], NULL, thread_main, &args[i]);
    }
    for (uint32_t i = 0; i < nthreads; i++) {
        pthread_join(th[i], NULL);
        if (args[i].rc != ORC_OK) rc_all = args[i].rc;
        if (stats) {
            stats->paths += args[i].st.paths; stats->rays += args[i].st.rays;
            stats->sphere_tests += args[i].st.sphere_tests; stats->not_unit += args[i].st.not_unit;
        }
    }
    free(th); free(args);
    return rc_all;
}

int orc_sample_radiance(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg,
                        uint32_t idx, uint32_t sample, int iterative, float rgb[3]) {
    trace_ctx tc;
    trace_ctx_init(&tc, sp, n, cfg);
    rng_t r;
    if (cfg->mode == ORC_MODE_DETERMINISTIC) { memset(&r, 0, sizeof r); r.kind = RNG_CONST; }
    else { rng_philox_path(&r, cfg->seed, idx, sample); r.direct = cfg->sampler == ORC_SAMPLER_DIRECT; }
    vec3 c = v3(0, 0, 0);
    const int rc = trace_sample(&tc, cam, cfg, idx, &r, &r, iterative, &c);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return rc;
}

/* main.zig:166-177 + material.zig:15: single thread, Zig RNG streams. */
int orc_render_sum_zig(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg,
                       orc_zigrng *zr, float *rgb_sum, orc_stats *stats) {
    if (!sp || !cam || !cfg || !rgb_sum || !zr) return ORC_ERR_BAD_ARG;
    trace_ctx tc;
    trace_ctx_init(&tc, sp, n, cfg);
    orc_zigrng mat_state;
    orc_zig_seed_kind(&mat_state, 0, zr->kind); /* material.zig:15 */
    rng_t r_cam, r_mat;
    memset(&r_cam, 0, sizeof r_cam); r_cam.kind = RNG_ZIG; r_cam.z = zr;
    memset(&r_mat, 0, sizeof r_mat); r_mat.kind = RNG_ZIG; r_mat.z = &mat_state;
    r_cam.direct = r_mat.direct = cfg->sampler == ORC_SAMPLER_DIRECT;
    uint64_t paths = 0;
    int rc_all = ORC_OK;
    for (uint32_t idx = cfg->idx_start; idx < cfg->idx_stop; idx++) {
        vec3 color_accum = v3(0.0f, 0.0f, 0.0f);
        for (uint32_t sample = 0; sample < cfg->subpixels; sample++)
            for (uint32_t rps = 0; rps < cfg->rays_per_subsample; rps++) {
                vec3 cs;
                const int rc = trace_sample(&tc, cam, cfg, idx, &r_cam, &r_mat, 0, &cs);
                paths++;
                if (rc != ORC_OK) { rc_all = rc; continue; }
                color_accum = vadd(color_accum, cs);
            }
        rgb_sum[3 * (size_t)idx + 0] = color_accum.x;
        rgb_sum[3 * (size_t)idx + 1] = color_accum.y;
        rgb_sum[3 * (size_t)idx + 2] = color_accum.z;
    }
    if (stats) { memset(stats, 0, sizeof *stats); stats->paths = paths; stats->rays = tc.rays; stats->sphere_tests = tc.rays * n; stats->not_unit = tc.not_unit; }
    return rc_all;
}

/* renderers/ppm_image.zig:70 */
void orc_resolve_mean(const orc_config *cfg, const float *rgb_sum, float *rgb_mean) {
    const float k1 = 1.0f / (float)cfg->subpixels;
    const float k2 = 1.0f / (float)cfg->rays_per_subsample;
    for (uint32_t idx = cfg->idx_start; idx < cfg->idx_stop; idx++) {
        vec3 c = v3(rgb_sum[3 * (size_t)idx], rgb_sum[3 * (size_t)idx + 1], rgb_sum[3 * (size_t)idx + 2]);
        c = vmul(vmul(c, k1), k2);
        rgb_mean[3 * (size_t)idx] = c.x; rgb_mean[3 * (size_t)idx + 1] = c.y; rgb_mean[3 * (size_t)idx + 2] = c.z;
    }
}

/* ------------------------------------------------------------------------ */
/* utils.zig, render_targets/ppm_image.zig                                    */
/* ------------------------------------------------------------------------ */
uint32_t orc_num_digits(uint64_t num) { /* utils.zig:12-20 */
    uint64_t x = num; uint32_t count = 0;
    while (x != 0) { x /= 10; count += 1; }
    return count;
}
float orc_degrees_to_radians(float deg) { return deg * 3.14159265358979323846f / 180.0f; } /* utils.zig:113-115 */

int orc_col_num_to_string(uint64_t max_value, uint32_t value, char *out, size_t cap) { /* utils.zig:62-84 */
    if (value > max_value) return ORC_ERR_VALUE_EXCEEDS_MAX;
    if (cap < 4) return ORC_ERR_NO_SPACE;
    if (value <= 9) snprintf(out, cap, "  %u", value);
    else if (value <= 99) snprintf(out, cap, " %u", value);
    else snprintf(out, cap, "%u", value);
    return ORC_OK;
}

void orc_image_new(uint32_t width, float aspect_ratio, uint32_t max_px_value, orc_image *img) { /* render_targets/ppm_image.zig:59-81 */
    const uint32_t height = (uint32_t)((float)width / aspect_ratio);
    img->aspect_ratio = aspect_ratio;
    img->width = width; img->height = height; img->max_px_value = max_px_value;
    img->px_size = 3 * orc_num_digits(max_px_value) + 2 + 1;
    img->header_size = 2 + 1 + orc_num_digits(width) + 1 + orc_num_digits(height) + 1 + orc_num_digits(max_px_value) + 1;
    img->num_pixels = (uint64_t)width * height;
    img->data_size = (uint64_t)img->px_size * width * height;
    img->size = img->header_size + img->data_size;
    img->blend_start[0] = 1.0f; img->blend_start[1] = 1.0f; img->blend_start[2] = 1.0f;
    img->blend_stop[0] = 0.5f; img->blend_stop[1] = 0.7f; img->blend_stop[2] = 1.0f;
}
int orc_image_header(const orc_image *img, char *buf, size_t cap) { /* render_targets/ppm_image.zig:84-88 */
    const int n = snprintf(buf, cap, "P3\n%u %u\n%u\n", img->width, img->height, img->max_px_value);
    if (n < 0 || (size_t)n >= cap) return ORC_ERR_NO_SPACE;
    return n;
}
/* @floatToInt(u8, x): truncation toward zero.  Out-of-range input is undefined in
 * the reference's ReleaseFast; the build clamps to [0,255]. */
static uint32_t float_to_u8(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 255.0f) return 255;
    return (uint32_t)x;
}
int orc_rgb_to_ascii(const orc_image *img, const float col[3], char *out, size_t cap) { /* render_targets/ppm_image.zig:93-103 */
    if (img->max_px_value != 255) return ORC_ERR_UNSUPPORTED;
    if (cap < img->px_size + 1) return ORC_ERR_NO_SPACE;
    char r[8], g[8], b[8];
    int rc;
    if ((rc = orc_col_num_to_string(img->max_px_value, float_to_u8(col[0]), r, sizeof r)) != ORC_OK) return rc;
    if ((rc = orc_col_num_to_string(img->max_px_value, float_to_u8(col[1]), g, sizeof g)) != ORC_OK) return rc;
    if ((rc = orc_col_num_to_string(img->max_px_value, float_to_u8(col[2]), b, sizeof b)) != ORC_OK) return rc;
    snprintf(out, cap, "%s %s %s\n", r, g, b);
    return ORC_OK;
}

/* renderers/ppm_image.zig:70-82 */
int orc_resolve_ppm(const orc_config *cfg, const orc_image *img, const float *rgb_sum, uint8_t *slice, size_t slice_len) {
    if (img->max_px_value != 255) return ORC_ERR_UNSUPPORTED;
    const float k1 = 1.0f / (float)cfg->subpixels;
    const float k2 = 1.0f / (float)cfg->rays_per_subsample;
    const float max_val_f = (float)img->max_px_value;
    int exceeded = 0;
    for (uint32_t idx = cfg->idx_start; idx < cfg->idx_stop; idx++) {
        vec3 c = v3(rgb_sum[3 * (size_t)idx], rgb_sum[3 * (size_t)idx + 1], rgb_sum[3 * (size_t)idx + 2]);
        c = vmul(vmul(c, k1), k2);
        const vec3 col = vmul(c, max_val_f);
        const float colf[3] = {col.x, col.y, col.z};
        char ascii[32];
        /* renderers/ppm_image.zig:75-77 asserts col <= 255: reported after every record is written (clamped) */
        if (col.x > max_val_f || col.y > max_val_f || col.z > max_val_f) exceeded = 1;
        const int rc = orc_rgb_to_ascii(img, colf, ascii, sizeof ascii);
        if (rc != ORC_OK) return rc;
        const size_t offset = (size_t)idx * img->px_size;
        if (img->header_size + offset + img->px_size > slice_len) return ORC_ERR_NO_SPACE;
        memcpy(slice + img->header_size + offset, ascii, img->px_size);
    }
    return exceeded ? ORC_ERR_VALUE_EXCEEDS_MAX : ORC_OK;
}

int orc_render_ppm(const orc_sphere *sp, uint32_t n, const orc_camera *cam, const orc_config *cfg, const orc_image *img,
                   uint8_t *slice, size_t slice_len, uint32_t nthreads, orc_stats *stats) {
    float *sum = (float *)calloc((size_t)cfg->width * cfg->height * 3, sizeof(float));
    if (!sum) return ORC_ERR_NO_SPACE;
    int rc = orc_render_sum(sp, n, cam, cfg, sum, nthreads, stats);
    if (rc == ORC_OK) rc = orc_resolve_ppm(cfg, img, sum, slice, slice_len);
    free(sum);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* scene.zig:32-94, main.zig:107-136, camera.zig:48-97                        */
/* ------------------------------------------------------------------------ */
static orc_sphere mk_sphere(float cx, float cy, float cz, float radius, uint32_t kind, float ar, float ag, float ab, float param) {
    orc_sphere s = {cx, cy, cz, radius, kind, ar, ag, ab, param};
    return s;
}
int orc_scene_18_19_20(orc_sphere *out, uint32_t cap) { /* scene.zig:40-49 */
    if (cap < 5) return ORC_ERR_NO_SPACE;
    out[0] = mk_sphere(0, -100.5f, -1, 100.5f, 0, 0.8f, 0.8f, 0.0f, 0.0f);  /* ground */
    out[1] = mk_sphere(0, 0, -1, 0.5f, 0, 0.1f, 0.2f, 0.5f, 0.0f);          /* center */
    out[2] = mk_sphere(-1, 0, -1, 0.5f, 2, 1.0f, 1.0f, 1.0f, 1.5f);         /* left, glass */
    out[3] = mk_sphere(-1, 0, -1, -0.45f, 2, 1.0f, 1.0f, 1.0f, 1.5f);       /* left, negative radius */
    out[4] = mk_sphere(1, 0, -1, 0.5f, 1, 0.8f, 0.6f, 0.2f, 0.0f);          /* right, metal fuzz 0 */
    return 5;
}
int orc_scene_21(orc_zigrng *r, orc_sphere *out, uint32_t cap) { /* scene.zig:52-84 */
    if (cap < 488) return ORC_ERR_NO_SPACE;
    uint32_t k = 0;
    out[k++] = mk_sphere(0, -1000, 0, 1000, 0, 0.5f, 0.5f, 0.5f, 0.0f);
    for (float a = -11; a < 11; a += 1) {
        for (float b = -11; b < 11; b += 1) {
            const float choose_mat = orc_zig_float(r);
            const float cx = a + 0.9f * orc_zig_float(r);
            const float cz = b + 0.9f * orc_zig_float(r);
            if (choose_mat < 0.8f) {
                const float ar = orc_zig_float(r), ag = orc_zig_float(r), ab = orc_zig_float(r);
                out[k++] = mk_sphere(cx, 0.2f, cz, 0.2f, 0, ar, ag, ab, 0.0f);
            } else if (choose_mat < 0.95f) {
                const float ar = orc_zig_float(r), ag = orc_zig_float(r), ab = orc_zig_float(r);
                const float fuzziness = 0.5f * orc_zig_float(r);
                out[k++] = mk_sphere(cx, 0.2f, cz, 0.2f, 1, ar, ag, ab, fuzziness);
            } else {
                out[k++] = mk_sphere(cx, 0.2f, cz, 0.2f, 2, 1.0f, 1.0f, 1.0f, 1.5f);
            }
        }
    }
    out[k++] = mk_sphere(0, 1, 0, 1.0f, 2, 1.0f, 1.0f, 1.0f, 1.5f);
    out[k++] = mk_sphere(-4, 1, 0, 1.0f, 0, 0.4f, 0.2f, 0.1f, 0.0f);
    out[k++] = mk_sphere(4, 1, 0, 1.0f, 1, 0.7f, 0.6f, 0.5f, 0.0f);
    return (int)k;
}
int orc_setup_scene(int n_scene, uint64_t seed, orc_sphere *out, uint32_t cap, orc_zigrng *rng_after) {
    return orc_setup_scene_prng(n_scene, seed, ORC_PRNG_DEFAULT, out, cap, rng_after);
}
int orc_setup_scene_prng(int n_scene, uint64_t seed, uint32_t prng, orc_sphere *out, uint32_t cap, orc_zigrng *rng_after) { /* scene.zig:86-94, main.zig:100-105 */
    orc_zigrng r;
    if (prng > ORC_PRNG_XOSHIRO256) return ORC_ERR_BAD_ARG;
    orc_zig_seed_kind(&r, seed, prng);
    int k;
    if (n_scene == 18 || n_scene == 19 || n_scene == 20) k = orc_scene_18_19_20(out, cap);
    else if (n_scene == 21) k = orc_scene_21(&r, out, cap);
    else return ORC_ERR_SCENE_NOT_AVAILABLE;
    if (rng_after) *rng_after = r;
    return k;
}
/* BASELINE config 5 (not in the reference): n spheres = ground (r=1000) + n-1 small
 * ones, centres uniform in [-50,50] x {r} x [-50,50], r in U[0.05,0.3], materials
 * 80/15/5 % as scene.zig:63-72, drawn from Xoshiro256++ seeded `seed`. */
int orc_scene_synthetic(uint32_t n, uint64_t seed, orc_sphere *out, uint32_t cap) {
    if (cap < n || n < 1) return ORC_ERR_NO_SPACE;
    orc_zigrng r;
    orc_zig_seed_kind(&r, seed, ORC_PRNG_XOSHIRO256); /* this scene's own definition (not a reference scene) */
    uint32_t k = 0;
    out[k++] = mk_sphere(0, -1000, 0, 1000, 0, 0.5f, 0.5f, 0.5f, 0.0f);
    while (k < n) {
        const float choose_mat = orc_zig_float(&r);
        const float radius = 0.05f + 0.25f * orc_zig_float(&r);
        const float cx = -50.0f + 100.0f * orc_zig_float(&r);
        const float cz = -50.0f + 100.0f * orc_zig_float(&r);
        if (choose_mat < 0.8f) {
            const float ar = orc_zig_float(&r), ag = orc_zig_float(&r), ab = orc_zig_float(&r);
            out[k++] = mk_sphere(cx, radius, cz, radius, 0, ar, ag, ab, 0.0f);
        } else if (choose_mat < 0.95f) {
            const float ar = orc_zig_float(&r), ag = orc_zig_float(&r), ab = orc_zig_float(&r);
            const float fuzziness = 0.5f * orc_zig_float(&r);
            out[k++] = mk_sphere(cx, radius, cz, radius, 1, ar, ag, ab, fuzziness);
        } else {
            out[k++] = mk_sphere(cx, radius, cz, radius, 2, 1.0f, 1.0f, 1.0f, 1.5f);
        }
    }
    return (int)k;
}

void orc_camera_new(const float lookfrom_[3], const float lookat_[3], const float vup_[3], float vfov,
                    float aspect_ratio, float aperture, float focus_dist, orc_camera *out) { /* camera.zig:48-97 */
    const vec3 lookfrom = v3(lookfrom_[0], lookfrom_[1], lookfrom_[2]);
    const vec3 lookat = v3(lookat_[0], lookat_[1], lookat_[2]);
    const vec3 vup = v3(vup_[0], vup_[1], vup_[2]);
    const float theta = orc_degrees_to_radians(vfov);
    /* Zig's f32 tan is not available here; tan evaluated in double and rounded
     * (unpinned: may differ from Zig's tan32 in the last bit). */
    const float h = (float)tan((double)(theta / 2));
    const float vh = 2 * h;
    const float vw = vh * aspect_ratio;
    const vec3 w = vunit(vsub(lookfrom, lookat));
    const vec3 u = vunit(vcross(vup, w));
    const vec3 v = vcross(w, u);
    const vec3 horizontal = vmul(vmul(u, vw), focus_dist);
    const vec3 vertical = vmul(vmul(v, vh), focus_dist);
    const vec3 half_vw = vmul(horizontal, 0.5f);
    const vec3 half_vh = vmul(vertical, 0.5f);
    const vec3 llc = vsub(vsub(vsub(lookfrom, half_vw), half_vh), vmul(w, focus_dist));
    const float lens_radius = aperture / 2.0f;
    out->lookfrom[0] = lookfrom.x; out->lookfrom[1] = lookfrom.y; out->lookfrom[2] = lookfrom.z;
    out->u[0] = u.x; out->u[1] = u.y; out->u[2] = u.z;
    out->v[0] = v.x; out->v[1] = v.y; out->v[2] = v.z;
    out->lower_left_corner[0] = llc.x; out->lower_left_corner[1] = llc.y; out->lower_left_corner[2] = llc.z;
    out->horizontal[0] = horizontal.x; out->horizontal[1] = horizontal.y; out->horizontal[2] = horizontal.z;
    out->vertical[0] = vertical.x; out->vertical[1] = vertical.y; out->vertical[2] = vertical.z;
    out->lens_radius = lens_radius;
}

int orc_camera_preset(int n_scene, float aspect_ratio, orc_camera *out) { /* main.zig:107-136 */
    float lookfrom[3] = {0.0f, 0.0f, 0.0f};
    const float lookat[3] = {0.0f, 0.0f, -1.0f};
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float vfov = 20.0f, focal_dist, aperture = 1.0f;
    if (n_scene == 18) { lookfrom[0] = -2.0f; lookfrom[1] = 2.0f; lookfrom[2] = 1.0f; vfov = 90.0f; aperture = 0.2f; }
    else if (n_scene == 19) { lookfrom[0] = -2.0f; lookfrom[1] = 2.0f; lookfrom[2] = 1.0f; aperture = 0.2f; }
    else if (n_scene == 20) { lookfrom[0] = 3.0f; lookfrom[1] = 3.0f; lookfrom[2] = 2.0f; aperture = 2.0f; }
    else if (n_scene == 21) { lookfrom[0] = 13.0f; lookfrom[1] = 2.0f; lookfrom[2] = 3.0f; aperture = 0.1f; }
    else return ORC_ERR_SCENE_NOT_AVAILABLE;
    if (n_scene == 21) focal_dist = 10.0f;
    else focal_dist = vlen(vsub(v3(lookfrom[0], lookfrom[1], lookfrom[2]), v3(lookat[0], lookat[1], lookat[2])));
    orc_camera_new(lookfrom, lookat, up, vfov, aspect_ratio, aperture, focal_dist, out);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* per-function exports                                                       */
/* ------------------------------------------------------------------------ */
static void rng_array(rng_t *r, const float *xi, uint32_t n_xi) { memset(r, 0, sizeof *r); r->kind = RNG_ARRAY; r->xi = xi; r->n_xi = n_xi; }

void orc_cast_ray(const orc_camera *cam, float s, float t, const float *xi, uint32_t n_xi, uint32_t *n_used, float o[3], float d[3]) {
    rng_t r; rng_array(&r, xi, n_xi);
    const ray_t ray = cast_ray(cam, s, t, &r);
    if (n_used) *n_used = r.n;
    o[0] = ray.origin.x; o[1] = ray.origin.y; o[2] = ray.origin.z;
    d[0] = ray.direction.x; d[1] = ray.direction.y; d[2] = ray.direction.z;
}
int orc_sphere_hit(const orc_sphere *sp, const float o[3], const float d[3], float t_min, float t_max, float *t, float p[3], float n[3]) {
    ray_t ray = {v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    hit_t h;
    if (!sphere_is_hit(sp, ray, t_min, t_max, &h)) return 0;
    *t = h.t; p[0] = h.p.x; p[1] = h.p.y; p[2] = h.p.z; n[0] = h.n.x; n[1] = h.n.y; n[2] = h.n.z;
    return 1;
}
int orc_scene_hit(const orc_sphere *sp, uint32_t n, const float o[3], const float d[3], float t_min, float t_max, float *t, float p[3], float nrm[3]) {
    ray_t ray = {v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    hit_t h;
    if (!scene_is_hit(sp, n, ray, t_min, t_max, &h)) return -1;
    *t = h.t; p[0] = h.p.x; p[1] = h.p.y; p[2] = h.p.z; nrm[0] = h.n.x; nrm[1] = h.n.y; nrm[2] = h.n.z;
    return h.index;
}
/* the two samplers on caller-supplied draws: sampler = ORC_SAMPLER_REJECTION (the reference's loops) or ORC_SAMPLER_DIRECT */
void orc_sample_disk(uint32_t sampler, const float *xi, uint32_t n_xi, uint32_t *n_used, float out[3]) {
    rng_t r; rng_array(&r, xi, n_xi); r.direct = sampler == ORC_SAMPLER_DIRECT;
    const vec3 p = random_vector_in_unit_disk(&r);
    if (n_used) *n_used = r.n;
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
void orc_sample_ball(uint32_t sampler, const float *xi, uint32_t n_xi, uint32_t *n_used, float out[3]) {
    rng_t r; rng_array(&r, xi, n_xi); r.direct = sampler == ORC_SAMPLER_DIRECT;
    const vec3 p = random_vector_in_unit_sphere(&r);
    if (n_used) *n_used = r.n;
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}
/* n records of `stride` draws each -> (p, n_used) per record; which: 0 disk, 1 ball */
void orc_sample_batch(int which, uint32_t sampler, const float *xi, uint32_t n, uint32_t stride, float *out4) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t used = 0;
        if (which == 0) orc_sample_disk(sampler, xi + (size_t)i * stride, stride, &used, out4 + 4 * (size_t)i);
        else orc_sample_ball(sampler, xi + (size_t)i * stride, stride, &used, out4 + 4 * (size_t)i);
        out4[4 * (size_t)i + 3] = (float)used;
    }
}
/* n rays (o, d: 6 floats each) -> (index or -1, t, p, n: 8 floats each): scene.zig:96-110 ray by ray */
void orc_scene_hit_batch(const orc_sphere *sp, uint32_t n_sp, const float *rays, uint32_t n, float t_min, float t_max, float *out8) {
    for (uint32_t i = 0; i < n; i++) {
        float t = 0.0f, p[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
        const int idx = orc_scene_hit(sp, n_sp, rays + 6 * (size_t)i, rays + 6 * (size_t)i + 3, t_min, t_max, &t, p, nn);
        float *w = out8 + 8 * (size_t)i;
        w[0] = (float)idx; w[1] = idx >= 0 ? t : 0.0f;
        for (int k = 0; k < 3; k++) { w[2 + k] = idx >= 0 ? p[k] : 0.0f; w[5 + k] = idx >= 0 ? nn[k] : 0.0f; }
    }
}
void orc_reflect(const float d[3], const float n[3], float out[3]) {
    vec3 r = reflect(v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_refract(const float d[3], const float n[3], float n1_over_n2, float out[3]) {
    vec3 r;
    if (!refract(v3(d[0], d[1], d[2]), v3(n[0], n[1], n[2]), n1_over_n2, &r)) return 0;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return 1;
}
float orc_schlick(float cosine, float refraction_index) { return schlick(cosine, refraction_index); }
void orc_scatter(const orc_sphere *sp, const float d[3], const float p[3], const float n[3],
                 const float *xi, uint32_t n_xi, uint32_t *n_used, float atten[3], float so[3], float sd[3]) {
    rng_t r; rng_array(&r, xi, n_xi);
    ray_t ray = {v3(0, 0, 0), v3(d[0], d[1], d[2])};
    hit_t h; memset(&h, 0, sizeof h);
    h.p = v3(p[0], p[1], p[2]); h.n = v3(n[0], n[1], n[2]); h.sphere = sp;
    const scatter_t s = material_scatter(ray, &h, &r);
    if (n_used) *n_used = r.n;
    atten[0] = s.attenuation.x; atten[1] = s.attenuation.y; atten[2] = s.attenuation.z;
    so[0] = s.ray.origin.x; so[1] = s.ray.origin.y; so[2] = s.ray.origin.z;
    sd[0] = s.ray.direction.x; sd[1] = s.ray.direction.y; sd[2] = s.ray.direction.z;
}
int orc_lerp(const float u[3], const float p0[3], const float p1[3], float out[3]) {
    vec3 r = v3(0, 0, 0);
    const int rc = lerp(v3(u[0], u[1], u[2]), v3(p0[0], p0[1], p0[2]), v3(p1[0], p1[1], p1[2]), &r);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return rc;
}
