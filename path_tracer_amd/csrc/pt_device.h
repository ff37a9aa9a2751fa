// pt_device.h — device functions of the path-tracing hot path (gfx950 / wave64).
//
// Every function here follows the reference's f32 arithmetic operation by operation
// (the translation unit is compiled with -ffp-contract=off; divisions and square
// roots are correctly rounded: plain `/` and sqrtf under
// -fhip-fp32-correctly-rounded-divide-sqrt; NOT __fsqrt_rn, which is the native approximation), so that a path traced on the GPU
// produces the same bits as the reference's CPU arithmetic for the same random
// draws.  Reference citations are file:line under src/ of jackdbd/path-tracer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptdev {

struct V3 { float x, y, z; };

__host__ __device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
// vec3.zig:42-58: add / sub / mul(k) (k * component) / elementwiseMul
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vmul(V3 a, float k) { return v3(k * a.x, k * a.y, k * a.z); }
__device__ __forceinline__ V3 vemul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
// vec3.zig:67-69: left-to-right sum of products, no FMA
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vec3.zig:37-39: pow(x,2)+pow(y,2)+pow(z,2); Zig's pow(x,2) is x*x (mantissa square, one rounding)
__device__ __forceinline__ float vlen2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float vlen(V3 a) { return sqrtf(vlen2(a)); } // vec3.zig:30-32
// vec3.zig:61-64: multiply by the reciprocal of the norm
__device__ __forceinline__ V3 vunit(V3 a) {
    const float inv_n = 1.0f / vlen(a);
    return v3(inv_n * a.x, inv_n * a.y, inv_n * a.z);
}

// ---- random draws -----------------------------------------------------------------
// Philox4x32-10 path stream: key = (seed lo, seed hi); counter = (pixel, sample,
// block, 0); draw n of a path is word n&3 of block n>>2; the (u,v) jitter pair, each disk
// try (2 draws) and each unit-sphere try (3 draws) skip to the next block when the current
// one cannot hold them.  Replaces
// std.rand.DefaultPrng (main.zig:100, renderers/ppm_image.zig:110, material.zig:15).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3) {
#ifndef PT_EXPERIMENT_PHILOX_ROUNDS // tools/build_variant.sh: timing experiments only (breaks parity)
#define PT_EXPERIMENT_PHILOX_ROUNDS 10
#endif
#pragma unroll
    for (int r = 0; r < PT_EXPERIMENT_PHILOX_ROUNDS; ++r) {
        // one 32x32->64 product each (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// Random.float(f32) of Zig 0.9.1: bits 0x3F800000 | (s >> 9) in [1,2), minus 1.
__device__ __forceinline__ float u32_to_float(uint32_t s) { return __uint_as_float(0x3f800000u | (s >> 9)) - 1.0f; }

struct PhiloxRng { // stochastic mode
    uint32_t k0, k1, pixel, sample, n;
    uint32_t b0, b1, b2, b3;
    __device__ __forceinline__ void init(uint32_t seed_lo, uint32_t seed_hi) {
        k0 = seed_lo; k1 = seed_hi; pixel = sample = n = 0; b0 = b1 = b2 = b3 = 0;
    }
    __device__ __forceinline__ void start_path(uint32_t px, uint32_t s) { pixel = px; sample = s; n = 0; }
    __device__ __forceinline__ float next() {
        const uint32_t k = n & 3u;
        if (k == 0) philox4x32_10(pixel, sample, n >> 2, 0u, k0, k1, b0, b1, b2, b3);
        // Select among VALUES read unconditionally: with the member reads inside the arms of ?: the optimizer
        // sinks them into one load from a selected ADDRESS, and the whole struct then lives in scratch memory
        // (seen in the ISA as scratch_store/scratch_load around every Philox call).
        const uint32_t x0 = b0, x1 = b1, x2 = b2, x3 = b3;
        const uint32_t w = (k == 0) ? x0 : (k == 1) ? x1 : (k == 2) ? x2 : x3;
        n++;
        return u32_to_float(w);
    }
    // Two / three draws that never straddle a Philox block: if fewer words than needed
    // are left in the current block, the rest of the block is skipped ("reserve").  A
    // rejection-loop try therefore runs Philox at most once, and at ONE program point for
    // all lanes of a wave, instead of possibly at each of its draws.
    __device__ __forceinline__ void next2(float &x, float &y) {
        uint32_t k = n & 3u;
        if (k == 3u) { n += 1u; k = 0u; }
        if (k == 0u) philox4x32_10(pixel, sample, n >> 2, 0u, k0, k1, b0, b1, b2, b3);
        const uint32_t x0 = b0, x1 = b1, x2 = b2, x3 = b3; // values, not addresses: see next()
        const uint32_t w0 = (k == 0u) ? x0 : (k == 1u) ? x1 : x2;
        const uint32_t w1 = (k == 0u) ? x1 : (k == 1u) ? x2 : x3;
        n += 2u;
        x = u32_to_float(w0); y = u32_to_float(w1);
    }
    __device__ __forceinline__ void next3(float &x, float &y, float &z) {
        uint32_t k = n & 3u;
        if (k > 1u) { n = (n + 3u) & ~3u; k = 0u; }
        if (k == 0u) philox4x32_10(pixel, sample, n >> 2, 0u, k0, k1, b0, b1, b2, b3);
        const uint32_t x0 = b0, x1 = b1, x2 = b2, x3 = b3; // values, not addresses: see next()
        const uint32_t w0 = (k == 0u) ? x0 : x1;
        const uint32_t w1 = (k == 0u) ? x1 : x2;
        const uint32_t w2 = (k == 0u) ? x2 : x3;
        n += 3u;
        x = u32_to_float(w0); y = u32_to_float(w1); z = u32_to_float(w2);
    }
};
struct ConstRng { // deterministic mode: every draw is 0.5
    __device__ __forceinline__ void init(uint32_t, uint32_t) {}
    __device__ __forceinline__ void start_path(uint32_t, uint32_t) {}
    __device__ __forceinline__ float next() { return 0.5f; }
    __device__ __forceinline__ void next2(float &x, float &y) { x = y = 0.5f; }
    __device__ __forceinline__ void next3(float &x, float &y, float &z) { x = y = z = 0.5f; }
};
struct ArrayRng { // per-function parity vectors: draws supplied by the caller
    const float *xi; uint32_t n, cap;
    __device__ __forceinline__ float next() { const float v = (n < cap) ? xi[n] : 0.5f; n++; return v; }
    __device__ __forceinline__ void next2(float &x, float &y) { x = next(); y = next(); }
    __device__ __forceinline__ void next3(float &x, float &y, float &z) { x = next(); y = next(); z = next(); }
};

// The reference's rejection loops are unbounded; both the oracle and this code stop
// after PT_MAX_TRIES and keep the last draw, so every wave reaches the loop's exit.
#ifndef PT_MAX_TRIES // overridden only by timing experiments (tools/build_variant.sh)
#define PT_MAX_TRIES 256
#endif

// material.zig:27-34 — components in [0,1): positive octant of the unit ball
template <class R>
__device__ __forceinline__ V3 random_vector_in_unit_sphere(R &r) {
    V3 p = v3(0.f, 0.f, 0.f);
#ifndef PT_MAX_TRIES_SPHERE
#define PT_MAX_TRIES_SPHERE PT_MAX_TRIES
#endif
    for (int tries = 0; tries < PT_MAX_TRIES_SPHERE; ++tries) {
        float x, y, z;
        r.next3(x, y, z);
        p = v3(x, y, z);
        if (vlen2(p) < 1.0f) break;
    }
    return p;
}
// camera.zig:113-120
template <class R>
__device__ __forceinline__ V3 random_vector_in_unit_disk(R &r) {
    V3 p = v3(0.f, 0.f, 0.f);
    for (int tries = 0; tries < PT_MAX_TRIES; ++tries) {
        float xi0, xi1;
        r.next2(xi0, xi1);
        const float x = 2.0f * xi0 - 1.0f;
        const float y = 2.0f * xi1 - 1.0f;
        p = v3(x, y, 0.0f);
        if (vlen2(p) < 1.0f) break;
    }
    return p;
}

struct Cam { // pt_camera (camera.zig:81-96)
    V3 lookfrom, u, v, llc, horizontal, vertical;
    float lens_radius;
};

// camera.zig:99-106
template <class R>
__device__ __forceinline__ void cast_ray(const Cam &c, float s, float t, R &r, V3 &origin, V3 &direction) {
    const V3 rd = vmul(random_vector_in_unit_disk(r), c.lens_radius);
    const V3 offset = vadd(vmul(c.u, rd.x), vmul(c.v, rd.y));
    origin = vadd(c.lookfrom, offset);
    direction = vunit(vsub(vsub(vadd(vadd(c.llc, vmul(c.horizontal, s)), vmul(c.vertical, t)), c.lookfrom), offset));
}

// sphere.zig:63-86: the near root of one sphere, or "no hit".  `a` = |d|^2 is the
// same value for every sphere of a ray and is passed in.
__device__ __forceinline__ bool sphere_near_root(V3 o, V3 d, float a, float cx, float cy, float cz, float radius,
                                                 float t_min, float t_max, float &t_out) {
    const V3 oc = vsub(o, v3(cx, cy, cz));
    const float half_b = vdot(oc, d);
    const float c = vlen2(oc) - radius * radius;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return false;
    // Shortcut with the reference's outcome: half_b >= 0 makes t = (-half_b - sqrtd) / a <= 0, which fails
    // t > t_min for any t_min >= 0 (NaN operands fail it too) — the sphere is behind the origin, or the ray
    // is leaving its surface.  About half of the filter's survivors end here, before the sqrt and the divide.
    if (t_min >= 0.0f && half_b >= 0.0f) return false;
    const float sqrtd = sqrtf(discriminant);
    const float t = (-half_b - sqrtd) / a;
    if (t < t_max && t > t_min) { t_out = t; return true; }
    return false;
}
// sphere.zig:87-88 (+ ray.zig:18): hit point and outward normal
__device__ __forceinline__ void sphere_hit_point(V3 o, V3 d, float t, float cx, float cy, float cz, float radius, V3 &p,
                                                 V3 &n) {
    p = vadd(o, vmul(d, t));
    n = vunit(vmul(vsub(p, v3(cx, cy, cz)), 1.0f / radius));
}
// The same with 1/radius supplied (the scene tables hold the correctly rounded quotient, computed
// once per sphere on the host: the same f32 value the division above yields).
__device__ __forceinline__ void sphere_hit_point_inv(V3 o, V3 d, float t, float cx, float cy, float cz, float inv_radius,
                                                     V3 &p, V3 &n) {
    p = vadd(o, vmul(d, t));
    n = vunit(vmul(vsub(p, v3(cx, cy, cz)), inv_radius));
}

// material.zig:59-65
__device__ __forceinline__ V3 reflect(V3 d, V3 n) { return vsub(d, vmul(n, 2.0f * vdot(d, n))); }

// material.zig:98-116
__device__ __forceinline__ bool refract(V3 d, V3 n, float n1_over_n2, V3 &out) {
    const float dn = vdot(vmul(d, -1.0f), n);
    const float cos_theta = (dn < 1.0f) ? dn : 1.0f;
    const V3 r_perpendicular = vmul(vadd(d, vmul(n, cos_theta)), n1_over_n2);
    const float discriminant = 1.0f - vlen2(r_perpendicular);
    if (discriminant < 0.0f) return false;
    const V3 r_parallel = vmul(n, -1.0f * sqrtf(discriminant));
    out = vadd(r_perpendicular, r_parallel);
    return true;
}

// material.zig:120-124; Zig's pow(x,5) = x * ((x*x)*(x*x)) (square-and-multiply on mantissas)
__host__ __device__ __forceinline__ float schlick_r0(float refraction_index) { // material.zig:121-122
    const float r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    return r0 * r0;
}
__device__ __forceinline__ float schlick_from_r0(float cosine, float r0) {       // material.zig:123
    const float x = 1.0f - cosine;
    const float x2 = x * x;
    const float x4 = x2 * x2;
    return r0 + (1.0f - r0) * (x * x4);
}
__device__ __forceinline__ float schlick(float cosine, float refraction_index) {
    return schlick_from_r0(cosine, schlick_r0(refraction_index));
}

// Material.scatter, material.zig:191-198 with 43-51 (Lambertian), 79-88 (Metal),
// 129-167 (Dielectric).  The attenuation is looked up from the material table when
// the path is finished (it is the albedo, or (1,1,1) for a dielectric).
//
// Per-sphere and per-ray values the reference recomputes at every hit are passed in: inv_param =
// 1/refraction_index and r0 = schlick_r0(refraction_index) (dielectrics only; the scene tables
// hold them), d_len = |d| (the caller already has |d|^2).  Each is the f32 value the reference's
// own expression yields, so the result is unchanged.
template <class R>
__device__ __forceinline__ V3 scatter_pre(uint32_t kind, float param, float inv_param, float r0, float d_len, V3 d, V3 p, V3 n,
                                          R &r) {
    V3 out;
    if (kind != 2u) {
        const V3 rnd = random_vector_in_unit_sphere(r);
        V3 pre; // the vector both kinds normalise
        if (kind == 0u) {
            const V3 target = vadd(p, vadd(n, rnd));
            pre = vsub(target, p);
        } else {
            const V3 sc = reflect(d, n);
            const V3 fc = vmul(rnd, param);
            pre = vadd(sc, fc);
        }
        out = vunit(pre);
    } else {
        V3 outward_normal;
        float n1_over_n2, cosine;
        const float dn = vdot(d, n);
        if (dn > 0.0f) {
            outward_normal = vunit(v3(-n.x, -n.y, -n.z));
            n1_over_n2 = param;
            cosine = param * dn / d_len;
        } else {
            outward_normal = n;
            n1_over_n2 = inv_param;
            cosine = -dn / d_len;
        }
        V3 refracted;
        if (refract(d, outward_normal, n1_over_n2, refracted)) {
            const float reflection_prob = schlick_from_r0(cosine, r0);
            if (r.next() < reflection_prob) out = reflect(d, n);
            else out = refracted;
        } else {
            out = reflect(d, n);
        }
    }
    return out;
}

template <class R>
__device__ __forceinline__ V3 scatter(uint32_t kind, float param, V3 d, V3 p, V3 n, R &r) {
    return scatter_pre(kind, param, 1.0f / param, schlick_r0(param), vlen(d), d, p, n, r);
}

// illumination.zig:27-34; returns false for DirectionIsNotUnitVector.  len_u = |u|.
__device__ __forceinline__ bool lerp_sky_len(V3 u, float len_u, V3 p0, V3 p1, V3 &out) {
    if (fabsf(len_u - 1.0f) > 0.00001f) return false; // constants.zig:3
    const float t = (u.y + 1.0f) * 0.5f;
    out = vadd(vmul(p0, 1.0f - t), vmul(p1, t));
    return true;
}
__device__ __forceinline__ bool lerp_sky(V3 u, V3 p0, V3 p1, V3 &out) { return lerp_sky_len(u, vlen(u), p0, p1, out); }

} // namespace ptdev
