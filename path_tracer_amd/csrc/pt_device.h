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
// ---- correctly rounded sqrt / reciprocal in fewer issue slots ----------------------------------------------------------
// The reference's arithmetic needs IEEE-correct sqrt and division (the oracle is gcc on the CPU).  The compiler's sequences
// for them (-fhip-fp32-correctly-rounded-divide-sqrt) cost 19 and 15 FMA issue slots (tools/op_cost_bench.hip): they scale
// denormal / extreme operands first and fix special values last.  For operands in 2^-96 <= x < 2^96 none of that applies, and
// the core is: v_sqrt_f32 (1 ulp) + the compiler's own one-ulp-up / one-ulp-down residual test; v_rcp_f32 (1 ulp) + one
// Newton step in FMA.  Operands outside the range (and negative, zero, infinite, NaN ones) take the compiler's sequence.
// PROVEN BY EXHAUSTION: every one of the 2^32 bit patterns gives the same bits as sqrtf(x), 1.0f / x and 1.0f / sqrtf(x)
// (tools/exhaustive_math.hip; re-checked on every GPU test run through PT_FN_EXHAUSTIVE_*, tests/test_gpu_lean_math.py).
__device__ __forceinline__ bool lean_range_sqrt(float x) { return __float_as_uint(x) - 0x0f800000u < 0x70000000u; } // 2^-96 <= x < inf
__device__ __forceinline__ bool lean_range_rcp(float x) { return __float_as_uint(x) - 0x0f800000u < 0x60000000u; }  // 2^-96 <= x < 2^96
__device__ __forceinline__ float sqrt_core(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = __builtin_fmaf(-dn, s, x), vs = __builtin_fmaf(-up, s, x);
    s = (vp <= 0.0f) ? dn : s;
    s = (vs > 0.0f) ? up : s;
    return s;
}
__device__ __forceinline__ float rcp_core(float s) {
    const float r = __builtin_amdgcn_rcpf(s);
    return __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
}
#ifndef PT_LEAN_MATH
#define PT_LEAN_MATH 1 // 0: the compiler's sequences everywhere (A/B builds)
#endif
__device__ __forceinline__ float sqrt_rn(float x) {          // == sqrtf(x)
    if (PT_LEAN_MATH && lean_range_sqrt(x)) return sqrt_core(x);
    return sqrtf(x);
}
__device__ __forceinline__ float rcp_rn(float x) {           // == 1.0f / x
    if (PT_LEAN_MATH && lean_range_rcp(x)) return rcp_core(x);
    return 1.0f / x;
}
__device__ __forceinline__ float inv_sqrt_rn(float x) {      // == 1.0f / sqrtf(x): two roundings, as vec3.zig:61-64 takes them
    if (PT_LEAN_MATH && lean_range_sqrt(x)) return rcp_core(sqrt_core(x));
    return 1.0f / sqrtf(x);
}
// illumination.zig:28's test |sqrt(a) - 1| > 1e-5 on a = |d|^2 (>= 0 or NaN) without the square root: sqrt is monotone, so the
// test is two comparisons of a against the f32 thresholds where it flips (found and checked by exhaustion, as above)
__device__ __forceinline__ bool not_unit_len2(float a) {
#if PT_LEAN_MATH
    return a < __uint_as_float(0x3f7ffeb2u) || a > __uint_as_float(0x3f8000a7u);
#else
    return fabsf(sqrtf(a) - 1.0f) > 0.00001f;
#endif
}

// vec3.zig:67-69: left-to-right sum of products, no FMA
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vec3.zig:37-39: pow(x,2)+pow(y,2)+pow(z,2); Zig's pow(x,2) is x*x (mantissa square, one rounding)
__device__ __forceinline__ float vlen2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float vlen(V3 a) { return sqrt_rn(vlen2(a)); } // vec3.zig:30-32
// vec3.zig:61-64: multiply by the reciprocal of the norm
__device__ __forceinline__ V3 vunit(V3 a) {
    const float inv_n = inv_sqrt_rn(vlen2(a));
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
        // three-way XOR in ONE instruction: gfx950's v_bitop3_b32 with truth table 0x96 (a ^ b ^ c); the compiler emits two
        // v_xor_b32 for the C expression.  Measured (tools/op_cost_bench.hip): a Philox block 77 -> 60 FMA-issue-equivalents.
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// Random.float(f32) of Zig 0.9.1: bits 0x3F800000 | (s >> 9) in [1,2), minus 1.
__device__ __forceinline__ float u32_to_float(uint32_t s) { return __uint_as_float(0x3f800000u | (s >> 9)) - 1.0f; }

struct PhiloxRng { // stochastic mode
    uint32_t k0, k1, pixel, sample, n; // n: draws taken so far; bit 31 = "block buffer not loaded" (set by resume())
    uint32_t b0, b1, b2, b3;
    __device__ __forceinline__ void init(uint32_t seed_lo, uint32_t seed_hi) {
        k0 = seed_lo; k1 = seed_hi; pixel = sample = n = 0; b0 = b1 = b2 = b3 = 0;
    }
    __device__ __forceinline__ void start_path(uint32_t px, uint32_t s) { pixel = px; sample = s; n = 0; }
    // Continue the stream of path (px, s) after its first n_drawn draws (the primary pass hands paths over
    // this way).  The words left in the current block are not carried along: the block is regenerated by
    // the first draw that wants one of them — at the Philox call that draw would make anyway when it opens
    // a block, so no extra program point.
    __device__ __forceinline__ void resume(uint32_t px, uint32_t s, uint32_t n_drawn) {
        pixel = px; sample = s; n = n_drawn | ((n_drawn & 3u) ? 0x80000000u : 0u);
    }
    __device__ __forceinline__ uint32_t drawn() const { return n & 0x7fffffffu; }
    __device__ __forceinline__ float next() {
        const bool stale = (int32_t)n < 0;
        const uint32_t nn = n & 0x7fffffffu;
        const uint32_t k = nn & 3u;
        if (k == 0 || stale) philox4x32_10(pixel, sample, nn >> 2, 0u, k0, k1, b0, b1, b2, b3);
        // Select among VALUES read unconditionally: with the member reads inside the arms of ?: the optimizer
        // sinks them into one load from a selected ADDRESS, and the whole struct then lives in scratch memory
        // (seen in the ISA as scratch_store/scratch_load around every Philox call).
        const uint32_t x0 = b0, x1 = b1, x2 = b2, x3 = b3;
        const uint32_t w = (k == 0) ? x0 : (k == 1) ? x1 : (k == 2) ? x2 : x3;
        n = nn + 1u;
        return u32_to_float(w);
    }
    // Two / three draws that never straddle a Philox block: if fewer words than needed
    // are left in the current block, the rest of the block is skipped ("reserve").  A
    // rejection-loop try therefore runs Philox at most once, and at ONE program point for
    // all lanes of a wave, instead of possibly at each of its draws.
    __device__ __forceinline__ void next2(float &x, float &y) {
        const bool stale = (int32_t)n < 0;
        uint32_t nn = n & 0x7fffffffu;
        uint32_t k = nn & 3u;
        if (k == 3u) { nn += 1u; k = 0u; }
        if (k == 0u || stale) philox4x32_10(pixel, sample, nn >> 2, 0u, k0, k1, b0, b1, b2, b3);
        const uint32_t x0 = b0, x1 = b1, x2 = b2, x3 = b3; // values, not addresses: see next()
        const uint32_t w0 = (k == 0u) ? x0 : (k == 1u) ? x1 : x2;
        const uint32_t w1 = (k == 0u) ? x1 : (k == 1u) ? x2 : x3;
        n = nn + 2u;
        x = u32_to_float(w0); y = u32_to_float(w1);
    }
    __device__ __forceinline__ void next3(float &x, float &y, float &z) {
        const bool stale = (int32_t)n < 0;
        uint32_t nn = n & 0x7fffffffu;
        uint32_t k = nn & 3u;
        if (k > 1u) { nn = (nn + 3u) & ~3u; k = 0u; }
        if (k == 0u || stale) philox4x32_10(pixel, sample, nn >> 2, 0u, k0, k1, b0, b1, b2, b3);
        const uint32_t x0 = b0, x1 = b1, x2 = b2, x3 = b3; // values, not addresses: see next()
        const uint32_t w0 = (k == 0u) ? x0 : x1;
        const uint32_t w1 = (k == 0u) ? x1 : x2;
        const uint32_t w2 = (k == 0u) ? x2 : x3;
        n = nn + 3u;
        x = u32_to_float(w0); y = u32_to_float(w1); z = u32_to_float(w2);
    }
};
struct ConstRng { // deterministic mode: every draw is 0.5
    __device__ __forceinline__ void init(uint32_t, uint32_t) {}
    __device__ __forceinline__ void start_path(uint32_t, uint32_t) {}
    __device__ __forceinline__ void resume(uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ uint32_t drawn() const { return 0u; }
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

// ---- PT_SAMPLER_DIRECT (include/ptgpu.h): the same two distributions without the rejection loops ----------------
// NOT in the reference (SURVEY.md 9.4, 9.5 name the equivalents): an opt-in.  Plain f32 multiplies and adds, one
// correctly rounded sqrt and integer bit operations — no sin/cos/cbrt of any library — in exactly the operations of
// oracle/pt_oracle.c (this file is compiled with -ffp-contract=off), so that the GPU and the oracle agree bit for bit.
// sin and cos of f * pi/2, f in [0,1): folded to [0, pi/4], Taylor polynomials
__device__ __forceinline__ void sincos_quarter(float f, float &s_out, float &c_out) {
    const bool swap = f > 0.5f;
    const float g = swap ? 1.0f - f : f;
    const float th = g * 1.57079632679489661923f;
    const float t2 = th * th;
    float sp = 2.75573192239858906526e-6f;
    sp = sp * t2 + -1.98412698412698412698e-4f;
    sp = sp * t2 + 8.33333333333333333333e-3f;
    sp = sp * t2 + -1.66666666666666666667e-1f;
    sp = sp * t2 + 1.0f;
    const float sn = th * sp;
    float cp = 2.48015873015873015873e-5f;
    cp = cp * t2 + -1.38888888888888888889e-3f;
    cp = cp * t2 + 4.16666666666666666667e-2f;
    cp = cp * t2 + -0.5f;
    cp = cp * t2 + 1.0f;
    s_out = swap ? cp : sn;
    c_out = swap ? sn : cp;
}
// x^(1/3), x in [0,1): x * r^2, r = x^(-1/3) from the exponent trick and three Newton steps r <- r (4 - x r^3) / 3
__device__ __forceinline__ float cbrt_unit(float x) {
    const uint32_t i = 0x54a21d2au - __float_as_uint(x) / 3u;
    float r = __uint_as_float(i);
    const float third = 1.0f / 3.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float r3 = (r * r) * r;
        r = (r * (4.0f - x * r3)) * third;
    }
    return (x == 0.0f) ? 0.0f : x * (r * r);
}
// uniform in the unit disk (the distribution of camera.zig:113-120): sqrt(xi0) * (cos, sin)(2 pi xi1)
__device__ __forceinline__ V3 direct_vector_in_unit_disk(float xi0, float xi1) {
    const float q = xi1 * 4.0f;
    const uint32_t k = (uint32_t)q;
    float s, c;
    sincos_quarter(q - (float)k, s, c);
    const float cx = (k == 0u) ? c : (k == 1u) ? -s : (k == 2u) ? -c : s;
    const float sy = (k == 0u) ? s : (k == 1u) ? c : (k == 2u) ? -s : -c;
    const float rr = sqrt_rn(xi0);
    return v3(rr * cx, rr * sy, 0.0f);
}
// uniform in the positive octant of the unit ball (the distribution of material.zig:27-34): direction with z = xi0
// uniform (equal areas), azimuth xi1 * pi/2, radius xi2^(1/3)
__device__ __forceinline__ V3 direct_vector_in_unit_sphere(float xi0, float xi1, float xi2) {
    float s, c;
    sincos_quarter(xi1, s, c);
    const float sxy = sqrt_rn(1.0f - xi0 * xi0);
    const float rho = cbrt_unit(xi2);
    return v3(rho * (sxy * c), rho * (sxy * s), rho * xi0);
}

// material.zig:27-34 — components in [0,1): positive octant of the unit ball
template <bool DIRECT = false, class R>
__device__ __forceinline__ V3 random_vector_in_unit_sphere(R &r) {
    V3 p = v3(0.f, 0.f, 0.f);
    if (DIRECT) {
        float x, y, z;
        r.next3(x, y, z);
        return direct_vector_in_unit_sphere(x, y, z);
    }
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
template <bool DIRECT = false, class R>
__device__ __forceinline__ V3 random_vector_in_unit_disk(R &r) {
    V3 p = v3(0.f, 0.f, 0.f);
    if (DIRECT) {
        float xi0, xi1;
        r.next2(xi0, xi1);
        return direct_vector_in_unit_disk(xi0, xi1);
    }
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

// camera.zig:99-106, from the unit-disk draw on (the primary pass wants to see that draw)
__device__ __forceinline__ void cast_ray_from_disk(const Cam &c, float s, float t, V3 disk, V3 &origin, V3 &direction) {
    const V3 rd = vmul(disk, c.lens_radius);
    const V3 offset = vadd(vmul(c.u, rd.x), vmul(c.v, rd.y));
    origin = vadd(c.lookfrom, offset);
    direction = vunit(vsub(vsub(vadd(vadd(c.llc, vmul(c.horizontal, s)), vmul(c.vertical, t)), c.lookfrom), offset));
}
template <bool DIRECT = false, class R>
__device__ __forceinline__ void cast_ray(const Cam &c, float s, float t, R &r, V3 &origin, V3 &direction) {
    cast_ray_from_disk(c, s, t, random_vector_in_unit_disk<DIRECT>(r), origin, direction);
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
    const float sqrtd = sqrt_rn(discriminant);
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
    const V3 r_parallel = vmul(n, -1.0f * sqrt_rn(discriminant));
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
// hold them), len2 = |d|^2 (the caller has it; |d| is taken where a dielectric needs it).  Each is the f32 value the
// reference's own expression yields, so the result is unchanged.
template <bool DIRECT = false, class R>
__device__ __forceinline__ V3 scatter_pre(uint32_t kind, float param, float inv_param, float r0, float len2, V3 d, V3 p, V3 n,
                                          R &r) {
    V3 out;
    if (kind != 2u) {
        const V3 rnd = random_vector_in_unit_sphere<DIRECT>(r);
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
        const float d_len = sqrt_rn(len2); // vlen(ray.direction), material.zig:140,144
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
    return scatter_pre(kind, param, 1.0f / param, schlick_r0(param), vlen2(d), d, p, n, r);
}

// illumination.zig:27-34; returns false for DirectionIsNotUnitVector.  len2_u = |u|^2.
__device__ __forceinline__ bool lerp_sky_len2(V3 u, float len2_u, V3 p0, V3 p1, V3 &out) {
    if (not_unit_len2(len2_u)) return false; // |sqrt(len2) - 1| > epsilon, constants.zig:3
    const float t = (u.y + 1.0f) * 0.5f;
    out = vadd(vmul(p0, 1.0f - t), vmul(p1, t));
    return true;
}
__device__ __forceinline__ bool lerp_sky(V3 u, V3 p0, V3 p1, V3 &out) { return lerp_sky_len2(u, vlen2(u), p0, p1, out); }

} // namespace ptdev
