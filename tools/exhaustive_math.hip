// exhaustive_math.hip — proof by exhaustion for the lean unary f32 sequences of pt_device.h: every one of the 2^32 bit patterns
// through the lean sequence and through the compiler's correctly rounded expression (-fhip-fp32-correctly-rounded-divide-sqrt),
// mismatches counted (NaN == NaN).   Build: hipcc -O3 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -o tools/bin/exhaustive_math tools/exhaustive_math.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ bool in_fast_range(float x) { return __float_as_uint(x) - 0x0f800000u < 0x70000000u; } // 2^-96 <= x < inf

__device__ __forceinline__ float sqrt_core(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = __builtin_fmaf(-dn, s, x), vs = __builtin_fmaf(-up, s, x);
    s = (vp <= 0.0f) ? dn : s;
    s = (vs > 0.0f) ? up : s;
    return s;
}
__device__ __forceinline__ float sqrt_lean(float x) { return in_fast_range(x) ? sqrt_core(x) : sqrtf(x); }

template <int STEPS>
__device__ __forceinline__ float rcp_core(float s) {
    float r = __builtin_amdgcn_rcpf(s);
#pragma unroll
    for (int k = 0; k < STEPS; ++k) { const float e = __builtin_fmaf(-s, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    return r;
}
// the compiler's own chain for 1/s without scale / fixup
__device__ __forceinline__ float rcp_chain(float s) {
    float r = __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = r;
    e = __builtin_fmaf(-s, q, 1.0f);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-s, q, 1.0f);
    return __builtin_fmaf(e, r, q);
}
template <int V>
__device__ __forceinline__ float inv_sqrt_lean(float x) {
    if (in_fast_range(x)) {
        const float s = sqrt_core(x);
        return V == 9 ? rcp_chain(s) : rcp_core<V>(s);
    }
    return 1.0f / sqrtf(x);
}
template <int V>
__device__ __forceinline__ float rcp_lean(float x) { // 1/x for 2^-96 <= x < 2^96 on the fast path
    if (__float_as_uint(x) - 0x0f800000u < 0x60000000u) return V == 9 ? rcp_chain(x) : rcp_core<V>(x);
    return 1.0f / x;
}

__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

template <int FN>
__global__ void check(unsigned long long *bad, uint32_t *first_bad) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = 0;
    for (uint32_t hi = 0; hi < 256u; ++hi) {
        const uint32_t bits = (hi << 24) | tid; // tid covers the low 24 bits
        const float x = __uint_as_float(bits);
        bool ok = true;
        if (FN == 0) ok = same(sqrt_lean(x), sqrtf(x));
        if (FN == 1) ok = same(inv_sqrt_lean<1>(x), 1.0f / sqrtf(x));
        if (FN == 2) ok = same(inv_sqrt_lean<2>(x), 1.0f / sqrtf(x));
        if (FN == 3) ok = same(inv_sqrt_lean<9>(x), 1.0f / sqrtf(x));
        if (FN == 4) ok = same(rcp_lean<1>(x), 1.0f / x);
        if (FN == 5) ok = same(rcp_lean<2>(x), 1.0f / x);
        if (FN == 6) ok = same(rcp_lean<9>(x), 1.0f / x);
        if (FN == 8) ok = same(rcp_lean<0>(x), 1.0f / x);   // control: the bare v_rcp_f32 must NOT pass
        if (FN == 7) { // lerp's unit check (illumination.zig:28) on a = |d|^2 >= 0 or NaN, by two thresholds
            const bool ref = fabsf(sqrtf(x) - 1.0f) > 0.00001f;
            const bool lean = x < __uint_as_float(0x3f7ffeb2u) || x > __uint_as_float(0x3f8000a7u);
            ok = (bits >> 31) ? true : (ref == lean);
        }
        if (!ok) { if (n == 0) atomicMin(first_bad, bits); n++; }
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}

template <int FN>
static void run(const char *name) {
    unsigned long long *d_bad, h_bad = 0;
    uint32_t *d_first, h_first = 0xffffffffu;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 4);
    hipMemcpy(d_bad, &h_bad, 8, hipMemcpyHostToDevice); hipMemcpy(d_first, &h_first, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check<FN>, dim3((1u << 24) / 256), dim3(256), 0, 0, d_bad, d_first);
    hipDeviceSynchronize();
    hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&h_first, d_first, 4, hipMemcpyDeviceToHost);
    printf("%-58s mismatches %llu of 4294967296  (lowest bad bits 0x%08x)\n", name, h_bad, h_first);
    hipFree(d_bad); hipFree(d_first);
}

int main() {
    run<0>("sqrt_lean(x) vs sqrtf(x)");
    run<1>("inv_sqrt: sqrt_core + rcp + 1 Newton step vs 1/sqrtf(x)");
    run<2>("inv_sqrt: sqrt_core + rcp + 2 Newton steps");
    run<3>("inv_sqrt: sqrt_core + the compiler's fma chain");
    run<4>("rcp_lean: rcp + 1 Newton step vs 1.0f/x");
    run<5>("rcp_lean: rcp + 2 Newton steps");
    run<6>("rcp_lean: the compiler's fma chain");
    run<7>("unit check by thresholds on a (a >= 0 or NaN)");
    run<8>("control: bare v_rcp_f32 vs 1.0f/x (must mismatch)");
    return 0;
}
