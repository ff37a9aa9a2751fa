// op_cost_bench.hip — issue cost (shader cycles per wave-instruction per SIMD, at the clock the run sustains) of the
// expensive operations of the path tracer outside the scan: the 64-bit multiply-add of Philox, 32-bit integer multiplies,
// the transcendental unit (rcp / sqrt / rsq), the correctly rounded f32 division and square root the library is built
// with (-fhip-fp32-correctly-rounded-divide-sqrt), and one whole Philox4x32-10 block.  8 independent chains per lane,
// 4 waves per SIMD.   Build: hipcc -O3 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -o tools/bin/op_cost_bench tools/op_cost_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 2048
#define CH 8

__device__ __forceinline__ void philox(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void philox_bitop3(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s0, uint32_t u0) {
    float f[CH], g[CH];
    uint32_t m[CH], n[CH];
    uint64_t w[CH];
    for (int i = 0; i < CH; ++i) { f[i] = 1.0f + threadIdx.x * 0.001f + i; g[i] = s0 + 0.37f * i; m[i] = threadIdx.x * 2654435761u + i; n[i] = u0 + i; w[i] = m[i]; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(g[i]), "v"(g[i]));
            if (MODE == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(m[i]), "v"(n[i]) : "vcc");
            if (MODE == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(m[i]) : "v"(n[i]));
            if (MODE == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(m[i]) : "v"(n[i]));
            if (MODE == 4) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(f[i]));
            if (MODE == 5) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(f[i]));
            if (MODE == 6) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(f[i]));
            if (MODE == 7) f[i] = g[i] / f[i];            // correctly rounded division
            if (MODE == 8) f[i] = sqrtf(f[i] + g[i]);     // correctly rounded square root (+ one add)
            if (MODE == 9) { uint32_t a = m[i], b = n[i], c = m[i] ^ 5u, d = n[i] + 7u; philox(a, b, c, d, u0, u0 + 1u); m[i] = a ^ b; n[i] = c ^ d; }
            if (MODE == 12) { uint32_t a = m[i], b = n[i], c = m[i] ^ 5u, d = n[i] + 7u; philox_bitop3(a, b, c, d, u0, u0 + 1u); m[i] = a ^ b; n[i] = c ^ d; }
            if (MODE == 13) m[i] = __builtin_amdgcn_bitop3_b32(m[i], n[i], u0, 0x96);
            if (MODE == 10) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(m[i]) : "v"(n[i]));
            if (MODE == 11) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(m[i]) : "v"(n[i]));
        }
    }
    float acc = 0.f;
    for (int i = 0; i < CH; ++i) acc += f[i] + (float)m[i] + (float)n[i] + (float)(uint32_t)w[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char *name, int insts_per_op, float *d_out) {
    const int grid = 256 * 4; // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 0.5f, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 0.5f, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops_per_simd = (double)ITERS * CH * 4.0; // per SIMD: 4 waves x ITERS x CH
    const double ns_per_op = ms * 1e6 / ops_per_simd;
    printf("%-44s %8.3f ms  %7.2f ns per op per SIMD = %6.1f cycles @2.4 GHz (%d op%s)\n", name, ms, ns_per_op, ns_per_op * 2.4, insts_per_op, insts_per_op > 1 ? "s: whole sequence" : "");
}

int main() {
    float *d_out;
    hipMalloc(&d_out, 256 * 4 * 256 * sizeof(float));
    run<0>("v_fma_f32", 1, d_out);
    run<10>("v_xor_b32", 1, d_out);
    run<1>("v_mad_u64_u32", 1, d_out);
    run<2>("v_mul_lo_u32", 1, d_out);
    run<3>("v_mul_hi_u32", 1, d_out);
    run<11>("v_mul_u32_u24", 1, d_out);
    run<4>("v_rcp_f32", 1, d_out);
    run<5>("v_sqrt_f32", 1, d_out);
    run<6>("v_rsq_f32", 1, d_out);
    run<7>("a / b, correctly rounded", 2, d_out);
    run<8>("sqrtf(a + b), correctly rounded", 2, d_out);
    run<9>("Philox4x32-10 block + 4 xor/add", 2, d_out);
    run<12>("Philox4x32-10 block, v_bitop3 xors", 2, d_out);
    run<13>("v_bitop3_b32 (a ^ b ^ sgpr)", 1, d_out);
    hipFree(d_out);
    return 0;
}
