// issue_bench.hip — issue cost (cycles per wave-instruction per SIMD) of the instructions the scan loop of
// pt_megakernel uses or could use, with the operand forms the kernel has (all-VGPR, explicit e32/e64
// encodings), and of whole per-test bodies.  Each mode runs 8 independent dependency chains per wave.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/issue_bench tools/issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 8192
#define UNROLL 8
typedef float float2_ __attribute__((ext_vector_type(2)));

struct Mode { const char *name; int n_inst; };
static const Mode kModes[] = {
    {"v_fmac_f32_e32 (vgpr)", 1},        // 0
    {"v_fma_f32 e64 (vgpr)", 1},         // 1
    {"v_mul_f32_e32 (vgpr)", 1},         // 2
    {"v_max_f32_e32 (vgpr)", 1},         // 3
    {"v_cmp_le_f32_e32 vcc", 1},         // 4
    {"v_addc_co_u32_e32", 1},            // 5
    {"v_and_b32_e32", 1},                // 6
    {"v_pk_fma_f32 (vgpr)", 1},          // 7
    {"v_alignbit_b32", 1},               // 8
    {"v_lshlrev_b32_e32", 1},            // 9
    {"v_cndmask_b32_e32", 1},            // 10
    {"v_cmp_le_f32_e64 sgpr", 1},        // 11
    {"v_sub_f32_e32 (vgpr)", 1},         // 12
    {"test 7fmac+cmp+addc", 9},          // 13
    {"test 7fmac+sub+alignbit", 9},      // 14
    {"test 7fmac only", 7},              // 15
    {"2 tests packed 7pk+2cmp+2addc", 11}, // 16
    {"v_fma_f32 e64 (1 sgpr)", 1},       // 17
    {"v_fmac_f32_e32 (sgpr src0)", 1},   // 18
    {"v_add_f32_e32 (vgpr)", 1},         // 19
    {"v_xor_b32_e32", 1},                // 20
    {"v_mov_b32_e32", 1},                // 21
};
constexpr int kNumModes = sizeof(kModes) / sizeof(kModes[0]);

#define TEST_FMAS(q, s, cx, cy, cz)                                                                                      \
    asm volatile("v_mul_f32_e32 %0, %2, %3\n\tv_fma_f32 %1, %3, %8, %11\n\tv_fmac_f32_e32 %0, %4, %5\n\t"             \
                 "v_fmac_f32_e32 %1, %5, %9\n\tv_fmac_f32_e32 %0, %6, %7\n\tv_fmac_f32_e32 %1, %7, %10\n\t"          \
                 "v_fmac_f32_e32 %1, %0, %0"                                                                            \
                 : "=&v"(q), "=&v"(s) : "v"(hx), "v"(cx), "v"(hy), "v"(cy), "v"(hz), "v"(cz), "v"(wx), "v"(wy), "v"(wz), "v"(E))

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s0) {
    float f[UNROLL], g[UNROLL], h[UNROLL];
    float2_ p[UNROLL], p2[UNROLL], p3[UNROLL];
    uint32_t m[UNROLL];
    for (int i = 0; i < UNROLL; ++i) {
        f[i] = threadIdx.x * 0.001f + i; g[i] = f[i] * 0.5f; h[i] = f[i] * 0.25f; m[i] = threadIdx.x + i;
        p[i] = float2_{f[i], g[i]}; p2[i] = float2_{g[i], h[i]}; p3[i] = float2_{h[i], f[i]};
    }
    const float hx = f[0] * 0.3f, hy = f[0] * 0.2f, hz = f[0] * 0.1f, wx = g[0], wy = g[0] * 2.f, wz = g[0] * 3.f, E = g[0] * 4.f;
    const float2_ hx2 = {hx, hx}, hy2 = {hy, hy}, hz2 = {hz, hz}, wx2 = {wx, wx}, wy2 = {wy, wy}, wz2 = {wz, wz}, E2 = {E, E};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (MODE == 0) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[i]) : "v"(g[i]), "v"(h[i]));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(g[i]), "v"(h[i]));
            if (MODE == 2) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(f[i]) : "v"(g[i]));
            if (MODE == 3) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(f[i]) : "v"(g[i]));
            if (MODE == 4) asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" : : "v"(g[i]), "v"(f[i]) : "vcc");
            if (MODE == 5) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m[i]) : : "vcc");
            if (MODE == 6) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(m[i]) : "v"(g[i]));
            if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p2[i]), "v"(p3[i]));
            if (MODE == 8) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(m[i]) : "v"(f[i]));
            if (MODE == 9) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(m[i]));
            if (MODE == 10) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(m[i]) : "v"(g[i]) : "vcc");
            if (MODE == 11) asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" : : "v"(g[i]), "v"(f[i]) : "s20", "s21");
            if (MODE == 12) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(g[i]));
            if (MODE == 13) {
                float q, s;
                TEST_FMAS(q, s, g[i], h[i], f[i]);
                asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(m[i]) : "v"(g[i]), "v"(s) : "vcc");
            }
            if (MODE == 14) {
                float q, s;
                TEST_FMAS(q, s, g[i], h[i], f[i]);
                asm volatile("v_sub_f32_e32 %1, %1, %2\n\tv_alignbit_b32 %0, %0, %1, 31" : "+v"(m[i]), "+v"(s) : "v"(g[i]));
            }
            if (MODE == 15) {
                float q, s;
                TEST_FMAS(q, s, g[i], h[i], f[i]);
                asm volatile("" : : "v"(s));
            }
            if (MODE == 16) { // two spheres per packed instruction: c = {c_a, c_b} per component
                float2_ q, s;
                asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_fma_f32 %1, %3, %8, %11\n\tv_pk_fma_f32 %0, %4, %5, %0\n\t"
                             "v_pk_fma_f32 %1, %5, %9, %1\n\tv_pk_fma_f32 %0, %6, %7, %0\n\tv_pk_fma_f32 %1, %7, %10, %1\n\t"
                             "v_pk_fma_f32 %1, %0, %0, %1"
                             : "=&v"(q), "=&v"(s)
                             : "v"(hx2), "v"(p[i]), "v"(hy2), "v"(p2[i]), "v"(hz2), "v"(p3[i]), "v"(wx2), "v"(wy2), "v"(wz2), "v"(E2));
                asm volatile("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
                             "v_cmp_le_f32_e32 vcc, %3, %4\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
                             : "+v"(m[i]) : "v"(g[i]), "v"(s.x), "v"(h[i]), "v"(s.y) : "vcc");
            }
            if (MODE == 17) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "s"(s0), "v"(h[i]));
            if (MODE == 18) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[i]) : "s"(s0), "v"(h[i]));
            if (MODE == 19) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(f[i]) : "v"(g[i]));
            if (MODE == 20) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(m[i]) : "v"(g[i]));
            if (MODE == 21) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(m[i]) : "v"(g[i]));
        }
    }
    float r = 0;
    for (int i = 0; i < UNROLL; ++i) r += f[i] + g[i] + h[i] + (float)m[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(int blocks, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0;
}

template <int MODE>
void run_all(int blocks, float *d, double *ms) {
    ms[MODE] = run<MODE>(blocks, d);
    if constexpr (MODE + 1 < kNumModes) run_all<MODE + 1>(blocks, d, ms);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    float *d; hipMalloc(&d, sizeof(float) * 256 * cus * 8);
    for (int wps = 4; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;
        double ms[kNumModes];
        run_all<0>(blocks, d, ms);
        for (int m = 0; m < kNumModes; ++m) {
            const double slots = (double)blocks * 4 * ITERS * UNROLL;          // wave-slots
            const double cyc = ms[m] * 1e-3 * 2.4e9 / (slots / (cus * 4.0));   // cycles per slot per SIMD @2.4 GHz
            printf("waves/SIMD %d %-32s %8.3f ms  %6.2f cyc/slot  %6.2f cyc/inst @2.4GHz\n", wps, kModes[m].name, ms[m], cyc,
                   cyc / kModes[m].n_inst);
        }
    }
    return 0;
}
