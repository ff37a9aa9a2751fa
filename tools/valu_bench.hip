// valu_bench.hip — measures the f32 VALU issue rates this design is priced against
// (v_fma_f32, v_pk_fma_f32, v_mul/v_add with an SGPR operand) at 1..8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_bench tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1) {
    float a[8];
    float2_ p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2_{a[i], a[i] + 0.5f}; }
    const float x = s0, y = s1;
    const float2_ x2 = {s0, s0}, y2 = {s1, s1};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], x, y);                       // v_fma_f32 (SGPR operand)
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], x2, y2);          // v_pk_fma_f32
            if (MODE == 2) { a[i] = a[i] * x; a[i] = a[i] + y; }                     // v_mul + v_add
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(int blocks, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    float *d; hipMalloc(&d, sizeof(float) * 256 * cus * 8);
    const char *names[3] = {"v_fma_f32", "v_pk_fma_f32", "v_mul+v_add"};
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * wps; // 256-thread blocks: 4 waves = 1 per SIMD
        double ms[3] = {run<0>(blocks, d), run<1>(blocks, d), run<2>(blocks, d)};
        for (int m = 0; m < 3; ++m) {
            const double lanes = (double)blocks * 256;
            const double inst = lanes / 64.0 * ITERS * 8 * (m == 2 ? 2 : 1);
            const double flop = lanes * ITERS * 8 * (m == 1 ? 4 : 2);
            printf("waves/SIMD %d %-12s %8.3f ms  %7.2f TFLOP/s  %6.2f cyc/wave-inst/SIMD @2.4GHz\n", wps, names[m], ms[m],
                   flop / ms[m] * 1e-9, ms[m] * 1e-3 * 2.4e9 / (inst / (cus * 4.0)));
        }
    }
    return 0;
}
