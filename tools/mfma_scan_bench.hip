// mfma_scan_bench.hip — prices the two candidate inner loops of the sphere scan on gfx950:
//   VALU : per (ray, sphere) 7 FMA-class + cmp + addc, filter record broadcast from LDS (the shipped loop)
//   MFMA : q = c.h and s = c.w + E - T through v_mfma_f32_16x16x4_f32 (16 spheres x 16 rays x K=4 per
//          instruction, 8 instructions per 16 spheres x 64 rays), then F = fma(q,q,s) and one sign-bit
//          shift per pair on the VALU.
// Prints ns and cycles@2.4GHz per 64 (ray,sphere) tests per SIMD.  Results are dumped so nothing is DCE'd.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NSPH 512
#define REPS 200

__global__ __launch_bounds__(256) void k_valu(const float4 *filt, unsigned *out, float seed) {
    __shared__ float4 s_filt[NSPH];
    for (int i = threadIdx.x; i < NSPH; i += 256) s_filt[i] = filt[i];
    __syncthreads();
    float hx = threadIdx.x * 0.001f + seed, hy = hx * 0.5f, hz = 1.0f - hx, wx = hx + 2.f, wy = hy - 1.f, wz = hz * 2.f, E = hx * hy;
    unsigned m = 0, acc = 0;
    for (int rep = 0; rep < REPS; ++rep) {
        for (int j0 = 0; j0 < NSPH; j0 += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 f = s_filt[j0 + k];
                float q = f.x * hx; q = fmaf(f.y, hy, q); q = fmaf(f.z, hz, q);
                float s = fmaf(f.x, wx, E); s = fmaf(f.y, wy, s); s = fmaf(f.z, wz, s);
                const float F = fmaf(q, q, s);
                asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(f.w), "v"(F) : "vcc");
            }
            if ((j0 & 31) == 24) acc ^= m;
        }
        hx += 1e-6f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NMFMA>  // 8 = q and s on the matrix pipe; 6/4 = one/two ray groups' s on the VALU instead
__global__ __launch_bounds__(256) void k_mfma(const float *soa, const float *negT, unsigned *out, float seed) {
    __shared__ float s_a[4][NSPH + 16];   // [k][sphere]: cx, cy, cz, 1
    __shared__ float s_t[NSPH];           // -T per sphere
    for (int i = threadIdx.x; i < NSPH; i += 256) {
        for (int k = 0; k < 4; ++k) s_a[k][i] = soa[k * NSPH + i];
        s_t[i] = negT[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int i16 = lane & 15, kq = lane >> 4;
    // B operands: lane supplies component kq of ray (16 b + i16); synthetic values here
    float bq[4], bs[4];
    for (int b = 0; b < 4; ++b) { bq[b] = 0.01f * (b + 1) + lane * 0.001f + seed; bs[b] = 0.02f * (b + 1) - lane * 0.001f; }
    float wx[4], wy[4], wz[4], Ev[4];
    for (int b = 0; b < 4; ++b) { wx[b] = bq[b] + 1; wy[b] = bq[b] - 1; wz[b] = bq[b] * 2; Ev[b] = bs[b]; }
    unsigned m[4] = {0, 0, 0, 0}, acc = 0;
    for (int rep = 0; rep < REPS; ++rep) {
        for (int base = 0; base < NSPH; base += 16) {
            const float a = s_a[kq][base + i16];
            const f32x4 c = *reinterpret_cast<const f32x4 *>(&s_t[base + kq * 4]);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 q = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq[b], zero, 0, 0, 0);
                f32x4 s;
                if (b < NMFMA - 4) {
                    s = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bs[b], c, 0, 0, 0);
                } else {   // VALU stand-in of the same work: 3 FMAs per pair (operands would come from LDS)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] = fmaf(c[r], wz[b], fmaf(q[r], wy[b], fmaf(a, wx[b], Ev[b])));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float F = fmaf(q[r], q[r], s[r]);
                    m[b] = __builtin_amdgcn_alignbit(m[b], __float_as_uint(F), 31);   // (m << 1) | sign(F)
                }
            }
            if ((base & 127) == 112) acc ^= m[0] ^ m[1] ^ m[2] ^ m[3];
        }
        bq[0] += 1e-6f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class L>
double timeit(L launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3.0;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<float4> filt(NSPH);
    std::vector<float> soa(4 * NSPH), negT(NSPH);
    for (int i = 0; i < NSPH; ++i) {
        filt[i] = make_float4(0.01f * i, 0.2f, -0.01f * i, 100.f + i);
        soa[i] = 0.01f * i; soa[NSPH + i] = 0.2f; soa[2 * NSPH + i] = -0.01f * i; soa[3 * NSPH + i] = 1.f; negT[i] = -100.f - i;
    }
    float4 *d_f; float *d_s, *d_t; unsigned *d_o;
    hipMalloc(&d_f, sizeof(float4) * NSPH); hipMalloc(&d_s, sizeof(float) * 4 * NSPH); hipMalloc(&d_t, sizeof(float) * NSPH);
    hipMalloc(&d_o, sizeof(unsigned) * 256 * cus * 8);
    hipMemcpy(d_f, filt.data(), sizeof(float4) * NSPH, hipMemcpyHostToDevice);
    hipMemcpy(d_s, soa.data(), sizeof(float) * 4 * NSPH, hipMemcpyHostToDevice);
    hipMemcpy(d_t, negT.data(), sizeof(float) * NSPH, hipMemcpyHostToDevice);
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int blocks = cus * wps;
        const double tests_per_simd = (double)wps * REPS * NSPH;   // wave-tests (64 pairs each) per SIMD
        double t;
        t = timeit([&] { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d_f, d_o, 0.1f); });
        printf("waves/SIMD %d  VALU   %8.3f ms  %6.2f cyc@2.4GHz per 64 tests\n", wps, t, t * 1e-3 * 2.4e9 / tests_per_simd);
        t = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, d_s, d_t, d_o, 0.1f); });
        printf("waves/SIMD %d  MFMA8  %8.3f ms  %6.2f cyc@2.4GHz per 64 tests\n", wps, t, t * 1e-3 * 2.4e9 / tests_per_simd);
        t = timeit([&] { hipLaunchKernelGGL(k_mfma<6>, dim3(blocks), dim3(256), 0, 0, d_s, d_t, d_o, 0.1f); });
        printf("waves/SIMD %d  MFMA6  %8.3f ms  %6.2f cyc@2.4GHz per 64 tests\n", wps, t, t * 1e-3 * 2.4e9 / tests_per_simd);
        t = timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, d_s, d_t, d_o, 0.1f); });
        printf("waves/SIMD %d  MFMA4  %8.3f ms  %6.2f cyc@2.4GHz per 64 tests\n", wps, t, t * 1e-3 * 2.4e9 / tests_per_simd);
    }
    return 0;
}
