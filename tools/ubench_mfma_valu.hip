// ubench_mfma_valu.hip -- can v_mfma_f32_32x32x16_f16 and fp32 VALU work overlap inside one wave on gfx950?
// Loop body: 1 MFMA (32 cycles of matrix pipe) + N independent VALU ops; 1, 2 or 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    float s = 1.0001f;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i % 16]) : "v"(s));
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[(i + 8) % 16]) : "v"(s));
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += v[i] + acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NV>
void run(float* out, int cus) {
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<NV>, dim3(cus * wps), dim3(256), 0, 0, out, 100);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<NV>, dim3(cus * wps), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wps waves x iters x (2 MFMA + 2*NV VALU)
        const double ns_per_iter = ms * 1e6 / iters / wps;
        printf("NV=%2d (2 MFMA + %2d VALU per iter) waves/SIMD=%d: %.1f ns per wave-iter = %.0f cycles@2.4GHz  (MFMA alone needs 64)\n",
               NV, 2 * NV, wps, ns_per_iter, ns_per_iter * 2.4);
    }
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float* out; (void)hipMalloc(&out, 1 << 24);
    run<0>(out, p.multiProcessorCount);
    run<4>(out, p.multiProcessorCount);
    run<8>(out, p.multiProcessorCount);
    run<12>(out, p.multiProcessorCount);
    run<16>(out, p.multiProcessorCount);
    run<24>(out, p.multiProcessorCount);
    return 0;
}
