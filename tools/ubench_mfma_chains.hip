// How many independent accumulator chains does v_mfma_f32_32x32x16_f16 need to run back to back on gfx950?
// Loop body: NCH MFMAs, chain c depends on its own previous result only; 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v acc[NCH];
    for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    }
    float r = 0;
    for (int c = 0; c < NCH; ++c) for (int i = 0; i < 16; ++i) r += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NCH>
void run(float* out, int cus) {
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<NCH>, dim3(cus * wps), dim3(256), 0, 0, out, 100);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<NCH>, dim3(cus * wps), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns_per_mfma = ms * 1e6 / iters / wps / NCH;   // SIMD time per MFMA
        printf("chains=%d waves/SIMD=%d: %.2f ns of SIMD time per MFMA  (%.0f TFLOP/s whole chip)\n", NCH, wps, ns_per_mfma,
               1024.0 * 32768.0 / ns_per_mfma / 1e3);
    }
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float* out; (void)hipMalloc(&out, 1 << 24);
    run<1>(out, p.multiProcessorCount);
    run<2>(out, p.multiProcessorCount);
    run<4>(out, p.multiProcessorCount);
    run<8>(out, p.multiProcessorCount);
    return 0;
}
