// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal INPUTS (gradual underflow) or flush them to zero?
// The prefilter's error bound (msfm_prefilter.hip.h) assumes |fl16(a) - a| <= max(2^-11 |a|, 2^-25), i.e. no flush.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float av, float bv) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    if ((lane >> 5) == 0) { a[0] = (_Float16)av; b[0] = (_Float16)bv; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) { out[0] = acc[0]; out[1] = (float)(_Float16)av; out[2] = (float)(_Float16)bv; }
}
int main() {
    float* d; hipMalloc(&d, 64); float h[3];
    const float cases[][2] = {{3.0e-5f, 1024.f}, {5.96e-8f, 32768.f}, {1.0e-6f, 1.0e-6f}, {6.2e-5f, 1.f}, {3.0e-5f, 3.0e-5f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g (fp16 %g) * b=%g (fp16 %g): mfma %g  exact-product-of-fp16 %g  %s\n", c[0], h[1], c[1], h[2], h[0], (double)h[1] * h[2],
               h[0] == (float)((double)h[1] * h[2]) ? "KEPT" : "differs");
    }
    return 0;
}
