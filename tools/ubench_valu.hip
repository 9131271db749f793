// ubench_valu.hip -- gfx950 micro-benchmarks behind the design choices in DESIGN.md:
//  (1) sustained rate of the exact-L2 inner operation (sub, mul, add; not fused) as scalar
//      VALU ops vs packed v_pk_*_f32 ops, at 1 and 2 waves per SIMD;
//  (2) exhaustive check that device sqrtf is correctly rounded (the top-2 epilogue relies on it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
#pragma clang fp contract(off)

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_scalar(float* out, int iters, float a0, float b0) {
    float p[NACC], a[NACC];
    for (int i = 0; i < NACC; ++i) { p[i] = 0.f; a[i] = a0 + i + threadIdx.x; }
    float b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            float t, m;
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(b));
            asm volatile("v_mul_f32 %0, %1, %1" : "=v"(m) : "v"(t));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m));
        }
    }
    float s = 0; for (int i = 0; i < NACC; ++i) s += p[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));
template <int NACC>
__global__ __launch_bounds__(256) void k_packed(float* out, int iters, float a0, float b0) {
    float2v p[NACC], a[NACC];
    for (int i = 0; i < NACC; ++i) { p[i] = (float2v){0.f, 0.f}; a[i] = (float2v){a0 + i + threadIdx.x, a0 - i}; }
    float2v b = (float2v){b0, b0 + 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            float2v t, m;
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a[i]), "v"(b));
            asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(m) : "v"(t));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m));
        }
    }
    float s = 0; for (int i = 0; i < NACC; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// fused variant: t = a-b; p = fma(t,t,p)
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a0, float b0) {
    float p[NACC], a[NACC];
    for (int i = 0; i < NACC; ++i) { p[i] = 0.f; a[i] = a0 + i + threadIdx.x; }
    float b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            float t;
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(b));
            asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(p[i]) : "v"(t));
        }
    }
    float s = 0; for (int i = 0; i < NACC; ++i) s += p[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_sqrt_check(unsigned long long* bad, unsigned* first_bad) {
    // all positive finite normal+subnormal floats: bits 1 .. 0x7f7fffff
    const unsigned long long total = 0x7f800000ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float y = sqrtf(x);
        // y is the correctly rounded sqrt iff (y - ulp/2)^2 <= x <= (y + ulp/2)^2 evaluated exactly;
        // products of 25-bit values are exact in double.
        const double yd = (double)y;
        const float yu = __uint_as_float(__float_as_uint(y) + 1), yl = (__float_as_uint(y) > 0) ? __uint_as_float(__float_as_uint(y) - 1) : 0.f;
        const double hi = 0.5 * (yd + (double)yu), lo = 0.5 * (yd + (double)yl);
        const double xd = (double)x;
        bool ok = (lo * lo <= xd) && (xd <= hi * hi);
        if (x == 0.f) ok = (y == 0.f);
        if (!ok) { atomicAdd(bad, 1ull); atomicMin(first_bad, (unsigned)i); }
    }
}

template <typename F>
static double time_kernel(F launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s %s CUs=%d clock=%d MHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    float* out; CHK(hipMalloc(&out, 256 * 8192 * sizeof(float)));
    const int iters = 4096;
    const int cus = prop.multiProcessorCount;
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD
        const int blocks = cus * wps;         // 256 threads = 4 waves = 1 per SIMD per block
        {
            double ms = time_kernel([&] { hipLaunchKernelGGL(k_scalar<32>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f, 0.25f); }, 5);
            double ops = (double)blocks * 256 * iters * 32 * 3;
            printf("scalar sub/mul/add  waves/SIMD=%d: %.3f ms  %.2f T lane-ops/s\n", wps, ms, ops / ms / 1e9);
        }
        {
            double ms = time_kernel([&] { hipLaunchKernelGGL(k_packed<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f, 0.25f); }, 5);
            double ops = (double)blocks * 256 * iters * 16 * 3 * 2;
            printf("packed pk sub/mul/add waves/SIMD=%d: %.3f ms  %.2f T lane-ops/s\n", wps, ms, ops / ms / 1e9);
        }
        {
            double ms = time_kernel([&] { hipLaunchKernelGGL(k_fma<32>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f, 0.25f); }, 5);
            double ops = (double)blocks * 256 * iters * 32 * 2;
            printf("scalar sub+fmac      waves/SIMD=%d: %.3f ms  %.2f T instr-lanes/s\n", wps, ms, ops / ms / 1e9);
        }
    }
    unsigned long long* bad; unsigned* first; CHK(hipMalloc(&bad, 8)); CHK(hipMalloc(&first, 4));
    CHK(hipMemset(bad, 0, 8)); CHK(hipMemset(first, 0xff, 4));
    hipLaunchKernelGGL(k_sqrt_check, dim3(cus * 8), dim3(256), 0, 0, bad, first);
    CHK(hipDeviceSynchronize());
    unsigned long long hbad; unsigned hfirst;
    CHK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hfirst, first, 4, hipMemcpyDeviceToHost));
    printf("sqrtf exhaustive: %llu incorrectly rounded of %llu (first bad bits 0x%08x)\n", hbad, 0x7f800000ull, hfirst);
    // spot-check against the host libm as well
    std::vector<float> xs(1 << 20), ys(1 << 20);
    return hbad == 0 ? 0 : 2;
}
