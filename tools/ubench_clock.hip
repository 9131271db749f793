// ubench_clock.hip -- what does s_memtime count, and what clock does the part sustain under an MFMA-only / a VALU-only
// loop?  Per kernel: wall time (HIP events), s_memtime ticks per iteration (one wave per SIMD reports) -> ticks per
// instruction and ticks per microsecond.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma(float* out, unsigned long long* ticks, int iters, unsigned seed) {
    h8 a, b;
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; a[i] = (_Float16)((x >> 8) * (1.f / 16777216.f)); x = x * 1664525u + 1013904223u; b[i] = (_Float16)((x >> 8) * (1.f / 16777216.f)); }
    f16v c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = c1[r] = c2[r] = c3[r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// other matrix instructions, same 4-chain loop: ticks per instruction
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void k_mfma_kind(float* out, unsigned long long* ticks, int iters, unsigned seed) {
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    h4 a4, b4;
    i4v ai, bi;
    for (int i = 0; i < 4; ++i) {
        x = x * 1664525u + 1013904223u; a4[i] = (_Float16)((x >> 8) * (1.f / 16777216.f)); ai[i] = (int)x;
        x = x * 1664525u + 1013904223u; b4[i] = (_Float16)((x >> 8) * (1.f / 16777216.f)); bi[i] = (int)x;
    }
    f16v c[4];
    i16v d[4];
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 16; ++r) { c[k][r] = 0.f; d[k][r] = 0; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (KIND == 0) c[k] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c[k], 0, 0, 0);
            if (KIND == 1) d[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d[k], 0, 0, 0);
            if (KIND == 2) d[k] = __builtin_amdgcn_mfma_i32_32x32x16_i8(((long*)&ai)[0], ((long*)&bi)[0], d[k], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) r += c[k][i] + (float)d[k][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_valu(float* out, unsigned long long* ticks, int iters, unsigned seed) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i + seed;
    const float s = 1.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(s));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float* out; (void)hipMalloc(&out, 1 << 24);
    unsigned long long* ticks; (void)hipMalloc(&ticks, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    if (argc >= 4 && std::strcmp(argv[1], "sustain") == 0) {
        // `ubench_clock sustain <mfma|valu> <seconds>`: back-to-back launches for that long (tools/gpu_power_trace.sh samples
        // power and clocks meanwhile); prints the rate of every ~second
        const bool mfma = std::strcmp(argv[2], "mfma") == 0;
        const double seconds = std::atof(argv[3]);
        const int iters = mfma ? 2000000 : 4000000;   // ~150 ms per launch
        double elapsed = 0;
        while (elapsed < seconds * 1e3) {
            (void)hipEventRecord(e0);
            for (int k = 0; k < 6; ++k) {
                if (mfma) hipLaunchKernelGGL(k_mfma, dim3(p.multiProcessorCount), dim3(256), 0, 0, out, ticks, iters, 7u);
                else hipLaunchKernelGGL(k_valu, dim3(p.multiProcessorCount), dim3(256), 0, 0, out, ticks, iters, 7u);
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            elapsed += ms;
            const double n_inst = (double)iters * (mfma ? 4 : 8);
            printf("sustained %s loop at %.1f s: %.2f ticks per instruction, %.1f ticks per microsecond (last launch), %.2f ns per instruction per wave\n",
                   mfma ? "MFMA-only" : "VALU-only", elapsed * 1e-3, t / n_inst, t / (ms / 6 * 1e3), ms / 6 * 1e6 / n_inst);
        }
        return 0;
    }
    for (int kind = 0; kind < 2; ++kind)
        for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
            const int iters = kind == 0 ? 200000 : 400000;
            const int grid = p.multiProcessorCount * blocks_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, ticks, iters, 7u);
                else hipLaunchKernelGGL(k_valu, dim3(grid), dim3(256), 0, 0, out, ticks, iters, 7u);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            const double n_inst = (double)iters * (kind == 0 ? 4 : 8);
            printf("%s waves/SIMD=%d: %.3f ms wall, %.2f ticks per instruction, %.1f ticks per microsecond, %.2f ns per instruction per wave\n",
                   kind == 0 ? "MFMA-only (4 chains, random fp16)" : "VALU-only (8 chains v_fma)    ", blocks_per_cu, ms, t / n_inst, t / (ms * 1e3),
                   ms * 1e6 / n_inst);
        }
    const char* names[3] = {"v_mfma_f32_32x32x8_f16 (legacy K=8) ", "v_mfma_i32_32x32x32_i8             ", "v_mfma_i32_32x32x16_i8 (legacy)    "};
    for (int kind = 0; kind < 3; ++kind) {
        const int iters = 100000;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k_mfma_kind<0>, dim3(p.multiProcessorCount), dim3(256), 0, 0, out, ticks, iters, 7u);
            if (kind == 1) hipLaunchKernelGGL(k_mfma_kind<1>, dim3(p.multiProcessorCount), dim3(256), 0, 0, out, ticks, iters, 7u);
            if (kind == 2) hipLaunchKernelGGL(k_mfma_kind<2>, dim3(p.multiProcessorCount), dim3(256), 0, 0, out, ticks, iters, 7u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("%s waves/SIMD=1: %.3f ms wall, %.2f ticks per instruction, %.1f ticks per microsecond\n", names[kind], ms, t / (iters * 4.0), t / (ms * 1e3));
    }
    return 0;
}
