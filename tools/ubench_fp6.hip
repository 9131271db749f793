// ubench_fp6.hip -- VERDICT r04 item 5: the sustained rate of the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 e2m3 / fp4 operands,
// K = 64 per instruction: two instructions per 128 dimensions where v_mfma_i32_32x32x32_i8 needs four) alone and beside the v_max3
// epilogue waves of a first sweep -- same harness as ubench_coissue.hip (per SIMD NM matrix waves over 4 independent accumulator chains,
// NV waves over independent v_max3 chains until the matrix waves are done), the int8 instruction measured next to it on the same box.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_fp6 tools/ubench_fp6.hip && tools/ubench_fp6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Rec { unsigned long long ticks, insts; };

template <int MODE>   // matrix instruction: 0 v_mfma_i32_32x32x32_i8, 2 v_mfma_scale_f32_32x32x64_f8f6f4 on fp6 (e2m3), 4 the same on fp4
__global__ __launch_bounds__(1024) void k_mix(float* out, Rec* rec, int nm, int nv, int iters_m, unsigned seed) {
    __shared__ int done;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool matrix = (wave >> 2) < nm;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float r = 0.f;
    unsigned long long t0 = 0, t1 = 0, n = 0;
    if (matrix) {
        if (MODE == 0) {
            i4v a, b;
            for (int i = 0; i < 4; ++i) { x = x * 1664525u + 1013904223u; a[i] = (int)x; x = x * 1664525u + 1013904223u; b[i] = (int)x; }
            i16v d[4];
            for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) d[k][q] = 0;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters_m; ++it) {
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d[k], 0, 0, 0);
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) r += (float)d[k][q];
        } else {
            // random operand bits (every fp6 / fp4 pattern is a finite number); fp6: 32 values x 6 bits = 6 registers, fp4: 4 of the 8
            i8v a, b;
            for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; a[i] = (int)x; x = x * 1664525u + 1013904223u; b[i] = (int)x; }
            f16v d[4];
            for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) d[k][q] = 0.f;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < iters_m; ++it) {
#pragma unroll
                for (int k = 0; k < 4; ++k)   // (cbsz = blgp = format of A / B; block scales 2^(120 - 127): the accumulators stay finite)
                    d[k] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d[k], MODE, MODE, 0, 120, 0, 120);
            }
            t1 = __builtin_amdgcn_s_memtime();
            for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) r += d[k][q];
        }
        n = 4ull * iters_m;
        if (lane == 0) atomicAdd(&done, 1);
    } else {
        int v[8];
        for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; v[i] = (int)x; }
        const int s0 = (int)(x >> 3), s1 = (int)(x >> 7);
        t0 = __builtin_amdgcn_s_memtime();
        const int target = nm * 4;
        while (true) {
            for (int it = 0; it < 64; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s0), "v"(s1));
            }
            n += 64 * 8;
            if (nm == 0) { if (n >= 8ull * 64 * 4000) break; }
            else if (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) r += (float)v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0 && blockIdx.x == 0) rec[wave] = Rec{t1 - t0, n};
}

template <int MODE>
static void run(const char* name, double ops_per_inst, int cus, float* out, Rec* rec, hipEvent_t e0, hipEvent_t e1) {
    const int cfg[][2] = {{1, 0}, {2, 0}, {1, 1}, {1, 2}, {2, 2}, {1, 3}};
    for (auto& c : cfg) {
        const int nm = c[0], nv = c[1], waves = (nm + nv) * 4, iters = 60000;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(rec, 0, 16 * sizeof(Rec));
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_mix<MODE>, dim3(cus), dim3(waves * 64), 0, 0, out, rec, nm, nv, iters, 7u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        Rec h[16]; (void)hipMemcpy(h, rec, sizeof(h), hipMemcpyDeviceToHost);
        double tm = 0, tv = 0; int cm = 0, cv = 0;
        for (int w = 0; w < waves; ++w) {
            if (!h[w].insts) continue;
            const double tpi = (double)h[w].ticks / (double)h[w].insts;
            if ((w >> 2) < nm) { tm += tpi; ++cm; } else { tv += tpi; ++cv; }
        }
        const double insts = 4.0 * iters * nm * 4 * cus;   // matrix instructions of the launch
        printf("%s: per SIMD %d matrix + %d VALU wave(s): %.3f ms wall = %.2f PetaOP/s, %.1f ns per matrix instruction and SIMD", name, nm, nv, ms,
               insts * ops_per_inst / (ms * 1e-3) / 1e15, ms * 1e6 / (4.0 * iters * nm));
        if (cm) printf(" | matrix: %.2f ticks per instruction and wave", tm / cm);
        if (cv) printf(" | v_max3_i32: one per %.2f ticks per SIMD", tv / cv / nv);
        printf("\n");
    }
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float* out; (void)hipMalloc(&out, 1 << 24);
    Rec* rec; (void)hipMalloc(&rec, 16 * sizeof(Rec));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run<0>("i8  32x32x32", 2.0 * 32 * 32 * 32, p.multiProcessorCount, out, rec, e0, e1);
    run<2>("fp6 32x32x64", 2.0 * 32 * 32 * 64, p.multiProcessorCount, out, rec, e0, e1);
    run<4>("fp4 32x32x64", 2.0 * 32 * 32 * 64, p.multiProcessorCount, out, rec, e0, e1);
    run<0>("i8  32x32x32 (again)", 2.0 * 32 * 32 * 32, p.multiProcessorCount, out, rec, e0, e1);
    return 0;
}
