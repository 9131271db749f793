// ubench_coissue.hip -- do VALU instructions of ONE wave issue under the matrix instructions of ANOTHER wave of the same SIMD?
// A workgroup of (NM + NV) x 4 waves: waves are dealt round-robin to the 4 SIMDs of a CU, so every SIMD hosts NM waves that
// loop over independent v_mfma_i32_32x32x32_i8 (4 accumulator chains) and NV waves that loop over independent v_max3_i32
// (8 chains) until the matrix waves are done.  Reported per role: s_memtime ticks per instruction and wave (alone: 34.0 and 8.0,
// profiles/r02_ubench_clock.txt), i.e. how much each side is slowed by the other.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

struct Rec { unsigned long long ticks, insts; };

template <int MODE>   // 0: v_max3_i32 chains, 1: v_max3_i32 chains reading a matrix wave's freshly written register pattern (ds traffic none)
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
        n = 4ull * iters_m;
        for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) r += (float)d[k][q];
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

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float* out; (void)hipMalloc(&out, 1 << 24);
    Rec* rec; (void)hipMalloc(&rec, 16 * sizeof(Rec));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int cfg[][2] = {{1, 0}, {2, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {1, 3}, {2, 1}, {2, 2}};
    for (auto& c : cfg) {
        const int nm = c[0], nv = c[1], waves = (nm + nv) * 4;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(rec, 0, 16 * sizeof(Rec));
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_mix<0>, dim3(p.multiProcessorCount), dim3(waves * 64), 0, 0, out, rec, nm, nv, 60000, 7u);
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
        printf("per SIMD %d matrix wave(s) + %d VALU wave(s): %.3f ms wall", nm, nv, ms);
        if (cm) printf(" | v_mfma_i32_32x32x32_i8: %.2f ticks per instruction and wave (matrix pipe busy %.0f %% at 32 cycles per instruction)", tm / cm, 100.0 * 32.0 * nm / (tm / cm));
        if (cv) printf(" | v_max3_i32: %.2f ticks per instruction and wave (= one per %.2f ticks per SIMD)", tv / cv, tv / cv / nv);
        printf("\n");
    }
    return 0;
}
