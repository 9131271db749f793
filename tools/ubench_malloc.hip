// ubench_malloc.hip -- what device allocations cost on this box, by size (fresh process; every size allocated, touched, freed)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f; }
int main(int argc, char** argv) {
    (void)hipSetDevice(0);
    hipStream_t s; (void)hipStreamCreate(&s);
    void* warm; (void)hipMalloc(&warm, 4096); touch<<<1, 64, 0, s>>>((float*)warm, 16); (void)hipStreamSynchronize(s);
    const size_t MB = 1 << 20;
    struct Case { size_t bytes; int count; } cases[] = {{5 * MB, 128}, {1 * MB, 512}, {16 * MB, 40}, {64 * MB, 10}, {128 * MB, 5}, {640 * MB, 1}, {1024 * MB, 1}, {4096 * MB, 1}, {16384 * MB, 1}, {5 * MB, 128}, {640 * MB, 1}};
    for (const Case& c : cases) {
        std::vector<void*> p(c.count);
        double t0 = now_ms();
        for (int i = 0; i < c.count; ++i) if (hipMalloc(&p[i], c.bytes) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
        double t1 = now_ms();
        for (int i = 0; i < c.count; ++i) touch<<<1024, 256, 0, s>>>((float*)p[i], c.bytes / 4);
        (void)hipStreamSynchronize(s);
        double t2 = now_ms();
        for (int i = 0; i < c.count; ++i) touch<<<1024, 256, 0, s>>>((float*)p[i], c.bytes / 4);
        (void)hipStreamSynchronize(s);
        double t3 = now_ms();
        for (int i = 0; i < c.count; ++i) (void)hipFree(p[i]);
        double t4 = now_ms();
        std::printf("%4d x %6zu MiB: hipMalloc %8.3f ms (%6.2f us per MiB) | first touch %7.3f ms, second %7.3f ms | hipFree %8.3f ms\n", c.count, c.bytes / MB, t1 - t0,
                    (t1 - t0) * 1e3 / (c.count * (c.bytes / (double)MB)), t2 - t1, t3 - t2, t4 - t3);
    }
    // the same through the stream-ordered pool
    hipMemPool_t pool; (void)hipDeviceGetDefaultMemPool(&pool, 0);
    unsigned long long thr = ~0ull; (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    for (int rep = 0; rep < 2; ++rep) {
        void* p[128];
        double t0 = now_ms();
        for (int i = 0; i < 128; ++i) (void)hipMallocAsync(&p[i], 5 * MB, s);
        (void)hipStreamSynchronize(s);
        double t1 = now_ms();
        for (int i = 0; i < 128; ++i) (void)hipFreeAsync(p[i], s);
        (void)hipStreamSynchronize(s);
        double t2 = now_ms();
        std::printf("128 x 5 MiB hipMallocAsync (rep %d): %.3f ms; hipFreeAsync %.3f ms\n", rep, t1 - t0, t2 - t1);
    }
    // page-locked host memory
    for (size_t mb : {8, 32, 128}) {
        void* h; double t0 = now_ms(); (void)hipHostMalloc(&h, mb * MB, hipHostMallocDefault); double t1 = now_ms(); (void)hipHostFree(h); double t2 = now_ms();
        std::printf("hipHostMalloc %zu MiB: %.3f ms, free %.3f ms\n", mb, t1 - t0, t2 - t1);
    }
    return 0;
}
