// ubench_upload.hip -- what a store upload can cost on this box: the ingredients of msfm_upload_image, one at a time.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_upload tools/ubench_upload.hip -lpthread && tools/ubench_upload
// Workload: 128 host buffers of 5038 x 128 floats (2.58 MB each, pageable, touched) = 330 MB -- the bench job's store.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHK(x)                                                                         \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void touch_kernel(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f;
}
__global__ void tiny_kernel(const float* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 2.f;
}

// `nthreads` workers copy [src, src + bytes) into dst in parallel (the caller is worker 0)
struct CopyPool {
    int n;
    std::vector<std::thread> th;
    std::atomic<int> gen{0}, done{0};
    std::atomic<bool> stop{false};
    const char* src = nullptr;
    char* dst = nullptr;
    size_t bytes = 0;
    explicit CopyPool(int nthreads) : n(nthreads) {
        for (int t = 1; t < n; ++t)
            th.emplace_back([this, t] {
                int seen = 0;
                for (;;) {
                    while (gen.load(std::memory_order_acquire) == seen && !stop.load()) { /* spin */ }
                    if (stop.load()) return;
                    seen = gen.load();
                    part(t);
                    done.fetch_add(1, std::memory_order_release);
                }
            });
    }
    void part(int t) {
        const size_t per = (bytes / n + 4095) & ~(size_t)4095, b = std::min(bytes, per * t), e = std::min(bytes, per * (t + 1));
        if (e > b) std::memcpy(dst + b, src + b, e - b);
    }
    void copy(void* d, const void* s, size_t nbytes) {
        src = (const char*)s, dst = (char*)d, bytes = nbytes;
        done.store(0);
        gen.fetch_add(1, std::memory_order_release);
        part(0);
        while (done.load(std::memory_order_acquire) < n - 1) { /* spin */ }
    }
    ~CopyPool() {
        stop.store(true);
        for (auto& t : th) t.join();
    }
};

int main() {
    const int N = 128, rows = 5038;
    const size_t bytes = (size_t)rows * 128 * 4;
    std::vector<float*> host(N);
    for (int i = 0; i < N; ++i) {
        host[i] = (float*)std::malloc(bytes);
        for (size_t k = 0; k < bytes / 4; k += 256) host[i][k] = (float)k;   // touched
        std::memset(host[i], i, bytes);
    }
    CHK(hipSetDevice(0));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* dev = nullptr;
    CHK(hipMalloc(&dev, bytes * N));
    hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, s, dev, bytes * N / 4);
    CHK(hipStreamSynchronize(s));
    std::printf("store: %d images x %zu bytes = %.1f MB\n", N, bytes, bytes * N / 1e6);

    // ---- (a) hipMalloc / hipFree of the sizes build_image asks for (per image: 3 x 2.6 MB, 1.4 MB, 0.9 MB, 6 small)
    {
        const size_t sz[11] = {bytes, bytes, bytes, (size_t)rows * 272, (size_t)rows * 176, (size_t)rows * 4, (size_t)rows * 4, (size_t)rows * 4, (size_t)rows * 4, (size_t)rows * 8, 4096};
        void* p[11 * 128];
        double t0 = now_ms();
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 11; ++k) CHK(hipMalloc(&p[i * 11 + k], sz[k]));
        double t1 = now_ms();
        for (int i = 0; i < N * 11; ++i) CHK(hipFree(p[i]));
        double t2 = now_ms();
        std::printf("(a) %d hipMalloc: %.2f ms (%.1f us each); hipFree: %.2f ms (%.1f us each)\n", N * 11, t1 - t0, (t1 - t0) * 1e3 / (N * 11), t2 - t1,
                    (t2 - t1) * 1e3 / (N * 11));
        t0 = now_ms();
        void* big = nullptr;
        CHK(hipMalloc(&big, (size_t)1 << 30));
        t1 = now_ms();
        hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, s, (float*)big, ((size_t)1 << 30) / 4);
        CHK(hipStreamSynchronize(s));
        t2 = now_ms();
        hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, s, (float*)big, ((size_t)1 << 30) / 4);
        CHK(hipStreamSynchronize(s));
        double t3 = now_ms();
        std::printf("    one 1 GiB hipMalloc: %.2f ms; first touch (fill kernel) %.2f ms, second %.2f ms\n", t1 - t0, t2 - t1, t3 - t2);
        CHK(hipFree(big));
    }
    // ---- (b) pageable hipMemcpyAsync, sync per image / sync at the end
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now_ms();
        for (int i = 0; i < N; ++i) {
            CHK(hipMemcpyAsync((char*)dev + bytes * i, host[i], bytes, hipMemcpyHostToDevice, s));
            CHK(hipStreamSynchronize(s));
        }
        double t1 = now_ms();
        for (int i = 0; i < N; ++i) CHK(hipMemcpyAsync((char*)dev + bytes * i, host[i], bytes, hipMemcpyHostToDevice, s));
        CHK(hipStreamSynchronize(s));
        double t2 = now_ms();
        std::printf("(b) pageable hipMemcpyAsync: sync per image %.2f ms (%.1f GB/s), one sync at the end %.2f ms (%.1f GB/s)\n", t1 - t0,
                    bytes * N / (t1 - t0) / 1e6, t2 - t1, bytes * N / (t2 - t1) / 1e6);
    }
    // ---- (c) memcpy into a page-locked ring + hipMemcpyAsync, 1 / 2 / 4 / 8 copy threads
    {
        const int R = 8;
        char* ring = nullptr;
        CHK(hipHostMalloc((void**)&ring, bytes * R, hipHostMallocDefault));
        std::memset(ring, 0, bytes * R);
        hipEvent_t ev[R];
        for (int r = 0; r < R; ++r) CHK(hipEventCreateWithFlags(&ev[r], hipEventDisableTiming));
        for (int nt : {1, 2, 4, 8}) {
            CopyPool pool(nt);
            for (int rep = 0; rep < 2; ++rep) {
                double t0 = now_ms(), tcopy = 0;
                for (int i = 0; i < N; ++i) {
                    const int r = i % R;
                    if (i >= R) CHK(hipEventSynchronize(ev[r]));
                    const double c0 = now_ms();
                    pool.copy(ring + bytes * r, host[i], bytes);
                    tcopy += now_ms() - c0;
                    CHK(hipMemcpyAsync((char*)dev + bytes * i, ring + bytes * r, bytes, hipMemcpyHostToDevice, s));
                    CHK(hipEventRecord(ev[r], s));
                }
                CHK(hipStreamSynchronize(s));
                double t1 = now_ms();
                if (rep) std::printf("(c) %d copy thread(s) -> page-locked ring -> async H2D: %.2f ms (%.1f GB/s); host memcpy alone %.2f ms (%.1f GB/s)\n", nt, t1 - t0,
                                     bytes * N / (t1 - t0) / 1e6, tcopy, bytes * N / tcopy / 1e6);
            }
        }
        // the DMA alone, from page-locked memory
        double t0 = now_ms();
        for (int i = 0; i < N; ++i) CHK(hipMemcpyAsync((char*)dev + bytes * i, ring + bytes * (i % R), bytes, hipMemcpyHostToDevice, s));
        CHK(hipStreamSynchronize(s));
        double t1 = now_ms();
        std::printf("    page-locked -> device alone: %.2f ms (%.1f GB/s)\n", t1 - t0, bytes * N / (t1 - t0) / 1e6);
        CHK(hipHostFree(ring));
    }
    // ---- (d) hipHostRegister the caller's buffer, async copy, unregister
    {
        double t0 = now_ms(), treg = 0;
        for (int i = 0; i < N; ++i) {
            const double r0 = now_ms();
            CHK(hipHostRegister(host[i], bytes, hipHostRegisterDefault));
            treg += now_ms() - r0;
            CHK(hipMemcpyAsync((char*)dev + bytes * i, host[i], bytes, hipMemcpyHostToDevice, s));
            CHK(hipStreamSynchronize(s));
            CHK(hipHostUnregister(host[i]));
        }
        double t1 = now_ms();
        std::printf("(d) hipHostRegister + async H2D + sync + unregister per image: %.2f ms (%.1f GB/s); register alone %.2f ms\n", t1 - t0,
                    bytes * N / (t1 - t0) / 1e6, treg);
    }
    // ---- (e) launch + sync latency: one tiny kernel + hipStreamSynchronize, and a 32-byte D2H read-back + sync
    {
        float* h = nullptr;
        CHK(hipHostMalloc((void**)&h, 4096, hipHostMallocDefault));
        double t0 = now_ms();
        for (int i = 0; i < 512; ++i) {
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, dev, dev + 64, 64);
            CHK(hipStreamSynchronize(s));
        }
        double t1 = now_ms();
        unsigned mx[8];
        for (int i = 0; i < 512; ++i) {
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, dev, dev + 64, 64);
            CHK(hipMemcpyAsync(mx, dev, 32, hipMemcpyDeviceToHost, s));
            CHK(hipStreamSynchronize(s));
        }
        double t2 = now_ms();
        for (int i = 0; i < 512; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, dev, dev + 64, 64);
        CHK(hipStreamSynchronize(s));
        double t3 = now_ms();
        std::printf("(e) kernel + sync: %.1f us; kernel + 32-byte pageable D2H + sync: %.1f us; kernel launch back to back: %.1f us\n", (t1 - t0) * 1e3 / 512,
                    (t2 - t1) * 1e3 / 512, (t3 - t2) * 1e3 / 512);
        CHK(hipHostFree(h));
    }
    return 0;
}
