// Which CUs does a stream created with hipExtStreamCreateWithCUMask run its workgroups on (gfx950, 8 XCDs x 32 CUs)?  And do two
// streams with complementary masks run side by side?   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_cumask tools/ubench_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <map>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup resident for a while so that the grid spreads over every CU the stream may use
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

__global__ void busy_kernel(unsigned long long* out, long long cycles) {
    long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned long long)clock64();
}

static int census(hipStream_t s, const char* tag, int blocks) {
    unsigned* d = nullptr;
    CHK(hipMalloc(&d, blocks * 8));
    CHK(hipMemsetAsync(d, 0xff, blocks * 8, s));
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(1024), 65536, s, d, 200000);
    CHK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * blocks);
    CHK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::map<unsigned, int>> per;   // xcc -> (se, cu) -> count
    for (int b = 0; b < blocks; ++b) {
        const unsigned xcc = h[2 * b] & 15, hw = h[2 * b + 1];
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // gfx9 HW_ID: [11:8] CU_ID, [12] SH_ID, [15:13] SE_ID
        per[xcc][(se << 8) | (sh << 4) | cu] += 1;
    }
    std::printf("%s: %d blocks ->", tag, blocks);
    int total = 0;
    for (auto& x : per) { std::printf(" xcc%u:%zu CUs", x.first, x.second.size()); total += (int)x.second.size(); }
    std::printf(" | %d distinct (xcc, se, sh, cu)\n", total);
    (void)hipFree(d);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    std::printf("%s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    hipStream_t s0;
    CHK(hipStreamCreate(&s0));
    if (census(s0, "no mask", 2048)) return 1;
    const unsigned full[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
    struct { const char* tag; unsigned m[8]; } masks[] = {
        {"bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bit k*8 (every 8th)", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
        {"all but bits 0..15", {0xffff0000u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"words 0..3", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
    };
    for (auto& mk : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mk.m);
        if (e != hipSuccess) { std::printf("%s: hipExtStreamCreateWithCUMask: %s\n", mk.tag, hipGetErrorString(e)); continue; }
        unsigned got[8] = {0};
        (void)hipExtStreamGetCUMask(s, 8, got);
        std::printf("  (mask read back: %08x %08x %08x %08x %08x %08x %08x %08x)\n", got[0], got[1], got[2], got[3], got[4], got[5], got[6], got[7]);
        if (census(s, mk.tag, 2048)) return 1;
        CHK(hipStreamDestroy(s));
    }
    // concurrency: stream A on "all but bits 0..15", stream B on bits 0..15: a 20 ms busy kernel on each, one workgroup per CU of its mask
    hipStream_t sa, sb;
    const unsigned ma[8] = {0xffff0000u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}, mb[8] = {0x0000ffffu, 0, 0, 0, 0, 0, 0, 0};
    CHK(hipExtStreamCreateWithCUMask(&sa, 8, ma));
    CHK(hipExtStreamCreateWithCUMask(&sb, 8, mb));
    unsigned long long* d = nullptr;
    CHK(hipMalloc(&d, 64));
    hipEvent_t e0, e1, e2, e3;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&e2)); CHK(hipEventCreate(&e3));
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(busy_kernel, dim3(240), dim3(1024), 120000, sa, d, 40000000LL);
        CHK(hipEventRecord(e1, sa));
        CHK(hipEventRecord(e2, sb));
        hipLaunchKernelGGL(busy_kernel, dim3(16 * 8), dim3(256), 0, sb, d + 1, 40000000LL);
        CHK(hipEventRecord(e3, sb));
        CHK(hipDeviceSynchronize());
        float a = 0, b = 0, span = 0;
        CHK(hipEventElapsedTime(&a, e0, e1)); CHK(hipEventElapsedTime(&b, e2, e3)); CHK(hipEventElapsedTime(&span, e0, e3));
        std::printf("two masked streams: A (240 big workgroups) %.2f ms, B (128 small ones) %.2f ms, A start -> B end %.2f ms (side by side if ~= max)\n", a, b, span);
    }
    (void)full;
    return 0;
}
