// msfm_match.hip -- host side of the C ABI declared in include/msfm_match.h.
//
// Owns the device-resident descriptor store, schedules image pairs onto the gfx950 kernels in
// msfm_kernels.hip.h and returns match lists.  Mirrors what FeatureMatcher::MatchImagePairs
// (src/Feature/FeatureMatching.cpp:10-73 of the reference) does between its two
// Database::ReadDescriptors calls and FeatureUtils::FilterMatches, without the per-pair
// descriptor re-read.  No CPU fallback: every entry point that needs the GPU fails loudly
// when there is none.
#include "msfm_match.h"
#include "msfm_kernels.hip.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace msfm;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Image {
    int n = -1;  // -1: not uploaded
    int nblk = 0;
    float* panel = nullptr;
    float* raw = nullptr;
};

constexpr int kSlots = 2 * MSFM_MAX_IMAGES;  // ids >= MSFM_MAX_IMAGES: auxiliary (top-scale subsets)

}  // namespace

struct msfm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int order = MSFM_ORDER_SSE4X4;
    int cu_count = 0, clock_mhz = 0;
    char dev_name[256] = {0};
    std::vector<Image> images;
    std::string err;

    DevBuf d_pairs, d_items, d_stage;
    DevBuf d_rp_s0, d_rp_i0, d_rp_s1, d_cp_s0, d_cp_i0, d_cp_s1;
    DevBuf d_k_i0, d_k_d0, d_k_d1;
    DevBuf d_st_qt, d_st_d, d_counts, d_offsets, d_out_qt, d_out_d;
    DevBuf d_fix_count, d_fix_list;

    // results of the last msfm_match_pairs call
    bool have_results = false;
    std::vector<int64_t> res_offsets;
    std::vector<int32_t> res_qt;
    std::vector<float> res_dist;

    msfm_profile prof = {};
    std::vector<hipEvent_t> ev_pool;
};

namespace {

int fail(msfm_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(ctx, MSFM_E_DEVICE,                                                   \
                        std::string(#call) + ": " + hipGetErrorString(e__));                  \
    } while (0)

constexpr int kFixCap = 1 << 16;

struct Batch {
    std::vector<PairDesc> pairs;
    std::vector<WorkItem> items;
    long long rp_elems = 0, cp_elems = 0, kf_elems = 0, kr_elems = 0, out_elems = 0;
    int max_npad = 0;
    int64_t desc_pairs = 0;
    int64_t algo_bytes = 0;
};

// Split the pair list [begin, end) into work items.  Items of one pair are contiguous; the list
// is then interleaved over the 8 XCDs (workgroup b runs on XCD b % 8) so that the workgroups
// streaming the same B panels share one L2.
void build_items(Batch& b) {
    long long total_ablocks = 0;
    for (auto& pd : b.pairs)
        if (pd.valid) total_ablocks += pd.a_blocks;
    std::vector<WorkItem> lin;
    for (size_t p = 0; p < b.pairs.size(); ++p) {
        PairDesc& pd = b.pairs[p];
        if (!pd.valid) continue;
        for (int r = 0; r < pd.ranges; ++r) {
            const int t0 = (int)((long long)pd.b_tiles * r / pd.ranges);
            const int t1 = (int)((long long)pd.b_tiles * (r + 1) / pd.ranges);
            for (int ab = 0; ab < pd.a_blocks; ++ab) {
                WorkItem w = {};
                w.pair = (int)p;
                w.a_blk = ab;
                w.bt_begin = t0;
                w.bt_end = t1;
                w.range = r;
                lin.push_back(w);
            }
        }
    }
    const size_t n = lin.size();
    const size_t per = (n + 7) / 8;
    b.items.assign(per * 8, WorkItem{-1, 0, 0, 0, 0, {0, 0, 0}});
    for (size_t k = 0; k < n; ++k) {
        const size_t x = k / per, j = k % per;
        b.items[j * 8 + x] = lin[k];
    }
}

int fill_pair(msfm_ctx* ctx, int id1, int id2, PairDesc& pd) {
    if (id1 < 0 || id1 >= kSlots || id2 < 0 || id2 >= kSlots)
        return fail(ctx, MSFM_E_INVALID, "image id out of range");
    const Image& a = ctx->images[id1];
    const Image& b = ctx->images[id2];
    if (a.n < 0 || b.n < 0) return fail(ctx, MSFM_E_NOIMAGE, "image not uploaded: " + std::to_string(a.n < 0 ? id1 : id2));
    pd = PairDesc{};
    pd.a_panel = a.panel;
    pd.b_panel = b.panel;
    pd.a_raw = a.raw;
    pd.b_raw = b.raw;
    pd.n1 = a.n;
    pd.n2 = b.n;
    pd.a_blocks = a.nblk;
    pd.b_tiles = b.nblk;
    pd.n1pad = a.nblk * kBM;
    pd.n2pad = b.nblk * kBN;
    pd.ranges = 1;
    // empty query or train set: knnMatch returns nothing, no device work
    pd.valid = (a.n >= 1 && b.n >= 1) ? 1 : 0;
    return MSFM_OK;
}

void assign_offsets(Batch& b, int target_items) {
    long long total_ablocks = 0;
    for (auto& pd : b.pairs)
        if (pd.valid) total_ablocks += pd.a_blocks;
    for (auto& pd : b.pairs) {
        if (pd.valid && total_ablocks > 0 && total_ablocks < target_items) {
            long long r = (target_items + total_ablocks - 1) / total_ablocks;
            pd.ranges = (int)std::max<long long>(1, std::min<long long>(r, pd.b_tiles));
        }
        pd.rp_off = b.rp_elems;
        pd.cp_off = b.cp_elems;
        pd.kf_off = b.kf_elems;
        pd.kr_off = b.kr_elems;
        pd.out_off = b.out_elems;
        if (pd.valid) {
            b.rp_elems += (long long)pd.ranges * pd.n1pad;
            b.cp_elems += (long long)pd.a_blocks * pd.n2pad;
            b.kf_elems += pd.n1pad;
            b.kr_elems += pd.n2pad;
            b.out_elems += pd.n1;
            b.desc_pairs += (int64_t)pd.n1 * pd.n2;
            // compulsory traffic, no cross-pair reuse: both descriptor sets once + both knn lists
            b.algo_bytes += ((int64_t)pd.n1 + pd.n2) * kDim * 4 + ((int64_t)pd.n1 + pd.n2) * 12;
            b.max_npad = std::max(b.max_npad, std::max(pd.n1pad, pd.n2pad));
        }
    }
}

hipEvent_t get_event(msfm_ctx* ctx, size_t i) {
    while (ctx->ev_pool.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[i];
}

// distance + merge + tie fix-up for one batch (device arrays left in ctx buffers)
int run_knn(msfm_ctx* ctx, Batch& b, size_t ev_base) {
    const size_t P = b.pairs.size();
    HIPCHK(ctx, ctx->d_pairs.ensure(P * sizeof(PairDesc)));
    HIPCHK(ctx, ctx->d_items.ensure(std::max<size_t>(1, b.items.size()) * sizeof(WorkItem)));
    HIPCHK(ctx, ctx->d_rp_s0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, ctx->d_rp_i0.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, ctx->d_rp_s1.ensure(std::max<long long>(1, b.rp_elems) * 4));
    HIPCHK(ctx, ctx->d_cp_s0.ensure(std::max<long long>(1, b.cp_elems) * 4));
    HIPCHK(ctx, ctx->d_cp_i0.ensure(std::max<long long>(1, b.cp_elems) * 4));
    HIPCHK(ctx, ctx->d_cp_s1.ensure(std::max<long long>(1, b.cp_elems) * 4));
    const long long kn = std::max<long long>(1, b.kf_elems + b.kr_elems);
    // reverse arrays live behind the forward ones in the same buffers
    for (auto& pd : b.pairs) pd.kr_off += b.kf_elems;
    HIPCHK(ctx, ctx->d_k_i0.ensure(kn * 4));
    HIPCHK(ctx, ctx->d_k_d0.ensure(kn * 4));
    HIPCHK(ctx, ctx->d_k_d1.ensure(kn * 4));
    HIPCHK(ctx, ctx->d_fix_count.ensure(4));
    HIPCHK(ctx, ctx->d_fix_list.ensure((size_t)kFixCap * sizeof(int4)));

    HIPCHK(ctx, hipMemcpyAsync(ctx->d_pairs.p, b.pairs.data(), P * sizeof(PairDesc), hipMemcpyHostToDevice, ctx->stream));
    if (!b.items.empty())
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_items.p, b.items.data(), b.items.size() * sizeof(WorkItem), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_fix_count.p, 0, 4, ctx->stream));

    hipEvent_t e0 = get_event(ctx, ev_base), e1 = get_event(ctx, ev_base + 1);
    if (!e0 || !e1) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    if (!b.items.empty()) {
        HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
        const dim3 grid((unsigned)b.items.size()), block(kThreads);
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL(dist_top2_kernel<0>, grid, block, kLdsBytes, ctx->stream,
                               ctx->d_pairs.as<PairDesc>(), ctx->d_items.as<WorkItem>(),
                               ctx->d_rp_s0.as<float>(), ctx->d_rp_i0.as<int>(), ctx->d_rp_s1.as<float>(),
                               ctx->d_cp_s0.as<float>(), ctx->d_cp_i0.as<int>(), ctx->d_cp_s1.as<float>());
        else
            hipLaunchKernelGGL(dist_top2_kernel<1>, grid, block, kLdsBytes, ctx->stream,
                               ctx->d_pairs.as<PairDesc>(), ctx->d_items.as<WorkItem>(),
                               ctx->d_rp_s0.as<float>(), ctx->d_rp_i0.as<int>(), ctx->d_rp_s1.as<float>(),
                               ctx->d_cp_s0.as<float>(), ctx->d_cp_i0.as<int>(), ctx->d_cp_s1.as<float>());
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
        ctx->prof.dist_kernel_launches += 1;

        const dim3 mgrid((unsigned)((b.max_npad + 255) / 256), (unsigned)P);
        hipLaunchKernelGGL(merge_knn_kernel, mgrid, dim3(256), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(),
                           ctx->d_rp_s0.as<float>(), ctx->d_rp_i0.as<int>(), ctx->d_rp_s1.as<float>(),
                           ctx->d_cp_s0.as<float>(), ctx->d_cp_i0.as<int>(), ctx->d_cp_s1.as<float>(),
                           ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>(), ctx->d_k_d1.as<float>(),
                           ctx->d_fix_count.as<int>(), ctx->d_fix_list.as<int4>(), kFixCap);
        HIPCHK(ctx, hipGetLastError());
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL(tie_fixup_kernel<0>, dim3(1024), dim3(64), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(),
                               ctx->d_fix_count.as<int>(), ctx->d_fix_list.as<int4>(), kFixCap,
                               ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>());
        else
            hipLaunchKernelGGL(tie_fixup_kernel<1>, dim3(1024), dim3(64), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(),
                               ctx->d_fix_count.as<int>(), ctx->d_fix_list.as<int4>(), kFixCap,
                               ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>());
        HIPCHK(ctx, hipGetLastError());
    }
    ctx->prof.descriptor_pairs += b.desc_pairs;
    ctx->prof.dist_algo_bytes += b.algo_bytes;
    return MSFM_OK;
}

int check_fix_overflow(msfm_ctx* ctx) {
    int nfix = 0;
    HIPCHK(ctx, hipMemcpyAsync(&nfix, ctx->d_fix_count.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (nfix > kFixCap)
        return fail(ctx, MSFM_E_CAPACITY, "tie fix-up list overflow (" + std::to_string(nfix) + " tied rows in one batch)");
    return MSFM_OK;
}

int accumulate_kernel_time(msfm_ctx* ctx, size_t ev_base, bool launched) {
    if (!launched) return MSFM_OK;
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[ev_base], ctx->ev_pool[ev_base + 1]));
    ctx->prof.dist_kernel_ms += ms;
    return MSFM_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* msfm_version(void) { return "msfm-match 0.1 (gfx950)"; }

int msfm_create(int device_ordinal, msfm_ctx** out_ctx) {
    if (!out_ctx) return MSFM_E_INVALID;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MSFM_E_DEVICE;  // no GPU, no fallback
    if (device_ordinal < 0 || device_ordinal >= count) return MSFM_E_INVALID;
    if (hipSetDevice(device_ordinal) != hipSuccess) return MSFM_E_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return MSFM_E_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "msfm_create: device %d is %s, this library is built for gfx950 only\n",
                     device_ordinal, prop.gcnArchName);
        return MSFM_E_DEVICE;
    }
    msfm_ctx* ctx = new (std::nothrow) msfm_ctx();
    if (!ctx) return MSFM_E_DEVICE;
    ctx->device = device_ordinal;
    ctx->images.resize(kSlots);
    ctx->cu_count = prop.multiProcessorCount;
    ctx->clock_mhz = prop.clockRate / 1000;
    std::snprintf(ctx->dev_name, sizeof(ctx->dev_name), "%s (%s)", prop.name, prop.gcnArchName);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return MSFM_E_DEVICE;
    }
    // the distance kernel needs 108 KiB of dynamic LDS
    hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(dist_top2_kernel<0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(dist_top2_kernel<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e0 != hipSuccess || e1 != hipSuccess) {
        std::fprintf(stderr, "msfm_create: cannot reserve %d bytes of LDS: %s\n", kLdsBytes,
                     hipGetErrorString(e0 != hipSuccess ? e0 : e1));
        (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return MSFM_E_DEVICE;
    }
    *out_ctx = ctx;
    return MSFM_OK;
}

void msfm_destroy(msfm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& im : ctx->images) {
        if (im.panel) (void)hipFree(im.panel);
        if (im.raw) (void)hipFree(im.raw);
    }
    DevBuf* bufs[] = {&ctx->d_pairs, &ctx->d_items, &ctx->d_stage, &ctx->d_rp_s0, &ctx->d_rp_i0, &ctx->d_rp_s1,
                      &ctx->d_cp_s0, &ctx->d_cp_i0, &ctx->d_cp_s1, &ctx->d_k_i0, &ctx->d_k_d0, &ctx->d_k_d1,
                      &ctx->d_st_qt, &ctx->d_st_d, &ctx->d_counts, &ctx->d_offsets, &ctx->d_out_qt,
                      &ctx->d_out_d, &ctx->d_fix_count, &ctx->d_fix_list};
    for (DevBuf* b : bufs) b->release();
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* msfm_last_error(const msfm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int msfm_device_info(const msfm_ctx* ctx, char* name, int name_cap, int* cu_count, int* clock_mhz) {
    if (!ctx) return MSFM_E_INVALID;
    if (name && name_cap > 0) std::snprintf(name, (size_t)name_cap, "%s", ctx->dev_name);
    if (cu_count) *cu_count = ctx->cu_count;
    if (clock_mhz) *clock_mhz = ctx->clock_mhz;
    return MSFM_OK;
}

int msfm_set_accum_order(msfm_ctx* ctx, int order) {
    if (!ctx) return MSFM_E_INVALID;
    if (order != MSFM_ORDER_SSE4X4 && order != MSFM_ORDER_AVX2_FMA) return fail(ctx, MSFM_E_INVALID, "unknown accumulation order");
    if (order == ctx->order) return MSFM_OK;
    // the panel layout stores dimensions in accumulation order: re-lay every resident image
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->order = order;
    for (auto& im : ctx->images) {
        if (im.n <= 0) continue;
        const int blocks = std::min(4096, im.nblk * 16);
        if (order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, float>), dim3(blocks), dim3(256), 0, ctx->stream, im.raw, (float*)nullptr, im.panel, im.n, im.nblk);
        else
            hipLaunchKernelGGL((layout_kernel<1, float>), dim3(blocks), dim3(256), 0, ctx->stream, im.raw, (float*)nullptr, im.panel, im.n, im.nblk);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MSFM_OK;
}

int msfm_get_profile(const msfm_ctx* ctx, msfm_profile* out) {
    if (!ctx || !out) return MSFM_E_INVALID;
    *out = ctx->prof;
    return MSFM_OK;
}

int msfm_upload_image(msfm_ctx* ctx, int image_id, const void* desc, int n, int dim, int dtype) {
    if (!ctx) return MSFM_E_INVALID;
    if (image_id < 0 || image_id >= kSlots) return fail(ctx, MSFM_E_INVALID, "image id out of range");
    if (n < 0 || dim != MSFM_DIM) return fail(ctx, MSFM_E_INVALID, "descriptors must be n x 128");
    if (n >= (1 << 18)) return fail(ctx, MSFM_E_INVALID, "more than 2^18 - 1 rows (BFMatcher packs the train index in 18 bits)");
    if (dtype != MSFM_DTYPE_F32 && dtype != MSFM_DTYPE_U8) return fail(ctx, MSFM_E_INVALID, "dtype must be F32 or U8");
    if (n > 0 && !desc) return fail(ctx, MSFM_E_INVALID, "null descriptor pointer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Image& im = ctx->images[image_id];
    if (im.panel) (void)hipFree(im.panel);
    if (im.raw) (void)hipFree(im.raw);
    im = Image{};
    im.n = n;
    im.nblk = (n + kBM - 1) / kBM;
    if (n == 0) return MSFM_OK;
    HIPCHK(ctx, hipMalloc((void**)&im.panel, (size_t)im.nblk * kPanelFloats * 4));
    HIPCHK(ctx, hipMalloc((void**)&im.raw, (size_t)n * kDim * 4));
    const int blocks = std::min(4096, im.nblk * 16);
    if (dtype == MSFM_DTYPE_F32) {
        HIPCHK(ctx, hipMemcpyAsync(im.raw, desc, (size_t)n * kDim * 4, hipMemcpyHostToDevice, ctx->stream));
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, float>), dim3(blocks), dim3(256), 0, ctx->stream, im.raw, (float*)nullptr, im.panel, n, im.nblk);
        else
            hipLaunchKernelGGL((layout_kernel<1, float>), dim3(blocks), dim3(256), 0, ctx->stream, im.raw, (float*)nullptr, im.panel, n, im.nblk);
    } else {
        HIPCHK(ctx, ctx->d_stage.ensure((size_t)n * kDim));
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage.p, desc, (size_t)n * kDim, hipMemcpyHostToDevice, ctx->stream));
        if (ctx->order == MSFM_ORDER_SSE4X4)
            hipLaunchKernelGGL((layout_kernel<0, unsigned char>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_stage.as<unsigned char>(), im.raw, im.panel, n, im.nblk);
        else
            hipLaunchKernelGGL((layout_kernel<1, unsigned char>), dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_stage.as<unsigned char>(), im.raw, im.panel, n, im.nblk);
    }
    HIPCHK(ctx, hipGetLastError());
    // the caller may free/reuse `desc` (and we reuse d_stage) as soon as we return
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MSFM_OK;
}

int msfm_image_rows(const msfm_ctx* ctx, int image_id, int* out_n) {
    if (!ctx || !out_n) return MSFM_E_INVALID;
    if (image_id < 0 || image_id >= kSlots) return MSFM_E_INVALID;
    if (ctx->images[image_id].n < 0) return MSFM_E_NOIMAGE;
    *out_n = ctx->images[image_id].n;
    return MSFM_OK;
}

int msfm_clear_images(msfm_ctx* ctx) {
    if (!ctx) return MSFM_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& im : ctx->images) {
        if (im.panel) (void)hipFree(im.panel);
        if (im.raw) (void)hipFree(im.raw);
        im = Image{};
    }
    return MSFM_OK;
}

int msfm_match_pairs(msfm_ctx* ctx, const int32_t* pairs, int n_pairs, const msfm_match_params* params,
                     int64_t* out_offsets) {
    if (!ctx) return MSFM_E_INVALID;
    if (n_pairs < 0 || (n_pairs > 0 && !pairs) || !out_offsets) return fail(ctx, MSFM_E_INVALID, "bad pair list");
    msfm_match_params prm = {0.8f, 1, 0.7};
    if (params) prm = *params;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->have_results = false;
    ctx->res_offsets.assign((size_t)n_pairs + 1, 0);
    ctx->res_qt.clear();
    ctx->res_dist.clear();
    ctx->prof = msfm_profile{};

    hipEvent_t ev_begin = get_event(ctx, 0), ev_end = get_event(ctx, 1);
    if (!ev_begin || !ev_end) return fail(ctx, MSFM_E_DEVICE, "hipEventCreate failed");
    HIPCHK(ctx, hipEventRecord(ev_begin, ctx->stream));

    // sub-batches bounded by the partial-result scratch (12 B per partial entry)
    const long long kScratchElems = (long long)1 << 28;  // 3 GiB of partials at most
    const int kMaxPairsPerBatch = 4096;
    size_t ev_next = 2;
    std::vector<size_t> ev_of_batch;
    int begin = 0;
    while (begin < n_pairs) {
        Batch b;
        long long est = 0;
        int end = begin;
        while (end < n_pairs && (end - begin) < kMaxPairsPerBatch) {
            PairDesc pd;
            int rc = fill_pair(ctx, pairs[2 * end], pairs[2 * end + 1], pd);
            if (rc != MSFM_OK) return rc;
            const long long need = pd.valid ? ((long long)pd.n1pad + (long long)pd.a_blocks * pd.n2pad) : 0;
            if (end > begin && est + need > kScratchElems) break;
            est += need;
            b.pairs.push_back(pd);
            ++end;
        }
        assign_offsets(b, 2 * ctx->cu_count * 2);
        build_items(b);
        const size_t P = b.pairs.size();
        const size_t ev_base = ev_next;
        ev_next += 2;
        int rc = run_knn(ctx, b, ev_base);
        if (rc != MSFM_OK) return rc;

        HIPCHK(ctx, ctx->d_st_qt.ensure(std::max<long long>(1, b.out_elems) * sizeof(int2)));
        HIPCHK(ctx, ctx->d_st_d.ensure(std::max<long long>(1, b.out_elems) * 4));
        HIPCHK(ctx, ctx->d_out_qt.ensure(std::max<long long>(1, b.out_elems) * sizeof(int2)));
        HIPCHK(ctx, ctx->d_out_d.ensure(std::max<long long>(1, b.out_elems) * 4));
        HIPCHK(ctx, ctx->d_counts.ensure(P * 4));
        HIPCHK(ctx, ctx->d_offsets.ensure((P + 1) * 8));
        EpiParams ep = {prm.ratio, prm.cross_check, prm.max_distance};
        hipLaunchKernelGGL(epilogue_kernel, dim3((unsigned)P), dim3(256), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(), ep,
                           ctx->d_k_i0.as<int>(), ctx->d_k_d0.as<float>(), ctx->d_k_d1.as<float>(),
                           ctx->d_st_qt.as<int2>(), ctx->d_st_d.as<float>(), ctx->d_counts.as<int>());
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->d_counts.as<int>(),
                           ctx->d_offsets.as<long long>(), (int)P);
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)P), dim3(256), 0, ctx->stream, ctx->d_pairs.as<PairDesc>(),
                           ctx->d_counts.as<int>(), ctx->d_offsets.as<long long>(), ctx->d_st_qt.as<int2>(),
                           ctx->d_st_d.as<float>(), ctx->d_out_qt.as<int2>(), ctx->d_out_d.as<float>());
        HIPCHK(ctx, hipGetLastError());

        std::vector<long long> offs(P + 1);
        HIPCHK(ctx, hipMemcpyAsync(offs.data(), ctx->d_offsets.p, (P + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        rc = check_fix_overflow(ctx);  // synchronises the stream
        if (rc != MSFM_OK) return rc;
        const long long total = offs[P];
        const size_t base = ctx->res_dist.size();
        ctx->res_qt.resize(2 * (base + (size_t)total));
        ctx->res_dist.resize(base + (size_t)total);
        if (total > 0) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->res_qt.data() + 2 * base, ctx->d_out_qt.p, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->res_dist.data() + base, ctx->d_out_d.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        for (size_t p = 0; p < P; ++p) ctx->res_offsets[(size_t)begin + p + 1] = (int64_t)base + offs[p + 1];
        rc = accumulate_kernel_time(ctx, ev_base, !b.items.empty());
        if (rc != MSFM_OK) return rc;
        begin = end;
    }
    HIPCHK(ctx, hipEventRecord(ev_end, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ev_end));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ev_begin, ev_end));
    ctx->prof.total_device_ms = ms;
    std::memcpy(out_offsets, ctx->res_offsets.data(), ((size_t)n_pairs + 1) * sizeof(int64_t));
    ctx->have_results = true;
    return MSFM_OK;
}

int msfm_fetch_matches(msfm_ctx* ctx, int32_t* out_qt, float* out_dist) {
    if (!ctx) return MSFM_E_INVALID;
    if (!ctx->have_results) return fail(ctx, MSFM_E_STATE, "msfm_fetch_matches without a completed msfm_match_pairs");
    if (out_qt && !ctx->res_qt.empty()) std::memcpy(out_qt, ctx->res_qt.data(), ctx->res_qt.size() * 4);
    if (out_dist && !ctx->res_dist.empty()) std::memcpy(out_dist, ctx->res_dist.data(), ctx->res_dist.size() * 4);
    return MSFM_OK;
}

int msfm_match_pair(msfm_ctx* ctx, int id1, int id2, float ratio, int cross_check, double max_distance,
                    int32_t* out_qt, float* out_dist, int* out_count) {
    if (!ctx || !out_count) return MSFM_E_INVALID;
    const int32_t pr[2] = {id1, id2};
    msfm_match_params prm = {ratio, cross_check, max_distance};
    int64_t offs[2] = {0, 0};
    int rc = msfm_match_pairs(ctx, pr, 1, &prm, offs);
    if (rc != MSFM_OK) return rc;
    *out_count = (int)offs[1];
    return msfm_fetch_matches(ctx, out_qt, out_dist);
}

int msfm_knn2_pair(msfm_ctx* ctx, int id1, int id2, int32_t* fwd_idx0, float* fwd_d0, float* fwd_d1,
                   int32_t* rev_idx0, float* rev_d0, float* rev_d1) {
    if (!ctx) return MSFM_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->prof = msfm_profile{};
    Batch b;
    PairDesc pd;
    int rc = fill_pair(ctx, id1, id2, pd);
    if (rc != MSFM_OK) return rc;
    b.pairs.push_back(pd);
    assign_offsets(b, 2 * ctx->cu_count * 2);
    build_items(b);
    rc = run_knn(ctx, b, 2);
    if (rc != MSFM_OK) return rc;
    rc = check_fix_overflow(ctx);
    if (rc != MSFM_OK) return rc;
    rc = accumulate_kernel_time(ctx, 2, !b.items.empty());
    if (rc != MSFM_OK) return rc;
    const PairDesc& q = b.pairs[0];
    const int n1 = ctx->images[id1].n, n2 = ctx->images[id2].n;
    if (!q.valid) {
        // an empty side: no neighbours
        for (int i = 0; i < n1; ++i) {
            if (fwd_idx0) fwd_idx0[i] = -1;
            if (fwd_d0) fwd_d0[i] = 3.402823466e+38f;
            if (fwd_d1) fwd_d1[i] = 3.402823466e+38f;
        }
        for (int i = 0; i < n2; ++i) {
            if (rev_idx0) rev_idx0[i] = -1;
            if (rev_d0) rev_d0[i] = 3.402823466e+38f;
            if (rev_d1) rev_d1[i] = 3.402823466e+38f;
        }
        return MSFM_OK;
    }
    struct Cp { void* dst; const DevBuf* src; long long off; int n; };
    const Cp cps[6] = {{fwd_idx0, &ctx->d_k_i0, q.kf_off, n1}, {fwd_d0, &ctx->d_k_d0, q.kf_off, n1},
                       {fwd_d1, &ctx->d_k_d1, q.kf_off, n1},   {rev_idx0, &ctx->d_k_i0, q.kr_off, n2},
                       {rev_d0, &ctx->d_k_d0, q.kr_off, n2},   {rev_d1, &ctx->d_k_d1, q.kr_off, n2}};
    for (const Cp& c : cps)
        if (c.dst && c.n > 0)
            HIPCHK(ctx, hipMemcpyAsync(c.dst, c.src->as<char>() + c.off * 4, (size_t)c.n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MSFM_OK;
}

// ---- host-only helpers --------------------------------------------------------------------

int msfm_topscale_select(const float* kpts, int n, int k, int32_t* out_idx, int* out_count) {
    if (n < 0 || k < 0 || !out_idx || !out_count || (n > 0 && !kpts)) return MSFM_E_INVALID;
    if (k > n) {  // "if(num_features > kpts.size()) top_scale_descriptors = descriptors"
        for (int i = 0; i < n; ++i) out_idx[i] = i;
        *out_count = n;
        return MSFM_OK;
    }
    std::vector<int32_t> order((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    // documented tie rule: size descending, then index ascending (the reference's
    // std::partial_sort leaves the order of equal sizes unspecified)
    std::partial_sort(order.begin(), order.begin() + k, order.end(), [kpts](int32_t a, int32_t b) {
        const float sa = kpts[(size_t)a * 4 + 2], sb = kpts[(size_t)b * 4 + 2];
        if (sa != sb) return sa > sb;
        return a < b;
    });
    for (int i = 0; i < k; ++i) out_idx[i] = order[(size_t)i];
    *out_count = k;
    return MSFM_OK;
}

int msfm_swap_image_pair(int id1, int id2) { return id1 > id2 ? 1 : 0; }

int msfm_pair_id(int id1, int id2, int32_t* out_pair_id) {
    if (!out_pair_id || id1 < 0 || id2 < 0 || id1 >= MSFM_MAX_IMAGES || id2 >= MSFM_MAX_IMAGES) return MSFM_E_INVALID;
    *out_pair_id = msfm_swap_image_pair(id1, id2) ? MSFM_MAX_IMAGES * id2 + id1 : MSFM_MAX_IMAGES * id1 + id2;
    return MSFM_OK;
}

int msfm_pair_from_id(int32_t pair_id, int* out_id1, int* out_id2) {
    if (!out_id1 || !out_id2 || pair_id < 0) return MSFM_E_INVALID;
    *out_id2 = pair_id % MSFM_MAX_IMAGES;
    *out_id1 = (pair_id - *out_id2) / MSFM_MAX_IMAGES;
    return MSFM_OK;
}

}  // extern "C"
