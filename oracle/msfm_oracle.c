/*
 * msfm_oracle.c -- CPU oracle for the ComputeMatches hot path (see msfm_oracle.h).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (no OpenCV, no reference golden vectors).
 *
 * Build: gcc -O3 -ffp-contract=off -fPIC -shared -pthread (see oracle/Makefile).
 * -ffp-contract=off is load-bearing: the SSE4X4 and SCALAR orders round the
 * multiply and the add separately.
 */
#include "msfm_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#define ORC_X86_SIMD 1
#endif

/* ------------------------------------------------------------------------- */
/* S(a,b): restatement of cv::hal::normL2Sqr_(const float*, const float*, n)  */
/* for n = 128 (8 iterations of 16 floats, no scalar tail).                   */
/* ------------------------------------------------------------------------- */

/* Plain-C statement of the SSE baseline order; the reference form the others
 * (intrinsics, numpy, HIP kernel) are checked against. */
static float l2sqr_sse4x4_scalar(const float* a, const float* b)
{
    float p[16];
    for (int L = 0; L < 16; ++L) {
        float t = a[L] - b[L];
        p[L] = t * t; /* 0 + t*t is exact */
    }
    for (int it = 1; it < 8; ++it)
        for (int L = 0; L < 16; ++L) {
            float t = a[16 * it + L] - b[16 * it + L];
            float m = t * t;
            p[L] = p[L] + m;
        }
    float s[4];
    for (int l = 0; l < 4; ++l)
        s[l] = ((p[l] + p[4 + l]) + p[8 + l]) + p[12 + l];
    return (s[0] + s[2]) + (s[1] + s[3]);
}

#if defined(__SSE2__)
static float l2sqr_sse4x4(const float* a, const float* b)
{
    __m128 d0 = _mm_setzero_ps(), d1 = _mm_setzero_ps(), d2 = _mm_setzero_ps(), d3 = _mm_setzero_ps();
    for (int j = 0; j < 128; j += 16) {
        __m128 t0 = _mm_sub_ps(_mm_loadu_ps(a + j), _mm_loadu_ps(b + j));
        __m128 t1 = _mm_sub_ps(_mm_loadu_ps(a + j + 4), _mm_loadu_ps(b + j + 4));
        __m128 t2 = _mm_sub_ps(_mm_loadu_ps(a + j + 8), _mm_loadu_ps(b + j + 8));
        __m128 t3 = _mm_sub_ps(_mm_loadu_ps(a + j + 12), _mm_loadu_ps(b + j + 12));
        d0 = _mm_add_ps(_mm_mul_ps(t0, t0), d0);
        d1 = _mm_add_ps(_mm_mul_ps(t1, t1), d1);
        d2 = _mm_add_ps(_mm_mul_ps(t2, t2), d2);
        d3 = _mm_add_ps(_mm_mul_ps(t3, t3), d3);
    }
    __m128 v = _mm_add_ps(_mm_add_ps(_mm_add_ps(d0, d1), d2), d3);
    /* v_reduce_sum(v_float32x4): (l0+l2)+(l1+l3) */
    v = _mm_add_ps(v, _mm_movehl_ps(v, v));
    v = _mm_add_ss(v, _mm_shuffle_ps(v, v, _MM_SHUFFLE(0, 0, 0, 1)));
    return _mm_cvtss_f32(v);
}
#else
#define l2sqr_sse4x4 l2sqr_sse4x4_scalar
#endif

static float l2sqr_avx2_fma_plainc(const float* a, const float* b)
{
    float p[32];
    for (int L = 0; L < 32; ++L) p[L] = 0.0f;
    for (int it = 0; it < 4; ++it)
        for (int L = 0; L < 32; ++L) {
            float t = a[32 * it + L] - b[32 * it + L];
            p[L] = fmaf(t, t, p[L]);
        }
    float s[8];
    for (int l = 0; l < 8; ++l)
        s[l] = ((p[l] + p[8 + l]) + p[16 + l]) + p[24 + l];
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

/* AVX-512 + FMA3 build of the same universal-intrinsics loop (CV_SIMD512: 16 lanes, four accumulators, two iterations,
 * v_muladd fused), restated from memory like the others: lane partials p[16 v + L], s[L] = ((p0+p1)+p2)+p3
 * per lane, then v_reduce_sum(v_float32x16): low + high 256-bit halves, low + high 128-bit halves, and the four-lane sum of
 * the SSE order:  y_l = (s_l + s_{l+8}) + (s_{l+4} + s_{l+12}),  result = (y0 + y2) + (y1 + y3).
 * A THIRD named order: its purpose is the cross-check of the order-invariance certificate (include/msfm_match.h), not a
 * claim about a particular OpenCV binary. */
static float l2sqr_avx512_fma_plainc(const float* a, const float* b)
{
    float p[64];
    for (int L = 0; L < 64; ++L) p[L] = 0.0f;
    for (int it = 0; it < 2; ++it)
        for (int L = 0; L < 64; ++L) {
            float t = a[64 * it + L] - b[64 * it + L];
            p[L] = fmaf(t, t, p[L]);
        }
    float s[16], y[4];
    for (int l = 0; l < 16; ++l)
        s[l] = ((p[l] + p[16 + l]) + p[32 + l]) + p[48 + l];
    for (int l = 0; l < 4; ++l)
        y[l] = (s[l] + s[8 + l]) + (s[4 + l] + s[12 + l]);
    return (y[0] + y[2]) + (y[1] + y[3]);
}

/* The fused orders with REAL intrinsics (round 6: the CPU baseline is bounded from the fast side too -- a real OpenCV on an AVX2 /
 * AVX-512 host dispatches these loops, not the SSE one): the same lane partials, the same reduction trees as the plain-C
 * restatements above, bit for bit (tests/test_oracle_kat.py compares them on random and adversarial rows).  Compiled with
 * per-function target attributes and chosen at run time (__builtin_cpu_supports): the library still loads on a host without them. */
#if defined(ORC_X86_SIMD)
__attribute__((target("avx2,fma"))) static inline float l2sqr_avx2_fma_simd(const float* a, const float* b)
{
    __m256 d0 = _mm256_setzero_ps(), d1 = _mm256_setzero_ps(), d2 = _mm256_setzero_ps(), d3 = _mm256_setzero_ps();
    for (int j = 0; j < 128; j += 32) {
        __m256 t0 = _mm256_sub_ps(_mm256_loadu_ps(a + j), _mm256_loadu_ps(b + j));
        __m256 t1 = _mm256_sub_ps(_mm256_loadu_ps(a + j + 8), _mm256_loadu_ps(b + j + 8));
        __m256 t2 = _mm256_sub_ps(_mm256_loadu_ps(a + j + 16), _mm256_loadu_ps(b + j + 16));
        __m256 t3 = _mm256_sub_ps(_mm256_loadu_ps(a + j + 24), _mm256_loadu_ps(b + j + 24));
        d0 = _mm256_fmadd_ps(t0, t0, d0);
        d1 = _mm256_fmadd_ps(t1, t1, d1);
        d2 = _mm256_fmadd_ps(t2, t2, d2);
        d3 = _mm256_fmadd_ps(t3, t3, d3);
    }
    __m256 v = _mm256_add_ps(_mm256_add_ps(_mm256_add_ps(d0, d1), d2), d3);   /* s[l] = ((p0 + p1) + p2) + p3 per lane */
    /* v_reduce_sum(v_float32x8): two horizontal adds inside the 128-bit halves, then low + high */
    v = _mm256_hadd_ps(v, v);      /* (s0+s1, s2+s3, ..) | (s4+s5, s6+s7, ..) */
    v = _mm256_hadd_ps(v, v);      /* ((s0+s1)+(s2+s3), ..) | ((s4+s5)+(s6+s7), ..) */
    return _mm_cvtss_f32(_mm_add_ss(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1)));
}
__attribute__((target("avx512f"))) static inline float l2sqr_avx512_fma_simd(const float* a, const float* b)
{
    __m512 d0 = _mm512_setzero_ps(), d1 = _mm512_setzero_ps(), d2 = _mm512_setzero_ps(), d3 = _mm512_setzero_ps();
    for (int j = 0; j < 128; j += 64) {
        __m512 t0 = _mm512_sub_ps(_mm512_loadu_ps(a + j), _mm512_loadu_ps(b + j));
        __m512 t1 = _mm512_sub_ps(_mm512_loadu_ps(a + j + 16), _mm512_loadu_ps(b + j + 16));
        __m512 t2 = _mm512_sub_ps(_mm512_loadu_ps(a + j + 32), _mm512_loadu_ps(b + j + 32));
        __m512 t3 = _mm512_sub_ps(_mm512_loadu_ps(a + j + 48), _mm512_loadu_ps(b + j + 48));
        d0 = _mm512_fmadd_ps(t0, t0, d0);
        d1 = _mm512_fmadd_ps(t1, t1, d1);
        d2 = _mm512_fmadd_ps(t2, t2, d2);
        d3 = _mm512_fmadd_ps(t3, t3, d3);
    }
    __m512 v = _mm512_add_ps(_mm512_add_ps(_mm512_add_ps(d0, d1), d2), d3);
    /* low + high 256-bit halves, low + high 128-bit halves, then the four-lane sum of the SSE order */
    __m256 h = _mm256_add_ps(_mm512_castps512_ps256(v), _mm256_castpd_ps(_mm512_extractf64x4_pd(_mm512_castps_pd(v), 1)));
    __m128 y = _mm_add_ps(_mm256_castps256_ps128(h), _mm256_extractf128_ps(h, 1));
    y = _mm_add_ps(y, _mm_movehl_ps(y, y));
    y = _mm_add_ss(y, _mm_shuffle_ps(y, y, _MM_SHUFFLE(0, 0, 0, 1)));
    return _mm_cvtss_f32(y);
}
#endif

/* bit 0: the AVX2 + FMA3 intrinsics are in use, bit 1: the AVX-512F ones (0: the plain-C restatements run instead) */
int orc_simd_level(void)
{
#if defined(ORC_X86_SIMD)
    static int level = -1;
    if (level < 0) {
        __builtin_cpu_init();
        level = ((__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0) | (__builtin_cpu_supports("avx512f") ? 2 : 0);
    }
    return level;
#else
    return 0;
#endif
}

static float l2sqr_avx2_fma(const float* a, const float* b)
{
#if defined(ORC_X86_SIMD)
    if (orc_simd_level() & 1) return l2sqr_avx2_fma_simd(a, b);
#endif
    return l2sqr_avx2_fma_plainc(a, b);
}

static float l2sqr_avx512_fma(const float* a, const float* b)
{
#if defined(ORC_X86_SIMD)
    if (orc_simd_level() & 2) return l2sqr_avx512_fma_simd(a, b);
#endif
    return l2sqr_avx512_fma_plainc(a, b);
}

static float l2sqr_scalar(const float* a, const float* b)
{
    float d = 0.0f;
    for (int c = 0; c < 128; ++c) {
        float t = a[c] - b[c];
        float m = t * t;
        d = d + m;
    }
    return d;
}

float orc_l2sqr(const float* a, const float* b, int order)
{
    switch (order) {
    case MSFM_ORC_ORDER_SSE4X4: return l2sqr_sse4x4(a, b);
    case MSFM_ORC_ORDER_AVX2_FMA: return l2sqr_avx2_fma(a, b);
    case MSFM_ORC_ORDER_SCALAR: return l2sqr_scalar(a, b);
    case MSFM_ORC_ORDER_AVX512_FMA: return l2sqr_avx512_fma(a, b);
    case 100: return l2sqr_sse4x4_scalar(a, b); /* test hook: plain-C SSE order */
    case 101: return l2sqr_avx2_fma_plainc(a, b);   /* test hooks: the plain-C statements of the fused orders */
    case 103: return l2sqr_avx512_fma_plainc(a, b);
    default: return NAN;
    }
}

/* ------------------------------------------------------------------------- */
/* knnMatch(k=2): restatement of cv::batchDistance(..., K=2, NORM_L2) as       */
/* called by BFMatcher::knnMatchImpl (call site FeatureUtils.cpp:146-149).    */
/* ------------------------------------------------------------------------- */

static inline int32_t f2i(float f)
{
    int32_t i;
    memcpy(&i, &f, 4);
    return i;
}
static inline float i2f(int32_t i)
{
    float f;
    memcpy(&f, &i, 4);
    return f;
}

static void knn2_rows(const float* q, int q_begin, int q_end, const float* t, int nt, int order,
                      int32_t* idx0, float* d0, int32_t* idx1, float* d1)
{
    const int K = nt < 2 ? nt : 2;
    float* buf = (float*)malloc(sizeof(float) * (size_t)(nt > 0 ? nt : 1));
    for (int i = q_begin; i < q_end; ++i) {
        const float* qi = q + (size_t)i * 128;
        /* batchDistL2_32f: dist[j] = std::sqrt(normL2Sqr(q, t_j)) */
        switch (order) {
        case MSFM_ORC_ORDER_SSE4X4:
            for (int j = 0; j < nt; ++j) buf[j] = sqrtf(l2sqr_sse4x4(qi, t + (size_t)j * 128));
            break;
        case MSFM_ORC_ORDER_AVX2_FMA:
            for (int j = 0; j < nt; ++j) buf[j] = sqrtf(l2sqr_avx2_fma(qi, t + (size_t)j * 128));
            break;
        default:
            for (int j = 0; j < nt; ++j) buf[j] = sqrtf(orc_l2sqr(qi, t + (size_t)j * 128, order));
        }
        /* dist initialised to FLT_MAX, nidx to -1; positive floats compared as ints */
        int32_t dist[2] = {f2i(FLT_MAX), f2i(FLT_MAX)};
        int32_t nidx[2] = {-1, -1};
        if (K > 0) {
            for (int j = 0; j < nt; ++j) {
                int32_t d = f2i(buf[j]);
                if (d < dist[K - 1]) {
                    int k;
                    for (k = K - 2; k >= 0 && dist[k] > d; --k) {
                        nidx[k + 1] = nidx[k];
                        dist[k + 1] = dist[k];
                    }
                    nidx[k + 1] = j;
                    dist[k + 1] = d;
                }
            }
        }
        idx0[i] = nidx[0];
        d0[i] = i2f(dist[0]);
        idx1[i] = nidx[1];
        d1[i] = i2f(dist[1]);
    }
    free(buf);
}

/* The same computation re-tiled for the cache: a block of QB query rows against one tile of TT train rows at a
 * time, tiles in ascending train order.  Per (query, train) element the arithmetic and the strict-< insertion are
 * those of knn2_rows, and every query still sees its train rows in ascending order, so the results are identical
 * bit for bit (tests/test_oracle_kat.py checks it); only the memory traffic differs (the train image is streamed once
 * per QB query rows instead of once per row).  Used by the pair-parallel CPU baseline, where 256 threads streaming
 * 2.5 MB per query row would measure the DRAM, not the cores. */
#define ORC_QB 32
#define ORC_TT 128
#define ORC_DEFINE_BLOCKED(NAME, ATTR, L2)                                                                             \
    ATTR static void NAME(const float* q, int q_begin, int q_end, const float* t, int nt, int order,                    \
                          int32_t* idx0, float* d0, int32_t* idx1, float* d1)                                          \
    {                                                                                                                  \
        (void)order;                                                                                                   \
        const int K = nt < 2 ? nt : 2;                                                                                 \
        int32_t dist[ORC_QB][2], nidx[ORC_QB][2];                                                                      \
        for (int b0 = q_begin; b0 < q_end; b0 += ORC_QB) {                                                             \
            const int nb = (q_end - b0 < ORC_QB) ? q_end - b0 : ORC_QB;                                                \
            for (int r = 0; r < nb; ++r) {                                                                             \
                dist[r][0] = dist[r][1] = f2i(FLT_MAX);                                                                \
                nidx[r][0] = nidx[r][1] = -1;                                                                          \
            }                                                                                                          \
            for (int j0 = 0; j0 < nt && K > 0; j0 += ORC_TT) {                                                         \
                const int j1 = (nt - j0 < ORC_TT) ? nt : j0 + ORC_TT;                                                  \
                for (int r = 0; r < nb; ++r) {                                                                         \
                    const float* qi = q + (size_t)(b0 + r) * 128;                                                      \
                    for (int j = j0; j < j1; ++j) {                                                                    \
                        const float s = L2;                                                                            \
                        const int32_t d = f2i(sqrtf(s));                                                               \
                        if (d < dist[r][K - 1]) {                                                                      \
                            int k;                                                                                     \
                            for (k = K - 2; k >= 0 && dist[r][k] > d; --k) {                                           \
                                nidx[r][k + 1] = nidx[r][k];                                                           \
                                dist[r][k + 1] = dist[r][k];                                                           \
                            }                                                                                          \
                            nidx[r][k + 1] = j;                                                                        \
                            dist[r][k + 1] = d;                                                                        \
                        }                                                                                              \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
            for (int r = 0; r < nb; ++r) {                                                                             \
                idx0[b0 + r] = nidx[r][0];                                                                             \
                d0[b0 + r] = i2f(dist[r][0]);                                                                          \
                idx1[b0 + r] = nidx[r][1];                                                                             \
                d1[b0 + r] = i2f(dist[r][1]);                                                                          \
            }                                                                                                          \
        }                                                                                                              \
    }
ORC_DEFINE_BLOCKED(knn2_rows_blocked_sse, , l2sqr_sse4x4(qi, t + (size_t)j * 128))
ORC_DEFINE_BLOCKED(knn2_rows_blocked_any, , orc_l2sqr(qi, t + (size_t)j * 128, order))
#if defined(ORC_X86_SIMD)
ORC_DEFINE_BLOCKED(knn2_rows_blocked_avx2, __attribute__((target("avx2,fma"))), l2sqr_avx2_fma_simd(qi, t + (size_t)j * 128))
ORC_DEFINE_BLOCKED(knn2_rows_blocked_avx512, __attribute__((target("avx512f"))), l2sqr_avx512_fma_simd(qi, t + (size_t)j * 128))
#endif
static void knn2_rows_blocked(const float* q, int q_begin, int q_end, const float* t, int nt, int order,
                              int32_t* idx0, float* d0, int32_t* idx1, float* d1)
{
    if (order == MSFM_ORC_ORDER_SSE4X4) return knn2_rows_blocked_sse(q, q_begin, q_end, t, nt, order, idx0, d0, idx1, d1);
#if defined(ORC_X86_SIMD)
    if (order == MSFM_ORC_ORDER_AVX2_FMA && (orc_simd_level() & 1)) return knn2_rows_blocked_avx2(q, q_begin, q_end, t, nt, order, idx0, d0, idx1, d1);
    if (order == MSFM_ORC_ORDER_AVX512_FMA && (orc_simd_level() & 2)) return knn2_rows_blocked_avx512(q, q_begin, q_end, t, nt, order, idx0, d0, idx1, d1);
#endif
    knn2_rows_blocked_any(q, q_begin, q_end, t, nt, order, idx0, d0, idx1, d1);
}

static __thread int g_use_blocked = 0;   /* set by the pair-parallel workers */

void orc_knn2_blocked(const float* q, int nq, const float* t, int nt, int order,
                      int32_t* idx0, float* d0, int32_t* idx1, float* d1)
{
    knn2_rows_blocked(q, 0, nq, t, nt, order, idx0, d0, idx1, d1);
}

void orc_knn2(const float* q, int nq, const float* t, int nt, int order,
              int32_t* idx0, float* d0, int32_t* idx1, float* d1)
{
    knn2_rows(q, 0, nq, t, nt, order, idx0, d0, idx1, d1);
}

typedef struct {
    const float *q, *t;
    int q_begin, q_end, nt, order;
    int32_t *idx0, *idx1;
    float *d0, *d1;
} knn_job;

static void* knn_thread(void* arg)
{
    knn_job* j = (knn_job*)arg;
    knn2_rows(j->q, j->q_begin, j->q_end, j->t, j->nt, j->order, j->idx0, j->d0, j->idx1, j->d1);
    return NULL;
}

void orc_knn2_mt(const float* q, int nq, const float* t, int nt, int order, int nthreads,
                 int32_t* idx0, float* d0, int32_t* idx1, float* d1)
{
    if (nthreads <= 1 || nq < 2 * nthreads) {
        if (g_use_blocked) knn2_rows_blocked(q, 0, nq, t, nt, order, idx0, d0, idx1, d1);
        else knn2_rows(q, 0, nq, t, nt, order, idx0, d0, idx1, d1);
        return;
    }
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    knn_job* jobs = (knn_job*)malloc(sizeof(knn_job) * (size_t)nthreads);
    for (int k = 0; k < nthreads; ++k) {
        knn_job jb = {q, t, (int)((long long)nq * k / nthreads), (int)((long long)nq * (k + 1) / nthreads),
                      nt, order, idx0, idx1, d0, d1};
        jobs[k] = jb;
        pthread_create(&th[k], NULL, knn_thread, &jobs[k]);
    }
    for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------------- */
/* FeatureUtils operators                                                     */
/* ------------------------------------------------------------------------- */

int orc_compute_matches(const float* d1, int n1, const float* d2, int n2, float ratio,
                        int order, int nthreads, int32_t* out_q, int32_t* out_t, float* out_d)
{
    /* FeatureUtils.cpp:152 indexes m[1] unconditionally: UB when the train set has
     * < 2 rows.  Build-defined behaviour: no matches.  Empty query/train => none. */
    if (n1 <= 0 || n2 < 2) return 0;
    int32_t* i0 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n1 * 2);
    float* dd = (float*)malloc(sizeof(float) * (size_t)n1 * 2);
    int32_t* i1 = i0 + n1;
    float *dist0 = dd, *dist1 = dd + n1;
    orc_knn2_mt(d1, n1, d2, n2, order, nthreads, i0, dist0, i1, dist1);
    int m = 0;
    for (int q = 0; q < n1; ++q) {
        if (i0[q] < 0 || i1[q] < 0) continue; /* fewer than 2 finite neighbours */
        float thr = ratio * dist1[q]; /* single-rounded fp32 product */
        if (dist0[q] < thr) {
            out_q[m] = q;
            out_t[m] = i0[q];
            out_d[m] = dist0[q];
            ++m;
        }
    }
    free(i0);
    free(dd);
    return m;
}

int orc_cross_check(const int32_t* q12, const int32_t* t12, const float* d12, int m12,
                    const int32_t* q21, const int32_t* t21, int m21,
                    int32_t* out_q, int32_t* out_t, float* out_d)
{
    /* vis[query_idx of reverse match] = its train_idx; unordered_map::operator[] on
     * lookup default-inserts 0 for a missing key (FeatureUtils.cpp:302). */
    int32_t maxkey = -1;
    for (int i = 0; i < m21; ++i)
        if (q21[i] > maxkey) maxkey = q21[i];
    for (int i = 0; i < m12; ++i)
        if (t12[i] > maxkey) maxkey = t12[i];
    int32_t* vis = (int32_t*)calloc((size_t)(maxkey + 2), sizeof(int32_t)); /* missing -> 0 */
    for (int i = 0; i < m21; ++i) vis[q21[i]] = t21[i];
    int m = 0;
    for (int i = 0; i < m12; ++i) {
        if (vis[t12[i]] == q12[i]) {
            out_q[m] = q12[i];
            out_t[m] = t12[i];
            out_d[m] = d12[i];
            ++m;
        }
    }
    free(vis);
    return m;
}

int orc_filter_by_distance(const int32_t* q, const int32_t* t, const float* d, int m,
                           double max_distance, int32_t* out_q, int32_t* out_t, float* out_d)
{
    int k = 0;
    for (int i = 0; i < m; ++i) {
        if ((double)d[i] > max_distance) continue;
        out_q[k] = q[i];
        out_t[k] = t[i];
        out_d[k] = d[i];
        ++k;
    }
    return k;
}

int orc_match_pair(const float* d1, int n1, const float* d2, int n2,
                   float ratio, int cross_check, double max_distance, int order, int nthreads,
                   int32_t* out_q, int32_t* out_t, float* out_d)
{
    if (n1 <= 0 || n2 <= 0) return 0;
    int cap = n1 > n2 ? n1 : n2;
    int32_t* ibuf = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap * 6);
    float* fbuf = (float*)malloc(sizeof(float) * (size_t)cap * 3);
    int32_t *q12 = ibuf, *t12 = ibuf + cap, *q21 = ibuf + 2 * cap, *t21 = ibuf + 3 * cap;
    int32_t *qc = ibuf + 4 * cap, *tc = ibuf + 5 * cap;
    float *dd12 = fbuf, *dd21 = fbuf + cap, *dc = fbuf + 2 * cap;
    int m12 = orc_compute_matches(d1, n1, d2, n2, ratio, order, nthreads, q12, t12, dd12);
    int m;
    if (cross_check) {
        /* build-defined: a side with < 2 rows yields no matches in that direction (UB in the
         * reference); the GPU library additionally returns 0 matches for the whole pair then. */
        int m21 = orc_compute_matches(d2, n2, d1, n1, ratio, order, nthreads, q21, t21, dd21);
        if (n1 < 2 || n2 < 2) m = 0;
        else m = orc_cross_check(q12, t12, dd12, m12, q21, t21, m21, qc, tc, dc);
    } else {
        m = m12;
        memcpy(qc, q12, sizeof(int32_t) * (size_t)m);
        memcpy(tc, t12, sizeof(int32_t) * (size_t)m);
        memcpy(dc, dd12, sizeof(float) * (size_t)m);
    }
    int k = orc_filter_by_distance(qc, tc, dc, m, max_distance, out_q, out_t, out_d);
    free(ibuf);
    free(fbuf);
    return k;
}

/* ------------------------------------------------------------------------- */
/* pre-emptive matching helper (FeatureUtils.cpp:68-96)                       */
/* ------------------------------------------------------------------------- */
/* a batch of independent pairs, parallel over pairs (FeatureMatching.cpp:14-49) */
/* ------------------------------------------------------------------------- */
typedef struct {
    const float* const* images;
    const int32_t* rows;
    const int32_t* pairs;
    int n_pairs;
    float ratio;
    int cross_check;
    double max_distance;
    int order;
    const int64_t* cap_off;  /* where pair p's staging region starts */
    int32_t *q, *t;
    float* d;
    int32_t* counts;
    int next;                /* shared work counter */
    double deadline;         /* CLOCK_MONOTONIC seconds after which no new pair is started (<= 0: none) */
} pairs_job;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* pairs_thread(void* arg)
{
    pairs_job* J = (pairs_job*)arg;
    g_use_blocked = 1;   /* same results, cache-blocked loop order (knn2_rows_blocked) */
    for (;;) {
        if (J->deadline > 0.0 && now_s() > J->deadline) break;
        const int p = __atomic_fetch_add(&J->next, 1, __ATOMIC_RELAXED);
        if (p >= J->n_pairs) break;
        const int32_t a = J->pairs[2 * p], b = J->pairs[2 * p + 1];
        const int64_t o = J->cap_off[p];
        J->counts[p] = orc_match_pair(J->images[a], J->rows[a], J->images[b], J->rows[b], J->ratio, J->cross_check,
                                      J->max_distance, J->order, 1, J->q + o, J->t + o, J->d + o);
    }
    return NULL;
}

int64_t orc_match_pairs_mt(const float* const* images, const int32_t* rows, const int32_t* pairs, int n_pairs,
                           float ratio, int cross_check, double max_distance, int order, int nthreads, double budget_s,
                           int* n_done, int64_t* out_offsets, int32_t* out_q, int32_t* out_t, float* out_d)
{
    out_offsets[0] = 0;
    if (n_done) *n_done = 0;
    if (n_pairs <= 0) return 0;
    int64_t* cap_off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_pairs + 1));
    int32_t* counts = (int32_t*)calloc((size_t)n_pairs, sizeof(int32_t));
    cap_off[0] = 0;
    for (int p = 0; p < n_pairs; ++p) cap_off[p + 1] = cap_off[p] + (rows[pairs[2 * p]] > 0 ? rows[pairs[2 * p]] : 0);
    pairs_job J = {images, rows, pairs, n_pairs, ratio, cross_check, max_distance, order, cap_off,
                   out_q, out_t, out_d, counts, 0, budget_s > 0.0 ? now_s() + budget_s : 0.0};
    for (int p = 0; p < n_pairs; ++p) counts[p] = -1;   /* -1: not computed (budget ran out) */
    if (nthreads > n_pairs) nthreads = n_pairs;
    if (nthreads <= 1) {
        const int keep = g_use_blocked;
        pairs_thread(&J);
        g_use_blocked = keep;
    } else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int k = 0; k < nthreads; ++k) pthread_create(&th[k], NULL, pairs_thread, &J);
        for (int k = 0; k < nthreads; ++k) pthread_join(th[k], NULL);
        free(th);
    }
    /* compact the per-pair staging regions into CSR order (regions are ordered, so an in-place forward move is safe) */
    int64_t at = 0;
    int done = n_pairs;
    for (int p = 0; p < n_pairs; ++p)
        if (counts[p] < 0) { done = p; break; }   /* pairs are started in order: the computed ones are a prefix */
    if (n_done) *n_done = done;
    for (int p = done; p < n_pairs; ++p) counts[p] = 0;
    for (int p = 0; p < n_pairs; ++p) {
        const int64_t o = cap_off[p];
        if (o != at && counts[p] > 0) {
            memmove(out_q + at, out_q + o, sizeof(int32_t) * (size_t)counts[p]);
            memmove(out_t + at, out_t + o, sizeof(int32_t) * (size_t)counts[p]);
            memmove(out_d + at, out_d + o, sizeof(float) * (size_t)counts[p]);
        }
        at += counts[p];
        out_offsets[p + 1] = at;
    }
    free(cap_off);
    free(counts);
    return at;
}

/* ------------------------------------------------------------------------- */

typedef struct {
    float size;
    int32_t idx;
} scale_ent;

static int scale_cmp(const void* a, const void* b)
{
    const scale_ent* x = (const scale_ent*)a;
    const scale_ent* y = (const scale_ent*)b;
    if (x->size > y->size) return -1;
    if (x->size < y->size) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int orc_topscale_select(const float* kpts, int n, int k, int32_t* out_idx)
{
    if (k > n) { /* "if(num_features > kpts.size()) top = descriptors" */
        for (int i = 0; i < n; ++i) out_idx[i] = i;
        return n;
    }
    scale_ent* e = (scale_ent*)malloc(sizeof(scale_ent) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        e[i].size = kpts[(size_t)i * 4 + 2];
        e[i].idx = i;
    }
    qsort(e, (size_t)n, sizeof(scale_ent), scale_cmp);
    for (int i = 0; i < k; ++i) out_idx[i] = e[i].idx;
    free(e);
    return k;
}

/* ------------------------------------------------------------------------- */
/* pair id codec (Database.cpp:6, 656-694)                                    */
/* ------------------------------------------------------------------------- */

#define K_MAX_NUM_IMAGES 10000

int orc_swap_image_pair(int32_t id1, int32_t id2) { return id1 > id2; }

int32_t orc_pair_id(int32_t id1, int32_t id2)
{
    if (orc_swap_image_pair(id1, id2)) return K_MAX_NUM_IMAGES * id2 + id1;
    return K_MAX_NUM_IMAGES * id1 + id2;
}

void orc_pair_from_id(int32_t pair_id, int32_t* id1, int32_t* id2)
{
    *id2 = pair_id % K_MAX_NUM_IMAGES;
    *id1 = (pair_id - *id2) / K_MAX_NUM_IMAGES;
}

/* ------------------------------------------------------------------------- */
/* pair enumeration (FeatureMatching.cpp:75-145)                              */
/* ------------------------------------------------------------------------- */

int64_t orc_enumerate_brute(int n_images, int max_pairs, int32_t* pairs, int64_t* batch_end,
                            int64_t* n_batches)
{
    int64_t np = 0, nb = 0;
    for (int i = 0; i < n_images; ++i) {
        int cur = 0;
        for (int j = 0; j < i; ++j) {
            if (pairs) {
                pairs[2 * np] = i;
                pairs[2 * np + 1] = j;
            }
            ++np;
            ++cur;
            if (cur == max_pairs) {
                if (batch_end) batch_end[nb] = np;
                ++nb;
                cur = 0;
            }
        }
        if (cur != 0) {
            if (batch_end) batch_end[nb] = np;
            ++nb;
        }
    }
    if (n_batches) *n_batches = nb;
    return np;
}

int64_t orc_enumerate_sequential(int n_images, int overlap, int32_t* pairs, int64_t* batch_end,
                                 int64_t* n_batches)
{
    int64_t np = 0, nb = 0;
    for (int i = 1; i < n_images; ++i) {
        for (int k = 1; k <= overlap; ++k) {
            int j = i - k;
            if (j < 0) break;
            if (pairs) {
                pairs[2 * np] = i;
                pairs[2 * np + 1] = j;
            }
            ++np;
        }
        /* MatchImagePairs is called once per i, even though the list is never empty for i>=1 */
        if (batch_end) batch_end[nb] = np;
        ++nb;
    }
    if (n_batches) *n_batches = nb;
    return np;
}
