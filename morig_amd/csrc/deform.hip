// DeformNet glue between CorrNet and GCNDeform (models/deformnet.py:41-97): visibility mask normalisation, cosine
// k-NN (k = num_interp; k = 1 is CorrNet's matching, models/corrnet.py:64) between feature sets of one cloud,
// similarity-weighted flow voting. The k-NN similarity matrix is GEMM-shaped and runs on the matrix cores; the other
// two are small per-vertex scans.
#include "common.h"

namespace morig {

// ---------------------------------------------------------------------------------------------------
// pred_vismask = sigmoid(logit); per mesh (m - min) / (max - min)   (deformnet.py:42-46)
// one workgroup per mesh; the sigmoid is recomputed in the second sweep (bit-identical to the first).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void sigmoid_minmax_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ ptr,
                                                             float* __restrict__ out, int ldo) {
    __shared__ float s_mn[4], s_mx[4];
    const int b = blockIdx.x;
    const int v0 = ptr[b], v1 = ptr[b + 1];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = v0 + threadIdx.x; i < v1; i += 256) {
        const float s = sigmoidf(x[(size_t)i * ldx]);
        mn = fminf(mn, s); mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
    mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    const float range = mx - mn;                       // 0 for a constant mask: 0/0 = NaN, as the reference
    for (int i = v0 + threadIdx.x; i < v1; i += 256) out[(size_t)i * ldo] = (sigmoidf(x[(size_t)i * ldx]) - mn) / range;
}

// ---------------------------------------------------------------------------------------------------
// knn(x, y, k, batch_x, batch_y, cosine=True) on L2-normalised rows (deformnet.py:49, :92; corrnet.py:64 with k = 1):
// for every query row of y the k rows of x of the same cloud with the largest dot product, most similar first, lowest
// index on ties (the order of torch_cluster's insertion scan).
//
// The similarity matrix IS a GEMM (queries x candidates, K = 64), so it runs on the matrix cores in the split-fp16
// arithmetic of the network kernels (3 f16 MFMAs per product, fp32 accumulate: dot products to ~1e-7, the rounding
// noise of an fp32 dot product). The MFMA result layout does the rest: with CANDIDATES as the M side and QUERIES as
// the N side of v_mfma_f32_32x32x16_f16, a lane owns ONE query (column lane & 31) and 16 candidate rows per tile, in
// increasing row order -- so every lane keeps a private running top-k in registers with no cross-lane traffic; the two
// lanes that share a query (lane, lane + 32: interleaved row groups of 4) merge their lists once at the end.
// A wave owns 64 queries (their B fragments stay in registers for the whole scan) and streams the cloud's candidates
// 32 rows at a time straight from global memory (every wave of the cloud reads the same rows: L2 hits).
// Features are scaled by 2^8 before the split so the low halves stay normal fp16 numbers (|x| <= 1).
//   split = 0: every y row queries every x row of its cloud.
//   split = 1: x and y are the same matrix with a visibility value per row: rows with vis < 0.5 query the rows with
//              vis >= 0.5 (the reference compacts both sets first, :57-63; indices here stay global).
// idx: [ny][k] global x rows, -1 where the cloud holds fewer than k candidates or the row does not query;
// sim (optional): [ny][k] the dot products.
// ---------------------------------------------------------------------------------------------------
constexpr int KNN_C = 64;
typedef float knn_f32x16 __attribute__((ext_vector_type(16)));
typedef float knn_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 knn_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 knn_h2 __attribute__((ext_vector_type(2)));

// 8 consecutive floats of one row -> (hi, lo) fp16 fragments of 2^8 * x
__device__ __forceinline__ void knn_split8(const float* __restrict__ src, knn_f16x8& hi, knn_f16x8& lo) {
    const knn_f32x4 a = *reinterpret_cast<const knn_f32x4*>(src), b = *reinterpret_cast<const knn_f32x4*>(src + 4);
    const float v[8] = {a[0] * 256.f, a[1] * 256.f, a[2] * 256.f, a[3] * 256.f, b[0] * 256.f, b[1] * 256.f, b[2] * 256.f, b[3] * 256.f};
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const knn_h2 h = __builtin_amdgcn_cvt_pkrtz(v[q], v[q + 1]);
        hi[q] = (_Float16)h[0]; hi[q + 1] = (_Float16)h[1];
        lo[q] = (_Float16)(v[q] - (float)h[0]); lo[q + 1] = (_Float16)(v[q + 1] - (float)h[1]);
    }
}

template <int K>
__global__ __launch_bounds__(256) void cosine_knn_kernel(const float* __restrict__ y, int ldy, const int* __restrict__ ptr_y,
                                                         const float* __restrict__ x, int ldx, const int* __restrict__ ptr_x,
                                                         const float* __restrict__ vis, int ld_vis, int split,
                                                         int* __restrict__ idx, float* __restrict__ sim) {
    constexpr int NT = 2;                                 // 2 x 32 queries per wave
    const int c = blockIdx.y;
    const int ys = ptr_y[c], ye = ptr_y[c + 1];
    const int xs = ptr_x[c], xe = ptr_x[c + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = ys + (blockIdx.x * 4 + wave) * (32 * NT);
    if (q0 >= ye) return;                                 // wave-uniform; the kernel has no block barrier

    // B fragments of my queries: B[k = 16 ks + 8 hi + j][col = l31]
    knn_f16x8 bh[NT][4], bl[NT][4];
    int qrow[NT]; bool asks[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        qrow[nt] = q0 + nt * 32 + l31;
        const bool live = qrow[nt] < ye;
        const int r = live ? qrow[nt] : ye - 1;
        asks[nt] = live && (!split || vis[(size_t)r * ld_vis] < 0.5f);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) knn_split8(y + (size_t)r * ldy + 16 * ks + 8 * hi, bh[nt][ks], bl[nt][ks]);
    }
    float bs[NT][K]; int bi[NT][K];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < K; ++j) { bs[nt][j] = -INFINITY; bi[nt][j] = 0x7fffffff; }

    for (int base = xs; base < xe; base += 32) {
        // A fragments of 32 candidate rows: A[row = l31][k = 16 ks + 8 hi + j]
        const int crow = min(base + l31, xe - 1);
        knn_f16x8 ah[4], al[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) knn_split8(x + (size_t)crow * ldx + 16 * ks + 8 * hi, ah[ks], al[ks]);
        // rows that may be selected: inside the cloud, and visible in split mode (bit r of `ok`)
        bool rok = base + l31 < xe;
        if (split && rok) rok = vis[(size_t)(base + l31) * ld_vis] >= 0.5f;
        const unsigned ok = (unsigned)(__ballot(rok) & 0xffffffffull);   // lanes 0..31 hold rows 0..31
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            knn_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[nt][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[nt][ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[nt][ks], acc, 0, 0, 0);
            }
            // my 16 values: candidate rows (r & 3) + 8 (r >> 2) + 4 hi of the tile, increasing in r
            float m = -INFINITY;
            if (ok != 0xffffffffu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (!((ok >> row) & 1u)) acc[r] = -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[r]);
            if (m > bs[nt][K - 1]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sv = acc[r];
                    if (sv > bs[nt][K - 1]) {              // strict: an equal later candidate never displaces an earlier one
                        bs[nt][K - 1] = sv; bi[nt][K - 1] = base + (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
                        for (int j = K - 1; j > 0; --j)
                            if (bs[nt][j] > bs[nt][j - 1]) {
                                const float ts = bs[nt][j]; bs[nt][j] = bs[nt][j - 1]; bs[nt][j - 1] = ts;
                                const int ti = bi[nt][j]; bi[nt][j] = bi[nt][j - 1]; bi[nt][j - 1] = ti;
                            }
                    }
                }
            }
        }
    }
    // merge the two half-lists of every query (lanes l and l + 32): value descending, then index ascending
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float os[K]; int oi[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { os[j] = __shfl_xor(bs[nt][j], 32); oi[j] = __shfl_xor(bi[nt][j], 32); }
        if (hi == 0 && qrow[nt] < ye) {
            int a = 0, b = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                // static indexing only (register arrays): pick heads by scanning
                float sa = -INFINITY, sb = -INFINITY; int ia = 0x7fffffff, ib = 0x7fffffff;
#pragma unroll
                for (int t = 0; t < K; ++t) { if (t == a) { sa = bs[nt][t]; ia = bi[nt][t]; } if (t == b) { sb = os[t]; ib = oi[t]; } }
                const bool take_a = a < K && (b >= K || sa > sb || (sa == sb && ia < ib));
                const float sv = take_a ? sa : sb; const int iv = take_a ? ia : ib;
                if (take_a) ++a; else ++b;
                const bool found = asks[nt] && sv > -INFINITY;
                idx[(size_t)qrow[nt] * K + j] = found ? iv : -1;
                if (sim) sim[(size_t)qrow[nt] * K + j] = found ? sv * (1.f / 65536.f) : 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Similarity-weighted voting (deformnet.py:50-54 and :93-95), one thread per vertex, neighbours in k-NN order:
//   mode 0, every vertex i:        w_t = <f_s[j_t], f_q[i]> * vis[i];  flow[i] = sum w_t (pos_s[j_t] - pos_q[i]) / sum w_t
//   mode 1, vertices with vis<0.5: w_t = <f_s[j_t], f_q[i]>;           flow[i] = sum w_t flow[j_t] / sum w_t
// (j_t visible, so their flow is final after mode 0). l1 rows are [flow(3) | vis] = GCNDeform's feature (:97).
// A vertex without neighbours or with zero weight sum gets 0/0 = NaN, as scatter_add/scatter_add does.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_vote_kernel(int mode, const int* __restrict__ idx, int k, int n,
                                                        const float* __restrict__ fq, int ldq, const float* __restrict__ fs, int lds, int C,
                                                        const float* __restrict__ pos_q, int ldpq, const float* __restrict__ pos_s, int ldps,
                                                        const float* __restrict__ vis, int ld_vis, float* __restrict__ l1, int ld_l1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float vi = vis[(size_t)i * ld_vis];
    if (mode == 1 && !(vi < 0.5f)) return;
    float ax = 0.f, ay = 0.f, az = 0.f, ws = 0.f;
    const float px = mode == 0 ? pos_q[(size_t)i * ldpq] : 0.f, py = mode == 0 ? pos_q[(size_t)i * ldpq + 1] : 0.f,
                pz = mode == 0 ? pos_q[(size_t)i * ldpq + 2] : 0.f;
    for (int t = 0; t < k; ++t) {
        const int j = idx[(size_t)i * k + t];
        if (j < 0) continue;
        float dot = 0.f;
        for (int cch = 0; cch < C; ++cch) dot += fs[(size_t)j * lds + cch] * fq[(size_t)i * ldq + cch];
        float w, vx, vy, vz;
        if (mode == 0) {
            w = dot * vi;
            vx = pos_s[(size_t)j * ldps] - px; vy = pos_s[(size_t)j * ldps + 1] - py; vz = pos_s[(size_t)j * ldps + 2] - pz;
        } else {
            w = dot;
            vx = l1[(size_t)j * ld_l1]; vy = l1[(size_t)j * ld_l1 + 1]; vz = l1[(size_t)j * ld_l1 + 2];
        }
        ax += vx * w; ay += vy * w; az += vz * w; ws += w;
    }
    l1[(size_t)i * ld_l1] = ax / ws; l1[(size_t)i * ld_l1 + 1] = ay / ws; l1[(size_t)i * ld_l1 + 2] = az / ws;
    if (mode == 0) l1[(size_t)i * ld_l1 + 3] = vi;
}

}  // namespace morig

using namespace morig;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int morig_sigmoid_minmax(const float* x, int32_t ldx, const int32_t* ptr, int32_t n_meshes, float* out, int32_t ldo,
                                    void* stream) {
    if (!x || !ptr || !out || n_meshes <= 0 || ldx < 1 || ldo < 1) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_FLOW_VOTE, s, 0.0, 0.0);
    hipLaunchKernelGGL(sigmoid_minmax_kernel, dim3(n_meshes), dim3(256), 0, s, x, ldx, ptr, out, ldo);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

static int launch_cosine_knn(const float* y, int ldy, const int* ptr_y, const float* x, int ldx, const int* ptr_x, int n_clouds,
                             int max_rows_per_cloud, int k, const float* vis, int ld_vis, int split, int* idx, float* sim,
                             hipStream_t s) {
    const dim3 grid(cdiv(max_rows_per_cloud, 256), n_clouds);
#define MORIG_KNN_CASE(KK) case KK: hipLaunchKernelGGL((cosine_knn_kernel<KK>), grid, dim3(256), 0, s, y, ldy, ptr_y, x, ldx, ptr_x, vis, ld_vis, split, idx, sim); break
    switch (k) {
        MORIG_KNN_CASE(1); MORIG_KNN_CASE(2); MORIG_KNN_CASE(3); MORIG_KNN_CASE(4);
        MORIG_KNN_CASE(5); MORIG_KNN_CASE(6); MORIG_KNN_CASE(7); MORIG_KNN_CASE(8);
        default: return MORIG_E_UNSUPPORTED;
    }
#undef MORIG_KNN_CASE
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_cosine_knn(const float* y, int32_t ldy, const int32_t* ptr_y, const float* x, int32_t ldx,
                                const int32_t* ptr_x, int32_t n_clouds, int32_t max_rows_per_cloud, int32_t C, int32_t k,
                                const float* vis, int32_t ld_vis, int32_t split, int32_t* idx, void* stream) {
    if (!y || !x || !ptr_y || !ptr_x || !idx || n_clouds <= 0 || max_rows_per_cloud <= 0) return MORIG_E_INVALID;
    if (C != KNN_C || k < 1 || k > 8) return MORIG_E_UNSUPPORTED;
    if (ldy < C || ldx < C || (ldy & 3) || (ldx & 3) || !aligned16(y) || !aligned16(x)) return MORIG_E_INVALID;
    if (split && (!vis || ld_vis < 1 || x != y || ptr_x != ptr_y)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_COSINE_KNN, s, 0.0, 0.0);
    return launch_cosine_knn(y, ldy, ptr_y, x, ldx, ptr_x, n_clouds, max_rows_per_cloud, k, vis, ld_vis, split, idx, nullptr, s);
}

// knn(out_pts, out_vtx, 1, cosine=True) (corrnet.py:64): the k = 1 case of the kernel above, with the similarity
extern "C" int morig_cosine_nn(const float* v, int32_t ldv, const int32_t* ptr_v, const float* p, int32_t ldp,
                               const int32_t* ptr_p, int32_t n_clouds, int32_t max_rows_per_cloud, int32_t C,
                               int32_t* nn, float* sim, void* stream) {
    if (!v || !p || !ptr_v || !ptr_p || !nn || !sim || n_clouds <= 0 || max_rows_per_cloud <= 0) return MORIG_E_INVALID;
    if (C != KNN_C) return MORIG_E_UNSUPPORTED;
    if (ldv < C || ldp < C || (ldv & 3) || (ldp & 3) || !aligned16(v) || !aligned16(p)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_COSINE_NN, s, 0.0, 0.0);
    return launch_cosine_knn(v, ldv, ptr_v, p, ldp, ptr_p, n_clouds, max_rows_per_cloud, 1, nullptr, 0, 0, nn, sim, s);
}

extern "C" int morig_flow_vote(int32_t mode, const int32_t* idx, int32_t k, int32_t n, const float* feat_q, int32_t ldq,
                               const float* feat_s, int32_t lds, int32_t C, const float* pos_q, int32_t ldpq,
                               const float* pos_s, int32_t ldps, const float* vis, int32_t ld_vis, float* l1, int32_t ld_l1,
                               void* stream) {
    if (!idx || !feat_q || !feat_s || !vis || !l1 || n < 0 || k < 1 || C <= 0 || ldq < C || lds < C || ld_l1 < 4 || ld_vis < 1)
        return MORIG_E_INVALID;
    if (mode != 0 && mode != 1) return MORIG_E_INVALID;
    if (mode == 0 && (!pos_q || !pos_s || ldpq < 3 || ldps < 3)) return MORIG_E_INVALID;
    if (n == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_FLOW_VOTE, s, 0.0, 0.0);
    hipLaunchKernelGGL(flow_vote_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, mode, idx, k, n, feat_q, ldq, feat_s, lds, C, pos_q,
                       ldpq, pos_s, ldps, vis, ld_vis, l1, ld_l1);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}
