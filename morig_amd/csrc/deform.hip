// DeformNet glue between CorrNet and GCNDeform (models/deformnet.py:41-97): visibility mask normalisation, cosine
// k-NN (k = num_interp) between feature sets of one cloud, similarity-weighted flow voting. All three are scans of one
// cloud at a time in fp32 VALU: no GEMM shape, bounded by LDS/HBM streaming of the candidate features.
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
// knn(x, y, k, batch_x, batch_y, cosine=True) on L2-normalised rows (deformnet.py:49, :92): for every query row of
// y the k rows of x of the same cloud with the largest dot product, most similar first, lowest index on ties
// (the stable order of torch_cluster's insertion scan). One thread per query (its 64-vector and the running top-k
// in registers), candidates streamed through LDS 128 rows at a time.
//   split = 0: every y row queries every x row of its cloud.
//   split = 1: x and y are the same matrix with a visibility value per row: rows with vis < 0.5 query the rows with
//              vis >= 0.5 (the reference compacts both sets first, :57-63; indices here stay global).
// idx: [ny][k] global x rows, -1 where the cloud holds fewer than k candidates or the row does not query.
// ---------------------------------------------------------------------------------------------------
constexpr int KNN_C = 64;
constexpr int KNN_TILE = 128;

template <int K>
__global__ __launch_bounds__(256) void cosine_knn_kernel(const float* __restrict__ y, int ldy, const int* __restrict__ ptr_y,
                                                         const float* __restrict__ x, int ldx, const int* __restrict__ ptr_x,
                                                         const float* __restrict__ vis, int ld_vis, int split,
                                                         int* __restrict__ idx) {
    __shared__ float sp[KNN_TILE * KNN_C];
    __shared__ int s_ok[KNN_TILE];
    const int c = blockIdx.y;
    const int ys = ptr_y[c], ye = ptr_y[c + 1];
    const int xs = ptr_x[c], xe = ptr_x[c + 1];
    if (ys + (int)(blockIdx.x * blockDim.x) >= ye) return;
    const int t = ys + blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < ye;
    const bool asks = live && (!split || vis[(size_t)t * ld_vis] < 0.5f);
    float q[KNN_C];
#pragma unroll
    for (int i = 0; i < KNN_C; ++i) q[i] = live ? y[(size_t)t * ldy + i] : 0.f;
    float bs[K]; int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bs[j] = -INFINITY; bi[j] = -1; }
    for (int base = xs; base < xe; base += KNN_TILE) {
        const int cnt = min(KNN_TILE, xe - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * KNN_C; i += blockDim.x) {
            const int r = i / KNN_C, cc = i - r * KNN_C;
            sp[i] = x[(size_t)(base + r) * ldx + cc];
        }
        if (threadIdx.x < cnt) s_ok[threadIdx.x] = !split || vis[(size_t)(base + threadIdx.x) * ld_vis] >= 0.5f;
        __syncthreads();
        if (!asks) continue;
        for (int r = 0; r < cnt; ++r) {
            if (!s_ok[r]) continue;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < KNN_C; i += 4) {
                a0 += q[i] * sp[r * KNN_C + i];
                a1 += q[i + 1] * sp[r * KNN_C + i + 1];
                a2 += q[i + 2] * sp[r * KNN_C + i + 2];
                a3 += q[i + 3] * sp[r * KNN_C + i + 3];
            }
            const float s = (a0 + a1) + (a2 + a3);
            if (s > bs[K - 1]) {                      // strict: an equal later candidate never displaces an earlier one
                bs[K - 1] = s; bi[K - 1] = base + r;
#pragma unroll
                for (int j = K - 1; j > 0; --j)
                    if (bs[j] > bs[j - 1]) {
                        const float ts = bs[j]; bs[j] = bs[j - 1]; bs[j - 1] = ts;
                        const int ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
                    }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < K; ++j) idx[(size_t)t * K + j] = bi[j];
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

extern "C" int morig_sigmoid_minmax(const float* x, int32_t ldx, const int32_t* ptr, int32_t n_meshes, float* out, int32_t ldo,
                                    void* stream) {
    if (!x || !ptr || !out || n_meshes <= 0 || ldx < 1 || ldo < 1) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_FLOW_VOTE, s, 0.0, 0.0);
    hipLaunchKernelGGL(sigmoid_minmax_kernel, dim3(n_meshes), dim3(256), 0, s, x, ldx, ptr, out, ldo);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_cosine_knn(const float* y, int32_t ldy, const int32_t* ptr_y, const float* x, int32_t ldx,
                                const int32_t* ptr_x, int32_t n_clouds, int32_t max_rows_per_cloud, int32_t C, int32_t k,
                                const float* vis, int32_t ld_vis, int32_t split, int32_t* idx, void* stream) {
    if (!y || !x || !ptr_y || !ptr_x || !idx || n_clouds <= 0 || max_rows_per_cloud <= 0) return MORIG_E_INVALID;
    if (C != KNN_C || k < 1 || k > 8) return MORIG_E_UNSUPPORTED;
    if (ldy < C || ldx < C) return MORIG_E_INVALID;
    if (split && (!vis || ld_vis < 1 || x != y || ptr_x != ptr_y)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_COSINE_KNN, s, 0.0, 0.0);
    const dim3 grid(cdiv(max_rows_per_cloud, 256), n_clouds);
#define MORIG_KNN_CASE(KK) case KK: hipLaunchKernelGGL((cosine_knn_kernel<KK>), grid, dim3(256), 0, s, y, ldy, ptr_y, x, ldx, ptr_x, vis, ld_vis, split, idx); break
    switch (k) {
        MORIG_KNN_CASE(1); MORIG_KNN_CASE(2); MORIG_KNN_CASE(3); MORIG_KNN_CASE(4);
        MORIG_KNN_CASE(5); MORIG_KNN_CASE(6); MORIG_KNN_CASE(7); MORIG_KNN_CASE(8);
    }
#undef MORIG_KNN_CASE
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
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
