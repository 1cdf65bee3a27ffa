// Backward operators of the train-mode path (backward half): what torch.autograd does for the two
// starred blocks of the reference in model.train() -- Seq(Linear, ReLU, BatchNorm1d) over vertices (models/basic_modules.py:31-36)
// and the per-edge MLP of EdgeConv / EdgeConvMotion with BatchNorm statistics over EDGES and max aggregation (:153-155, :179-202)
// -- restated on the operators of the native forward (morig_amd/train_forward.py). Contractions that are GEMM-shaped with the
// rows as the free dimension (dX = dU W) go through morig_gemm; the ones that contract over the rows (dW = dU^T X) are
// morig_gemm_tn below. Everything else is HBM-bound row work:
//
//   morig_bn_backward_stats        per column: sum_r dz and sum_r dz * xhat  (xhat = (y - mean) * rstd), fp64, fixed order
//                                  = dbeta and dgamma of a training-mode BatchNorm1d; with y == NULL a plain column sum (= dbias)
//   morig_bn_relu_backward         du = [y > 0] * gamma * rstd * (dz - sum_dz / n - xhat * sum_dzx / n): the BatchNorm and the ReLU
//                                  in front of it, in one pass (y = ReLU output = BatchNorm input)
//   morig_segmax_affine_arg        morig_segmax_affine that also records WHICH row of the segment won (first on ties; -1 = empty)
//   morig_segmax_bn_backward_stats the same two sums for the BatchNorm that sits in front of a max aggregation, straight from the
//                                  per-segment gradient and the arg-max table: the per-row gradient is one-hot per (segment,
//                                  column) and is never materialised
//   morig_segmax_bn_relu_backward  du[e] for every row e of every segment from that one-hot gradient (dense result: every row
//                                  gets the mean terms of the BatchNorm)
//   morig_edge_scatter_backward    dA[v] = sum of dG over the edges INTO v (contiguous CSR segment, fixed order),
//                                  dB[u] = sum of dG over the edges OUT of u (float atomics: the order varies run to run)
//   morig_gemm_tn                  C[N x K] = A^T B over the rows (fp32 MFMA, 32x32x2: both operands are read row-major, a lane
//                                  takes one float of each), row range split over workgroups, partials summed in a fixed order
#include "common.h"

namespace morig {

constexpr int BS_ROWS = 512;

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ y, int ldy,
                                                             int rows_host, const int* __restrict__ rows_dev, int cols,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             double* __restrict__ part /* [slabs][2][cols] */) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * BS_ROWS;
    double s = 0.0, q = 0.0;
    if (c < cols) {
        const int r1 = min(r0 + BS_ROWS, rows);
        const float m = y ? mean[c] : 0.f, rs = y ? rstd[c] : 0.f;
        for (int r = r0 + rl; r < r1; r += 4) {
            const float g = dz[(size_t)r * ldz + c];
            s += (double)g;
            if (y) q += (double)(g * ((y[(size_t)r * ldy + c] - m) * rs));
        }
    }
    __shared__ double sh[2][4][64];
    sh[0][rl][threadIdx.x & 63] = s; sh[1][rl][threadIdx.x & 63] = q;
    __syncthreads();
    if (rl == 0 && c < cols) {
        const int l = threadIdx.x & 63;
        part[((size_t)blockIdx.y * 2 + 0) * cols + c] = (sh[0][0][l] + sh[0][1][l]) + (sh[0][2][l] + sh[0][3][l]);
        part[((size_t)blockIdx.y * 2 + 1) * cols + c] = (sh[1][0][l] + sh[1][1][l]) + (sh[1][2][l] + sh[1][3][l]);
    }
}

__global__ void bn_bwd_final_kernel(const double* __restrict__ part, int slabs_cap, int rows_host, const int* __restrict__ rows_dev,
                                    int cols, float* __restrict__ sum_dz, float* __restrict__ sum_dzx) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const int slabs = min(slabs_cap, (rows + BS_ROWS - 1) / BS_ROWS);
    double s = 0.0, q = 0.0;
    for (int b = 0; b < slabs; ++b) { s += part[((size_t)b * 2 + 0) * cols + c]; q += part[((size_t)b * 2 + 1) * cols + c]; }
    sum_dz[c] = (float)s;
    if (sum_dzx) sum_dzx[c] = (float)q;
}

__global__ void bn_relu_bwd_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ y, int ldy, int rows_host,
                                   const int* __restrict__ rows_dev, int cols, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ sum_dz,
                                   const float* __restrict__ sum_dzx, float* __restrict__ du, int ldu) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const float inv_n = rows > 0 ? 1.f / (float)rows : 0.f;
    const int64_t total = (int64_t)rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / cols; const int c = (int)(i - r * cols);
        const float yv = y[r * ldy + c];
        const float xh = (yv - mean[c]) * rstd[c];
        const float g = gamma[c] * rstd[c] * (dz[r * ldz + c] - sum_dz[c] * inv_n - xh * (sum_dzx[c] * inv_n));
        du[r * ldu + c] = yv > 0.f ? g : 0.f;
    }
}

// one wave per (segment, 64-column group), as segmax_affine_kernel, plus the winning row
__global__ __launch_bounds__(256) void segmax_arg_kernel(const float* __restrict__ Z, int ldz, const int* __restrict__ rowptr, int n_seg,
                                                         int H, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* __restrict__ out, int ldo, int* __restrict__ arg, int ld_arg) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (v >= n_seg) return;
    const int c = blockIdx.y * 64 + lane;
    if (c >= H) return;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    if (e0 >= e1) { out[(size_t)v * ldo + c] = 0.f; arg[(size_t)v * ld_arg + c] = -1; return; }
    const float s = scale ? scale[c] : 1.f, t = shift ? shift[c] : 0.f;
    float m = Z[(size_t)e0 * ldz + c] * s + t;
    int a = e0;
    for (int e = e0 + 1; e < e1; ++e) {
        const float z = Z[(size_t)e * ldz + c] * s + t;
        if (z > m) { m = z; a = e; }                                 // strict: the first maximum keeps the gradient
    }
    out[(size_t)v * ldo + c] = m;
    arg[(size_t)v * ld_arg + c] = a;
}

// sums over SEGMENTS of the one-hot row gradient: dz[arg[v][c]][c] = dout[v][c]
__global__ __launch_bounds__(256) void segmax_bwd_partial_kernel(const float* __restrict__ dout, int ldd, const int* __restrict__ arg,
                                                                 int ld_arg, const float* __restrict__ Z, int ldz, int n_seg, int cols,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 double* __restrict__ part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int v0 = blockIdx.y * BS_ROWS;
    double s = 0.0, q = 0.0;
    if (c < cols) {
        const int v1 = min(v0 + BS_ROWS, n_seg);
        const float m = mean[c], rs = rstd[c];
        for (int v = v0 + rl; v < v1; v += 4) {
            const int a = arg[(size_t)v * ld_arg + c];
            if (a < 0) continue;
            const float g = dout[(size_t)v * ldd + c];
            s += (double)g;
            q += (double)(g * ((Z[(size_t)a * ldz + c] - m) * rs));
        }
    }
    __shared__ double sh[2][4][64];
    sh[0][rl][threadIdx.x & 63] = s; sh[1][rl][threadIdx.x & 63] = q;
    __syncthreads();
    if (rl == 0 && c < cols) {
        const int l = threadIdx.x & 63;
        part[((size_t)blockIdx.y * 2 + 0) * cols + c] = (sh[0][0][l] + sh[0][1][l]) + (sh[0][2][l] + sh[0][3][l]);
        part[((size_t)blockIdx.y * 2 + 1) * cols + c] = (sh[1][0][l] + sh[1][1][l]) + (sh[1][2][l] + sh[1][3][l]);
    }
}

__global__ void segmax_bn_relu_bwd_kernel(const float* __restrict__ dout, int ldd, const int* __restrict__ arg, int ld_arg,
                                          const float* __restrict__ Z, int ldz, const int* __restrict__ rowptr, int n_seg,
                                          const int* __restrict__ seg_of_row, int cols, const float* __restrict__ mean,
                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                          const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx, int relu,
                                          float* __restrict__ du, int ldu) {
    const int rows = rowptr[n_seg];
    const float inv_n = rows > 0 ? 1.f / (float)rows : 0.f;
    const int64_t total = (int64_t)rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t e = i / cols; const int c = (int)(i - e * cols);
        const int v = seg_of_row[e];
        const float zv = Z[e * ldz + c];
        const float xh = (zv - mean[c]) * rstd[c];
        const float dzv = arg[(size_t)v * ld_arg + c] == (int)e ? dout[(size_t)v * ldd + c] : 0.f;
        const float g = gamma[c] * rstd[c] * (dzv - sum_dz[c] * inv_n - xh * (sum_dzx[c] * inv_n));
        du[e * ldu + c] = (!relu || zv > 0.f) ? g : 0.f;
    }
}

// dA[v] = sum over the rows of segment v (fixed order); dB[src(e)] += dG[e] (atomics). One wave per (segment, 64 columns).
__global__ __launch_bounds__(256) void edge_scatter_bwd_kernel(const float* __restrict__ dG, int ldg, const int* __restrict__ rowptr,
                                                               const int* __restrict__ srcS, int n_nodes, int H,
                                                               float* __restrict__ dA, int lda, float* __restrict__ dB, int ldb) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (v >= n_nodes) return;
    const int c = blockIdx.y * 64 + lane;
    if (c >= H) return;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    float s = 0.f;
    for (int e = e0; e < e1; ++e) {
        const float g = dG[(size_t)e * ldg + c];
        s += g;
        atomicAdd(dB + (size_t)srcS[e] * ldb + c, g);
    }
    dA[(size_t)v * lda + c] = s;
}

// ---- C = A^T B over the rows -----------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32: A operand lane l = A'[i = l & 31][k = l >> 5], B operand lane l = B'[k = l >> 5][j = l & 31]. With
// A' = A^T (i = a column n of A, k = a row r) both operands are 32 consecutive floats of a row: the row-major tiles go into LDS
// as they are and every fragment is one conflict-free ds_read_b32. Workgroup tile 128 (n) x 128 (k), 4 waves x (2 x 2) MFMA
// tiles, 32 rows staged at a time; blockIdx.z walks its share of the row range and writes one partial tile.
constexpr int TN_T = 128, TN_R = 32;
typedef float tn_f32x16 __attribute__((ext_vector_type(16)));
typedef float tn_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                      int rows_host, const int* __restrict__ rows_dev, int N, int K, int chunk_rows,
                                                      float* __restrict__ part /* [chunks][N][K] */) {
    __shared__ __attribute__((aligned(16))) float sA[TN_R][TN_T + 4], sB[TN_R][TN_T + 4];
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int n0 = blockIdx.x * TN_T, k0 = blockIdx.y * TN_T;
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
    const int r_begin = blockIdx.z * chunk_rows, r_end = min(r_begin + chunk_rows, rows);
    tn_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int lr = tid >> 5, lc = (tid & 31) * 4;                  // loader: 8 rows x 32 float4 per pass, 4 passes per operand
    const bool a_vec = (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
    const bool b_vec = (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
    for (int r0 = r_begin; r0 < r_end; r0 += TN_R) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < TN_R / 8; ++p) {
            const int r = r0 + p * 8 + lr;
            tn_f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (r < r_end) {
                const float* pa = A + (size_t)r * lda + n0 + lc;
                const float* pb = B + (size_t)r * ldb + k0 + lc;
                if (a_vec && n0 + lc + 4 <= N) va = *reinterpret_cast<const tn_f32x4*>(pa);
                else { for (int q = 0; q < 4; ++q) if (n0 + lc + q < N) va[q] = pa[q]; }
                if (b_vec && k0 + lc + 4 <= K) vb = *reinterpret_cast<const tn_f32x4*>(pb);
                else { for (int q = 0; q < 4; ++q) if (k0 + lc + q < K) vb[q] = pb[q]; }
            }
            *reinterpret_cast<tn_f32x4*>(&sA[p * 8 + lr][lc]) = va;
            *reinterpret_cast<tn_f32x4*>(&sB[p * 8 + lr][lc]) = vb;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < TN_R; kk += 2) {
            float fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = sA[kk + hi][wn + a * 32 + l31];
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = sB[kk + hi][wk + b * 32 + l31];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }
    float* o = part + (size_t)blockIdx.z * N * K;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, k = k0 + wk + b * 32 + l31;
                if (n < N && k < K) o[(size_t)n * K + k] = acc[a][b][r];
            }
}

__global__ void gemm_tn_reduce_kernel(const float* __restrict__ part, int chunks, int N, int K, float* __restrict__ out, int ldo) {
    const int64_t total = (int64_t)N * K;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        float s = 0.f;
        for (int c = 0; c < chunks; ++c) s += part[(size_t)c * total + i];
        const int64_t n = i / K; const int k = (int)(i - n * K);
        out[n * ldo + k] = s;
    }
}

static int tn_chunks(int rows, int N, int K) {
    const int tiles = cdiv(N, TN_T) * cdiv(K, TN_T);
    int chunks = cdiv(1024, tiles);                                   // ~4 workgroups per CU
    const int max_chunks = cdiv(rows > 0 ? rows : 1, 4 * TN_R);       // at least 128 rows per chunk
    if (chunks > max_chunks) chunks = max_chunks;
    return chunks < 1 ? 1 : chunks;
}

}  // namespace morig

using namespace morig;

extern "C" int morig_bn_backward_stats(const float* dz, int32_t ldz, const float* y, int32_t ldy, int32_t rows, const int32_t* rows_dev,
                                       int32_t cols, const float* mean, const float* rstd, double* workspace, int64_t workspace_doubles,
                                       float* sum_dz, float* sum_dzx, void* stream) {
    if (!dz || !workspace || !sum_dz || rows < 0 || cols <= 0 || ldz < cols) return MORIG_E_INVALID;
    if (y && (!mean || !rstd || !sum_dzx || ldy < cols)) return MORIG_E_INVALID;
    const int slabs = cdiv(rows > 0 ? rows : 1, BS_ROWS);
    if (workspace_doubles < (int64_t)slabs * 2 * cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 8.0 * rows * (double)cols);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dz, ldz, y, ldy, rows, rows_dev, cols, mean, rstd,
                       workspace);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, workspace, slabs, rows, rows_dev, cols, sum_dz,
                       y ? sum_dzx : nullptr);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_bn_relu_backward(const float* dz, int32_t ldz, const float* y, int32_t ldy, int32_t rows, const int32_t* rows_dev,
                                      int32_t cols, const float* mean, const float* rstd, const float* gamma, const float* sum_dz,
                                      const float* sum_dzx, float* du, int32_t ldu, void* stream) {
    if (!dz || !y || !mean || !rstd || !gamma || !sum_dz || !sum_dzx || !du) return MORIG_E_INVALID;
    if (rows < 0 || cols <= 0 || ldz < cols || ldy < cols || ldu < cols) return MORIG_E_INVALID;
    if (rows == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = ((int64_t)rows * cols + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(K_MISC, s, 0.0, 12.0 * rows * (double)cols);
    hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3((int)blocks), dim3(256), 0, s, dz, ldz, y, ldy, rows, rows_dev, cols, mean, rstd, gamma,
                       sum_dz, sum_dzx, du, ldu);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_segmax_affine_arg(const float* Z, int32_t ldz, const int32_t* rowptr, int32_t n_segments, int32_t H,
                                       const float* scale, const float* shift, float* out, int32_t ldo, int32_t* arg, int32_t ld_arg,
                                       void* stream) {
    if (!Z || !rowptr || !out || !arg || n_segments <= 0 || H <= 0 || ldz < H || ldo < H || ld_arg < H) return MORIG_E_INVALID;
    if ((scale == nullptr) != (shift == nullptr)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 0.0);
    hipLaunchKernelGGL(segmax_arg_kernel, dim3(cdiv(n_segments, 4), cdiv(H, 64)), dim3(256), 0, s, Z, ldz, rowptr, n_segments, H, scale,
                       shift, out, ldo, arg, ld_arg);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_segmax_bn_backward_stats(const float* dout, int32_t ldd, const int32_t* arg, int32_t ld_arg, const float* Z,
                                              int32_t ldz, int32_t n_segments, int32_t cols, const float* mean, const float* rstd,
                                              double* workspace, int64_t workspace_doubles, float* sum_dz, float* sum_dzx, void* stream) {
    if (!dout || !arg || !Z || !mean || !rstd || !workspace || !sum_dz || !sum_dzx) return MORIG_E_INVALID;
    if (n_segments <= 0 || cols <= 0 || ldd < cols || ld_arg < cols || ldz < cols) return MORIG_E_INVALID;
    const int slabs = cdiv(n_segments, BS_ROWS);
    if (workspace_doubles < (int64_t)slabs * 2 * cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 12.0 * n_segments * (double)cols);
    hipLaunchKernelGGL(segmax_bwd_partial_kernel, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dout, ldd, arg, ld_arg, Z, ldz, n_segments,
                       cols, mean, rstd, workspace);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, workspace, slabs, n_segments, nullptr, cols, sum_dz,
                       sum_dzx);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_segmax_bn_relu_backward(const float* dout, int32_t ldd, const int32_t* arg, int32_t ld_arg, const float* Z,
                                             int32_t ldz, const int32_t* rowptr, int32_t n_segments, const int32_t* seg_of_row,
                                             int32_t row_capacity, int32_t cols, const float* mean, const float* rstd, const float* gamma,
                                             const float* sum_dz, const float* sum_dzx, int32_t relu, float* du, int32_t ldu,
                                             void* stream) {
    if (!dout || !arg || !Z || !rowptr || !seg_of_row || !mean || !rstd || !gamma || !sum_dz || !sum_dzx || !du) return MORIG_E_INVALID;
    if (n_segments <= 0 || row_capacity <= 0 || cols <= 0 || ldd < cols || ld_arg < cols || ldz < cols || ldu < cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = ((int64_t)row_capacity * cols + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(K_MISC, s, 0.0, 16.0 * row_capacity * (double)cols);
    hipLaunchKernelGGL(segmax_bn_relu_bwd_kernel, dim3((int)blocks), dim3(256), 0, s, dout, ldd, arg, ld_arg, Z, ldz, rowptr, n_segments,
                       seg_of_row, cols, mean, rstd, gamma, sum_dz, sum_dzx, relu, du, ldu);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_edge_scatter_backward(const float* dG, int32_t ldg, const int32_t* rowptr, const int32_t* src_sorted, int32_t n_nodes,
                                           int32_t n_src_nodes, int32_t H, float* dA, int32_t lda, float* dB, int32_t ldb, void* stream) {
    if (!dG || !rowptr || !src_sorted || !dA || !dB || n_nodes <= 0 || n_src_nodes <= 0 || H <= 0) return MORIG_E_INVALID;
    if (ldg < H || lda < H || ldb < H) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 0.0);
    MORIG_HIP_TRY(hipMemset2DAsync(dB, (size_t)ldb * sizeof(float), 0, (size_t)H * sizeof(float), (size_t)n_src_nodes, s));
    hipLaunchKernelGGL(edge_scatter_bwd_kernel, dim3(cdiv(n_nodes, 4), cdiv(H, 64)), dim3(256), 0, s, dG, ldg, rowptr, src_sorted, n_nodes, H,
                       dA, lda, dB, ldb);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int64_t morig_gemm_tn_workspace(int32_t rows, int32_t N, int32_t K) {
    if (rows < 0 || N <= 0 || K <= 0) return 0;
    return (int64_t)tn_chunks(rows, N, K) * N * K;
}

extern "C" int morig_gemm_tn(const float* A, int32_t lda, const float* B, int32_t ldb, int32_t rows, const int32_t* rows_dev, int32_t N,
                             int32_t K, float* workspace, int64_t workspace_floats, float* out, int32_t ldo, void* stream) {
    if (!A || !B || !workspace || !out || rows < 0 || N <= 0 || K <= 0 || lda < N || ldb < K || ldo < K) return MORIG_E_INVALID;
    const int chunks = tn_chunks(rows, N, K);
    if (workspace_floats < (int64_t)chunks * N * K) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int chunk_rows = cdiv(cdiv(rows > 0 ? rows : 1, chunks), TN_R) * TN_R;
    ProfScope ps(K_MISC, s, 2.0 * rows * (double)N * K, 4.0 * rows * ((double)N + K));
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(cdiv(N, TN_T), cdiv(K, TN_T), chunks), dim3(256), 0, s, A, lda, B, ldb, rows, rows_dev, N, K,
                       chunk_rows, workspace);
    MORIG_LAUNCH_CHECK();
    int64_t blocks = ((int64_t)N * K + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((int)blocks), dim3(256), 0, s, workspace, chunks, N, K, out, ldo);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}
