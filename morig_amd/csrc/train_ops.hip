// Train-mode forward support (forward half): in model.train() every BatchNorm1d of the reference
// normalises with the statistics of the CURRENT batch -- over vertices for the dense MLPs, over EDGES inside the per-edge MLPs
// of EdgeConv / EdgeConvMotion (models/basic_modules.py:31-36, 153-155, 192-195; training/train_rig.py:136-195) -- so a layer
// cannot be fused across its BatchNorm: the batch mean / variance of a layer's ReLU output must be known before anything
// consumes it. The contractions stay on the MFMA kernels (morig_gemm, morig_edge_hidden); these are the HBM-bound pieces
// between them:
//   morig_col_stats        per-column mean and biased variance of a row-major matrix, fp64 accumulation, two passes in a fixed
//                          order (deterministic; no atomics). The row count may live on the device (E' = rowptr[n]).
//   morig_col_affine       x <- s * x + t per column, in place (the BatchNorm once its statistics are known)
//   morig_edge_gather_relu Z1[e] = relu(A[dst_e] + B[src_e]): the first edge ReLU, materialised for its statistics
//   morig_segmax_affine    out[v] = max over the rows of segment v of (s * Z + t): max-aggregation behind the LAST BatchNorm
//                          of an edge MLP (and scatter_max over a mesh's vertices with s = 1, t = 0); s < 0 handled exactly
#include "common.h"

namespace morig {

constexpr int CS_ROWS = 512;          // rows per partial block of col_stats

// partial sums of one (row slab, 64-column group): 256 threads = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void col_stats_partial_kernel(const float* __restrict__ x, int ldx, int rows_host,
                                                                const int* __restrict__ rows_dev, int cols,
                                                                double* __restrict__ part /* [slabs][2][cols] */) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * CS_ROWS;
    double s = 0.0, q = 0.0;
    if (c < cols) {
        const int r1 = min(r0 + CS_ROWS, rows);
        for (int r = r0 + rl; r < r1; r += 4) {
            const double v = (double)x[(size_t)r * ldx + c];
            s += v; q += v * v;
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

__global__ void col_stats_final_kernel(const double* __restrict__ part, int slabs_cap, int rows_host, const int* __restrict__ rows_dev,
                                       int cols, float* __restrict__ mean, float* __restrict__ var, float* __restrict__ count) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && count) *count = (float)rows;
    if (c >= cols) return;
    const int slabs = min(slabs_cap, (rows + CS_ROWS - 1) / CS_ROWS);
    double s = 0.0, q = 0.0;
    for (int b = 0; b < slabs; ++b) { s += part[((size_t)b * 2 + 0) * cols + c]; q += part[((size_t)b * 2 + 1) * cols + c]; }
    const double n = rows > 0 ? (double)rows : 1.0;
    const double m = s / n;
    double v = q / n - m * m;                          // fp64: the cancellation costs ~1e-16 relative, far below fp32
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m; var[c] = (float)v;
}

__global__ void col_affine_kernel(float* __restrict__ x, int ldx, int rows_host, const int* __restrict__ rows_dev, int cols,
                                  const float* __restrict__ scale, const float* __restrict__ shift) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int64_t total = (int64_t)rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / cols; const int c = (int)(i - r * cols);
        float* p = x + r * ldx + c;
        *p = *p * scale[c] + shift[c];
    }
}

__global__ void edge_gather_relu_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                        const int* __restrict__ rowptr, int n_nodes, const int* __restrict__ srcS,
                                        const int* __restrict__ dstS, int H, float* __restrict__ Z, int ldz) {
    const int E = rowptr[n_nodes];
    const int64_t total = (int64_t)E * H;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t e = i / H; const int c = (int)(i - e * H);
        const float v = A[(size_t)dstS[e] * lda + c] + B[(size_t)srcS[e] * ldb + c];
        Z[e * ldz + c] = v > 0.f ? v : 0.f;
    }
}

// one wave per (segment, 64-column group): lanes = columns, rows walked in order
__global__ __launch_bounds__(256) void segmax_affine_kernel(const float* __restrict__ Z, int ldz, const int* __restrict__ rowptr,
                                                            int n_seg, int H, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (v >= n_seg) return;
    const int c = blockIdx.y * 64 + lane;
    if (c >= H) return;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    if (e0 >= e1) { out[(size_t)v * ldo + c] = 0.f; return; }       // torch_scatter 'max' fill of an empty segment
    const float s = scale ? scale[c] : 1.f, t = shift ? shift[c] : 0.f;
    float m = -INFINITY;
    for (int e = e0; e < e1; ++e) m = fmaxf(m, Z[(size_t)e * ldz + c] * s + t);
    out[(size_t)v * ldo + c] = m;
}

}  // namespace morig

using namespace morig;

extern "C" int morig_col_stats(const float* x, int32_t ldx, int32_t rows, const int32_t* rows_dev, int32_t cols, double* workspace,
                               int64_t workspace_doubles, float* mean, float* var, float* count, void* stream) {
    if (!x || !workspace || !mean || !var || rows < 0 || cols <= 0 || ldx < cols) return MORIG_E_INVALID;
    const int slabs = cdiv(rows > 0 ? rows : 1, CS_ROWS);           // `rows` is the capacity when rows_dev is given
    if (workspace_doubles < (int64_t)slabs * 2 * cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 4.0 * rows * (double)cols);
    hipLaunchKernelGGL(col_stats_partial_kernel, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, x, ldx, rows, rows_dev, cols, workspace);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_stats_final_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, workspace, slabs, rows, rows_dev, cols, mean, var, count);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_col_affine(float* x, int32_t ldx, int32_t rows, const int32_t* rows_dev, int32_t cols, const float* scale,
                                const float* shift, void* stream) {
    if (!x || !scale || !shift || rows < 0 || cols <= 0 || ldx < cols) return MORIG_E_INVALID;
    if (rows == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = ((int64_t)rows * cols + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ProfScope ps(K_MISC, s, 0.0, 8.0 * rows * (double)cols);
    hipLaunchKernelGGL(col_affine_kernel, dim3((int)blocks), dim3(256), 0, s, x, ldx, rows, rows_dev, cols, scale, shift);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_edge_gather_relu(const float* A, int32_t lda, const float* B, int32_t ldb, const int32_t* rowptr, int32_t n_nodes,
                                      const int32_t* src_sorted, const int32_t* dst_sorted, int32_t edge_capacity, int32_t H,
                                      float* Z, int32_t ldz, void* stream) {
    if (!A || !B || !rowptr || !src_sorted || !dst_sorted || !Z || n_nodes <= 0 || edge_capacity <= 0 || H <= 0) return MORIG_E_INVALID;
    if (lda < H || ldb < H || ldz < H) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = ((int64_t)edge_capacity * H + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(K_MISC, s, 0.0, 12.0 * edge_capacity * (double)H);
    hipLaunchKernelGGL(edge_gather_relu_kernel, dim3((int)blocks), dim3(256), 0, s, A, lda, B, ldb, rowptr, n_nodes, src_sorted, dst_sorted,
                       H, Z, ldz);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_segmax_affine(const float* Z, int32_t ldz, const int32_t* rowptr, int32_t n_segments, int32_t H, const float* scale,
                                   const float* shift, float* out, int32_t ldo, void* stream) {
    if (!Z || !rowptr || !out || n_segments <= 0 || H <= 0 || ldz < H || ldo < H) return MORIG_E_INVALID;
    if ((scale == nullptr) != (shift == nullptr)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 0.0);
    hipLaunchKernelGGL(segmax_affine_kernel, dim3(cdiv(n_segments, 4), cdiv(H, 64)), dim3(256), 0, s, Z, ldz, rowptr, n_segments, H, scale,
                       shift, out, ldo);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}
