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


// partial sums of one (row slab, 64-column group)
__global__ __launch_bounds__(256) void col_stats_partial_kernel(const float* __restrict__ x, int ldx, int rows_host,
                                                                const int* __restrict__ rows_dev, int cols, int slab_rows,
                                                                double* __restrict__ part /* [slabs][2][cols] */) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int c0 = blockIdx.x * 64;
    const int quads = min(16, (cols - c0 + 3) >> 2);
    const int RL = 256 / quads;
    const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
    const int c = c0 + q * 4;
    const int r0 = blockIdx.y * slab_rows, r1 = min(r0 + slab_rows, rows);
    double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (rl < RL) {
        if ((ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && c + 4 <= cols) {
#pragma unroll 4
            for (int r = r0 + rl; r < r1; r += RL) {
                const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
                const double d[4] = {(double)v.x, (double)v.y, (double)v.z, (double)v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[0][j] += d[j]; acc[1][j] += d[j] * d[j]; }
            }
        } else {
            for (int r = r0 + rl; r < r1; r += RL)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < cols) { const double d = (double)x[(size_t)r * ldx + c + j]; acc[0][j] += d; acc[1][j] += d * d; }
        }
    }
    stats_block_store(acc, quads, RL, q, rl, c0, cols, blockIdx.y, part);
}

__global__ __launch_bounds__(256) void col_stats_final_kernel(const double* __restrict__ part, int slabs, int rows_host,
                                                              const int* __restrict__ rows_dev, int cols, float* __restrict__ mean,
                                                              float* __restrict__ var, float* __restrict__ count) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int c = blockIdx.x * STATS_FC + (threadIdx.x % STATS_FC);
    if (blockIdx.x == 0 && threadIdx.x == 0 && count) *count = (float)rows;
    double s, q;
    stats_final_sums(part, slabs, cols, c, s, q);
    if (threadIdx.x >= STATS_FC || c >= cols) return;
    const double n = rows > 0 ? (double)rows : 1.0;
    const double m = s / n;
    double v = q / n - m * m;                          // fp64: the cancellation costs ~1e-16 relative, far below fp32
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m; var[c] = (float)v;
}

template <int V>
__global__ void col_affine_kernel(const float* x, int ldx, int rows_host, const int* __restrict__ rows_dev, int cols,
                                  const float* __restrict__ scale, const float* __restrict__ shift, float* out, int ldo) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int qn = cols / V;
    const int64_t total = (int64_t)rows * qn;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / qn; const int c = (int)(i - r * qn) * V;
        VecF<V> v = ldv<V>(x + r * ldx + c);                   // (out may be x: every thread reads its elements before it writes them)
        const VecF<V> sc = ldv<V>(scale + c), sh = ldv<V>(shift + c);
#pragma unroll
        for (int j = 0; j < V; ++j) v.v[j] = v.v[j] * sc.v[j] + sh.v[j];
        stv<V>(out + r * ldo + c, v);
    }
}

// One workgroup per (64 columns, slab of rows), threads as in col_stats_partial_kernel; with `part` it also leaves the slab's fp64
// column sums of z and z^2: the BatchNorm statistics of the first edge layer come from the pass that writes it.
template <int V>
__global__ __launch_bounds__(256) void edge_gather_relu_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                        const int* __restrict__ rowptr, int n_nodes, const int* __restrict__ srcS,
                                        const int* __restrict__ dstS, int H, int slab_rows, float* __restrict__ Z, int ldz,
                                        double* __restrict__ part /* NULL or [slabs][2][H] */) {
    const int E = rowptr[n_nodes];
    const int c0 = blockIdx.x * 64;
    const int quads = min(16, (H - c0 + 3) >> 2);
    const int RL = 256 / quads;
    const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
    const int c = c0 + q * 4;
    const int r0 = blockIdx.y * slab_rows, r1 = min(r0 + slab_rows, E);
    double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (rl < RL) {
        int e = r0 + rl;
        if (V == 4) {
            // four rows in flight: their edge ids first, then the eight gathered 16-byte pieces (one row at a time is a chain of two
            // dependent latencies per 32 bytes: 3.6 ms per training step for 9.1 GB written, twice the HBM time)
            for (; e + 3 * RL < r1; e += 4 * RL) {
                int di[4], si[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { di[u] = dstS[e + u * RL]; si[u] = srcS[e + u * RL]; }
                float4 a4[4], b4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a4[u] = *reinterpret_cast<const float4*>(A + (size_t)di[u] * lda + c);
                    b4[u] = *reinterpret_cast<const float4*>(B + (size_t)si[u] * ldb + c);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float z[4] = {a4[u].x + b4[u].x, a4[u].y + b4[u].y, a4[u].z + b4[u].z, a4[u].w + b4[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        z[j] = z[j] > 0.f ? z[j] : 0.f;
                        const double d = (double)z[j];
                        acc[0][j] += d; acc[1][j] += d * d;
                    }
                    *reinterpret_cast<float4*>(Z + (size_t)(e + u * RL) * ldz + c) = make_float4(z[0], z[1], z[2], z[3]);
                }
            }
        }
        for (; e < r1; e += RL) {
            const float* pa = A + (size_t)dstS[e] * lda + c;
            const float* pb = B + (size_t)srcS[e] * ldb + c;
            float z[4];
            if (V == 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(pa), b4 = *reinterpret_cast<const float4*>(pb);
                z[0] = a4.x + b4.x; z[1] = a4.y + b4.y; z[2] = a4.z + b4.z; z[3] = a4.w + b4.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = c + j < H ? pa[j] + pb[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                z[j] = z[j] > 0.f ? z[j] : 0.f;
                const double d = (double)z[j];
                acc[0][j] += d; acc[1][j] += d * d;
            }
            if (V == 4) *reinterpret_cast<float4*>(Z + (size_t)e * ldz + c) = make_float4(z[0], z[1], z[2], z[3]);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < H) Z[(size_t)e * ldz + c + j] = z[j];
        }
    }
    if (part) stats_block_store(acc, quads, RL, q, rl, c0, H, blockIdx.y, part);
}

// one thread per (segment, V columns): the threads of a segment read consecutive columns of a row, rows walked in order
template <int V>
__global__ __launch_bounds__(256) void segmax_affine_kernel(const float* __restrict__ Z, int ldz, const int* __restrict__ rowptr,
                                                            int n_seg, int H, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ out, int ldo) {
    const int qn = H / V;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n_seg * qn) return;
    const int v = (int)(t / qn), c = (int)(t - (int64_t)v * qn) * V;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    VecF<V> m;
    if (e0 >= e1) {                                                   // torch_scatter 'max' fill of an empty segment
#pragma unroll
        for (int j = 0; j < V; ++j) m.v[j] = 0.f;
        stv<V>(out + (size_t)v * ldo + c, m);
        return;
    }
    VecF<V> s, sh;
#pragma unroll
    for (int j = 0; j < V; ++j) { s.v[j] = scale ? scale[c + j] : 1.f; sh.v[j] = shift ? shift[c + j] : 0.f; m.v[j] = -INFINITY; }
    for (int e = e0; e < e1; ++e) {
        const VecF<V> z = ldv<V>(Z + (size_t)e * ldz + c);
#pragma unroll
        for (int j = 0; j < V; ++j) m.v[j] = fmaxf(m.v[j], z.v[j] * s.v[j] + sh.v[j]);
    }
    stv<V>(out + (size_t)v * ldo + c, m);
}

}  // namespace morig

using namespace morig;

extern "C" int morig_col_stats(const float* x, int32_t ldx, int32_t rows, const int32_t* rows_dev, int32_t cols, double* workspace,
                               int64_t workspace_doubles, float* mean, float* var, float* count, void* stream) {
    if (!x || !workspace || !mean || !var || rows < 0 || cols <= 0 || ldx < cols) return MORIG_E_INVALID;
    const int slab_rows = stats_slab_rows(rows, cols);                    // `rows` is the capacity when rows_dev is given
    const int slabs = cdiv(rows > 0 ? rows : 1, slab_rows);
    if (workspace_doubles < (int64_t)slabs * 2 * cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 4.0 * rows * (double)cols);
    hipLaunchKernelGGL(col_stats_partial_kernel, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, x, ldx, rows, rows_dev, cols, slab_rows,
                       workspace);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(col_stats_final_kernel, dim3(cdiv(cols, STATS_FC)), dim3(256), 0, s, workspace, slabs, rows, rows_dev, cols, mean, var, count);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_col_affine(float* x, int32_t ldx, int32_t rows, const int32_t* rows_dev, int32_t cols, const float* scale,
                                const float* shift, float* out, int32_t ldo, void* stream) {
    if (!x || !scale || !shift || rows < 0 || cols <= 0 || ldx < cols || (out && ldo < cols)) return MORIG_E_INVALID;
    if (rows == 0) return MORIG_OK;
    if (!out) { out = x; ldo = ldx; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool v4 = (cols & 3) == 0 && vec4_ptr(x, ldx) && vec4_ptr(out, ldo) && vec4_ptr(scale, 0) && vec4_ptr(shift, 0);
    int64_t blocks = ((int64_t)rows * (cols / (v4 ? 4 : 1)) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    ProfScope ps(K_MISC, s, 0.0, 8.0 * rows * (double)cols);
    if (v4) hipLaunchKernelGGL(col_affine_kernel<4>, dim3((int)blocks), dim3(256), 0, s, x, ldx, rows, rows_dev, cols, scale, shift, out, ldo);
    else hipLaunchKernelGGL(col_affine_kernel<1>, dim3((int)blocks), dim3(256), 0, s, x, ldx, rows, rows_dev, cols, scale, shift, out, ldo);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

// ---- BatchNorm1d in training mode, everything that follows the column statistics in ONE launch (torch.nn.functional.batch_norm
// semantics; models/basic_modules.py:31-36 has momentum = 0.1): the batch affine s = gamma / sqrt(var + eps), t = beta - mean s
// (biased variance normalises), rstd for the backward, and the running-buffer update with the UNBIASED variance. As torch
// elementwise calls this was a dozen launches per BatchNorm layer, 182 layers per JointNetMotion step. ----
__global__ void bn_finalize_kernel(const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, long long* __restrict__ nbt,
                                   float* __restrict__ s, float* __restrict__ t, float* __restrict__ rstd, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && nbt) nbt[0] += 1;
    if (i >= n) return;
    const float v = var[i], m = mean[i];
    const float sd = __fsqrt_rn(v + eps);
    const float sc = __fdiv_rn(gamma ? gamma[i] : 1.f, sd);
    s[i] = sc;
    t[i] = (beta ? beta[i] : 0.f) - m * sc;
    if (rstd) rstd[i] = __fdiv_rn(1.f, sd);
    if (running_mean && running_var) {
        const float c = count[0];
        const float unbiased = v * (c / fmaxf(c - 1.f, 1.f));
        running_mean[i] = running_mean[i] * (1.f - momentum) + m * momentum;
        running_var[i] = running_var[i] * (1.f - momentum) + unbiased * momentum;
    }
}

extern "C" int morig_bn_finalize(const float* mean, const float* var, const float* count, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                 float* s, float* t, float* rstd, int32_t n, void* stream) {
    if (!mean || !var || !s || !t || n <= 0 || (running_mean == nullptr) != (running_var == nullptr)) return MORIG_E_INVALID;
    if (running_mean && !count) return MORIG_E_INVALID;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, st, 0.0, 32.0 * n);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, mean, var, count, gamma, beta, eps, momentum, running_mean,
                       running_var, reinterpret_cast<long long*>(num_batches_tracked), s, t, rstd, n);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_edge_gather_relu(const float* A, int32_t lda, const float* B, int32_t ldb, const int32_t* rowptr, int32_t n_nodes,
                                      const int32_t* src_sorted, const int32_t* dst_sorted, int32_t edge_capacity, int32_t H,
                                      float* Z, int32_t ldz, double* workspace, int64_t workspace_doubles, float* mean, float* var,
                                      float* count, void* stream) {
    if (!A || !B || !rowptr || !src_sorted || !dst_sorted || !Z || n_nodes <= 0 || edge_capacity <= 0 || H <= 0) return MORIG_E_INVALID;
    if (lda < H || ldb < H || ldz < H) return MORIG_E_INVALID;
    const int slab_rows = stats_slab_rows(edge_capacity, H);
    const int slabs = cdiv(edge_capacity, slab_rows);
    if (mean && (!var || !workspace || workspace_doubles < (int64_t)slabs * 2 * H)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool v4 = (H & 3) == 0 && vec4_ptr(A, lda) && vec4_ptr(B, ldb) && vec4_ptr(Z, ldz);
    ProfScope ps(K_MISC, s, 0.0, 12.0 * edge_capacity * (double)H);
    double* part = mean ? workspace : nullptr;
    if (v4) hipLaunchKernelGGL(edge_gather_relu_kernel<4>, dim3(cdiv(H, 64), slabs), dim3(256), 0, s, A, lda, B, ldb, rowptr, n_nodes,
                               src_sorted, dst_sorted, H, slab_rows, Z, ldz, part);
    else hipLaunchKernelGGL(edge_gather_relu_kernel<1>, dim3(cdiv(H, 64), slabs), dim3(256), 0, s, A, lda, B, ldb, rowptr, n_nodes,
                            src_sorted, dst_sorted, H, slab_rows, Z, ldz, part);
    MORIG_LAUNCH_CHECK();
    if (mean) {
        hipLaunchKernelGGL(col_stats_final_kernel, dim3(cdiv(H, STATS_FC)), dim3(256), 0, s, workspace, slabs, edge_capacity, rowptr + n_nodes, H,
                           mean, var, count);
        MORIG_LAUNCH_CHECK();
    }
    return MORIG_OK;
}

extern "C" int morig_segmax_affine(const float* Z, int32_t ldz, const int32_t* rowptr, int32_t n_segments, int32_t H, const float* scale,
                                   const float* shift, float* out, int32_t ldo, void* stream) {
    if (!Z || !rowptr || !out || n_segments <= 0 || H <= 0 || ldz < H || ldo < H) return MORIG_E_INVALID;
    if ((scale == nullptr) != (shift == nullptr)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 0.0);
    const bool v4 = (H & 3) == 0 && vec4_ptr(Z, ldz) && vec4_ptr(out, ldo);
    const int blocks = cdiv((long)n_segments * (H / (v4 ? 4 : 1)), 256);
    if (v4) hipLaunchKernelGGL(segmax_affine_kernel<4>, dim3(blocks), dim3(256), 0, s, Z, ldz, rowptr, n_segments, H, scale, shift, out, ldo);
    else hipLaunchKernelGGL(segmax_affine_kernel<1>, dim3(blocks), dim3(256), 0, s, Z, ldz, rowptr, n_segments, H, scale, shift, out, ldo);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}
