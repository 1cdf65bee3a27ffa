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
#include <stdlib.h>

namespace morig {

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ y, int ldy,
                                                             int rows_host, const int* __restrict__ rows_dev, int cols, int slab_rows,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             double* __restrict__ part /* [slabs][2][cols] */,
                                                             const int* __restrict__ live /* NULL, or: entries < 0 drop dz */, int ld_live) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int c0 = blockIdx.x * 64;
    const int quads = min(16, (cols - c0 + 3) >> 2);
    const int RL = 256 / quads;
    const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
    const int c = c0 + q * 4;
    const int r0 = blockIdx.y * slab_rows, r1 = min(r0 + slab_rows, rows);
    double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (rl < RL) {
        float m[4], rs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[j] = (y && c + j < cols) ? mean[c + j] : 0.f; rs[j] = (y && c + j < cols) ? rstd[c + j] : 0.f; }
        const bool vec = (ldz & 3) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0 && c + 4 <= cols &&
                         (!y || ((ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0));
        if (vec) {
#pragma unroll 4
            for (int r = r0 + rl; r < r1; r += RL) {
                const float4 g4 = *reinterpret_cast<const float4*>(dz + (size_t)r * ldz + c);
                float g[4] = {g4.x, g4.y, g4.z, g4.w};
                if (live) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (live[(size_t)r * ld_live + c + j] < 0) g[j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0][j] += (double)g[j];
                if (y) {
                    const float4 y4 = *reinterpret_cast<const float4*>(y + (size_t)r * ldy + c);
                    const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[1][j] += (double)(g[j] * ((yv[j] - m[j]) * rs[j]));
                }
            }
        } else {
            for (int r = r0 + rl; r < r1; r += RL)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < cols) {
                        const float g = (live && live[(size_t)r * ld_live + c + j] < 0) ? 0.f : dz[(size_t)r * ldz + c + j];
                        acc[0][j] += (double)g;
                        if (y) acc[1][j] += (double)(g * ((y[(size_t)r * ldy + c + j] - m[j]) * rs[j]));
                    }
        }
    }
    stats_block_store(acc, quads, RL, q, rl, c0, cols, blockIdx.y, part);
}

// second pass: STATS_FC columns x 64 slab lanes per block, fixed order (common.h)
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const double* __restrict__ part, int slabs, int cols,
                                                           float* __restrict__ sum_dz, float* __restrict__ sum_dzx) {
    const int c = blockIdx.x * STATS_FC + (threadIdx.x % STATS_FC);
    double s, q;
    stats_final_sums(part, slabs, cols, c, s, q);
    if (threadIdx.x >= STATS_FC || c >= cols) return;
    sum_dz[c] = (float)s;
    if (sum_dzx) sum_dzx[c] = (float)q;
}

// dz and du may be the SAME buffer (the autograd block runs it in place): neither is __restrict__; every thread reads its
// elements of dz before it writes them to du
template <int V>
__global__ void bn_relu_bwd_kernel(const float* dz, int ldz, const float* __restrict__ y, int ldy, int rows_host,
                                   const int* __restrict__ rows_dev, int cols, const float* __restrict__ mean,
                                   const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ sum_dz,
                                   const float* __restrict__ sum_dzx, float* du, int ldu) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const float inv_n = rows > 0 ? 1.f / (float)rows : 0.f;
    const int qn = cols / V;
    const int64_t total = (int64_t)rows * qn;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / qn; const int c = (int)(i - r * qn) * V;
        const VecF<V> yv = ldv<V>(y + r * ldy + c), g = ldv<V>(dz + r * ldz + c);
        VecF<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float xh = (yv.v[j] - mean[c + j]) * rstd[c + j];
            const float d = gamma[c + j] * rstd[c + j] * (g.v[j] - sum_dz[c + j] * inv_n - xh * (sum_dzx[c + j] * inv_n));
            o.v[j] = yv.v[j] > 0.f ? d : 0.f;
        }
        stv<V>(du + r * ldu + c, o);
    }
}

// the same in slab form (threads as in bn_bwd_partial_kernel), leaving the fp64 column sums of du per slab: the bias gradient of the
// Linear in front comes out of the pass that writes du. du may alias dz (every element is read before it is written).
template <int V>
__global__ __launch_bounds__(256) void bn_relu_bwd_sum_kernel(const float* dz, int ldz, const float* __restrict__ y, int ldy, int rows_host,
                                                              const int* __restrict__ rows_dev, int cols, int slab_rows,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ sum_dz,
                                                              const float* __restrict__ sum_dzx, float* du, int ldu, double* __restrict__ part) {
    const int rows = rows_dev ? *rows_dev : rows_host;
    const float inv_n = rows > 0 ? 1.f / (float)rows : 0.f;
    const int c0 = blockIdx.x * 64;
    const int quads = min(16, (cols - c0 + 3) >> 2);
    const int RL = 256 / quads;
    const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
    const int c = c0 + q * 4;
    const int r0 = blockIdx.y * slab_rows, r1 = min(r0 + slab_rows, rows);
    double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (rl < RL) {
        float m[4], rs[4], grs[4], k0[4], k1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = c + j < cols;
            m[j] = in ? mean[c + j] : 0.f; rs[j] = in ? rstd[c + j] : 0.f; grs[j] = in ? gamma[c + j] * rs[j] : 0.f;
            k0[j] = in ? sum_dz[c + j] * inv_n : 0.f; k1[j] = in ? sum_dzx[c + j] * inv_n : 0.f;
        }
        for (int r = r0 + rl; r < r1; r += RL) {
            float yv[4], g[4], o[4];
            if (V == 4) {
                const float4 y4 = *reinterpret_cast<const float4*>(y + (size_t)r * ldy + c);
                const float4 g4 = *reinterpret_cast<const float4*>(dz + (size_t)r * ldz + c);
                yv[0] = y4.x; yv[1] = y4.y; yv[2] = y4.z; yv[3] = y4.w; g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const bool in = c + j < cols; yv[j] = in ? y[(size_t)r * ldy + c + j] : 0.f; g[j] = in ? dz[(size_t)r * ldz + c + j] : 0.f; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (yv[j] - m[j]) * rs[j];
                const float d = grs[j] * (g[j] - k0[j] - xh * k1[j]);
                o[j] = yv[j] > 0.f ? d : 0.f;
                acc[0][j] += (double)o[j];
            }
            if (V == 4) *reinterpret_cast<float4*>(du + (size_t)r * ldu + c) = make_float4(o[0], o[1], o[2], o[3]);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < cols) du[(size_t)r * ldu + c + j] = o[j];
        }
    }
    stats_block_store(acc, quads, RL, q, rl, c0, cols, blockIdx.y, part);
}

// one thread per (segment, V columns), as segmax_affine_kernel, plus the winning row
template <int V>
__global__ __launch_bounds__(256) void segmax_arg_kernel(const float* __restrict__ Z, int ldz, const int* __restrict__ rowptr, int n_seg,
                                                         int H, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* __restrict__ out, int ldo, int* __restrict__ arg, int ld_arg,
                                                         float* __restrict__ zwin, int ldw) {
    const int qn = H / V;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n_seg * qn) return;
    const int v = (int)(t / qn), c = (int)(t - (int64_t)v * qn) * V;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    VecF<V> m;
    int a[V];
    if (e0 >= e1) {
#pragma unroll
        for (int j = 0; j < V; ++j) { m.v[j] = 0.f; arg[(size_t)v * ld_arg + c + j] = -1; }
        stv<V>(out + (size_t)v * ldo + c, m);
        if (zwin) stv<V>(zwin + (size_t)v * ldw + c, m);
        return;
    }
    VecF<V> s, sh, zw;
#pragma unroll
    for (int j = 0; j < V; ++j) { s.v[j] = scale ? scale[c + j] : 1.f; sh.v[j] = shift ? shift[c + j] : 0.f; }
    {
        const VecF<V> z = ldv<V>(Z + (size_t)e0 * ldz + c);
#pragma unroll
        for (int j = 0; j < V; ++j) { m.v[j] = z.v[j] * s.v[j] + sh.v[j]; a[j] = e0; zw.v[j] = z.v[j]; }
    }
    for (int e = e0 + 1; e < e1; ++e) {
        const VecF<V> z = ldv<V>(Z + (size_t)e * ldz + c);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float w = z.v[j] * s.v[j] + sh.v[j];
            if (w > m.v[j]) { m.v[j] = w; a[j] = e; zw.v[j] = z.v[j]; }   // strict: the first maximum keeps the gradient
        }
    }
    stv<V>(out + (size_t)v * ldo + c, m);
    VecF<V> ab;
#pragma unroll
    for (int j = 0; j < V; ++j) ab.v[j] = __int_as_float(a[j]);
    stv<V>(reinterpret_cast<float*>(arg) + (size_t)v * ld_arg + c, ab);
    if (zwin) stv<V>(zwin + (size_t)v * ldw + c, zw);
}

// FEW, LONG segments (the per-mesh pooling: 8 segments of 4 096 rows; one thread per (segment, V columns) would walk them serially
// on a handful of waves: 1.1 ms per call): one workgroup per (segment, 16 columns), 16 V row lanes with four loads in flight each,
// then a tree over the row lanes in LDS. Same result as the serial walk: the largest value, on ties the LOWEST row.
template <int V>
__global__ __launch_bounds__(256) void segmax_arg_long_kernel(const float* __restrict__ Z, int ldz, const int* __restrict__ rowptr, int H,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ out, int ldo, int* __restrict__ arg, int ld_arg,
                                                              float* __restrict__ zwin, int ldw) {
    constexpr int CG = 16 / V, RL = 256 / CG;
    const int v = blockIdx.y;
    const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int cl = cg * V, c = blockIdx.x * 16 + cl;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    float m[V], zw[V], s[V], sh[V];
    int a[V];
    const bool on = c + V <= H;                                      // (H is a multiple of V: a column group is in or out as a whole)
#pragma unroll
    for (int j = 0; j < V; ++j) {
        m[j] = 0.f; zw[j] = 0.f; a[j] = -1;
        s[j] = (on && scale) ? scale[c + j] : 1.f; sh[j] = (on && shift) ? shift[c + j] : 0.f;
    }
    if (on) {
        auto take = [&](const VecF<V>& z, int e) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float w = z.v[j] * s[j] + sh[j];
                if (a[j] < 0 || w > m[j]) { m[j] = w; a[j] = e; zw[j] = z.v[j]; }
            }
        };
        int e = e0 + rl;
        for (; e + 3 * RL < e1; e += 4 * RL) {
            const VecF<V> z0 = ldv<V>(Z + (size_t)e * ldz + c), z1 = ldv<V>(Z + (size_t)(e + RL) * ldz + c);
            const VecF<V> z2 = ldv<V>(Z + (size_t)(e + 2 * RL) * ldz + c), z3 = ldv<V>(Z + (size_t)(e + 3 * RL) * ldz + c);
            take(z0, e); take(z1, e + RL); take(z2, e + 2 * RL); take(z3, e + 3 * RL);
        }
        for (; e < e1; e += RL) take(ldv<V>(Z + (size_t)e * ldz + c), e);
    }
    __shared__ float sm[RL][16], sz[RL][16];
    __shared__ int sa[RL][16];
#pragma unroll
    for (int j = 0; j < V; ++j) { sm[rl][cl + j] = m[j]; sz[rl][cl + j] = zw[j]; sa[rl][cl + j] = a[j]; }
    __syncthreads();
    for (int h = RL / 2; h > 0; h >>= 1) {
        if (rl < h) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float mo = sm[rl + h][cl + j]; const int ao = sa[rl + h][cl + j];
                const float mm = sm[rl][cl + j]; const int am = sa[rl][cl + j];
                if (ao >= 0 && (am < 0 || mo > mm || (mo == mm && ao < am))) {
                    sm[rl][cl + j] = mo; sa[rl][cl + j] = ao; sz[rl][cl + j] = sz[rl + h][cl + j];
                }
            }
        }
        __syncthreads();
    }
    if (rl == 0 && on) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            out[(size_t)v * ldo + c + j] = sm[0][cl + j];
            arg[(size_t)v * ld_arg + c + j] = sa[0][cl + j];
            if (zwin) zwin[(size_t)v * ldw + c + j] = sz[0][cl + j];
        }
    }
}

// sums over SEGMENTS of the one-hot row gradient: dz[arg[v][c]][c] = dout[v][c]
__global__ __launch_bounds__(256) void segmax_bwd_partial_kernel(const float* __restrict__ dout, int ldd, const int* __restrict__ arg,
                                                                 int ld_arg, const float* __restrict__ Z, int ldz, int n_seg, int cols,
                                                                 int slab_rows, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, double* __restrict__ part) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int v0 = blockIdx.y * slab_rows;
    double s = 0.0, q = 0.0;
    if (c < cols) {
        const int v1 = min(v0 + slab_rows, n_seg);
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

// One workgroup per (64 columns, slab of rows), threads as in bn_bwd_partial_kernel (column quads x row lanes); besides du it leaves
// the fp64 column sums of du over its slab (= the bias gradient of the Linear behind this BatchNorm: no second pass over du).
template <int V>
__global__ __launch_bounds__(256) void segmax_bn_relu_bwd_kernel(const float* __restrict__ dout, int ldd, const int* __restrict__ arg, int ld_arg,
                                          const float* __restrict__ Z, int ldz, const int* __restrict__ rowptr, int n_seg,
                                          const int* __restrict__ seg_of_row, int row_capacity, int cols, int slab_rows,
                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                          const float* __restrict__ gamma, const float* __restrict__ sum_dz,
                                          const float* __restrict__ sum_dzx, int relu, float* __restrict__ du, int ldu,
                                          double* __restrict__ part /* NULL or [slabs][2][cols] */) {
    const int rows = rowptr[n_seg];
    const float inv_n = rows > 0 ? 1.f / (float)rows : 0.f;
    const int c0 = blockIdx.x * 64;
    const int quads = min(16, (cols - c0 + 3) >> 2);
    const int RL = 256 / quads;
    const int q = threadIdx.x % quads, rl = threadIdx.x / quads;
    const int c = c0 + q * 4;
    const int r0 = blockIdx.y * slab_rows, r1 = min(r0 + slab_rows, row_capacity);   // rows past the live count are zeroed: du feeds a GEMM over the capacity
    double acc[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (rl < RL) {
        float m[4], rs[4], grs[4], k0[4], k1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = c + j < cols;
            m[j] = in ? mean[c + j] : 0.f; rs[j] = in ? rstd[c + j] : 0.f; grs[j] = in ? gamma[c + j] * rs[j] : 0.f;
            k0[j] = in ? sum_dz[c + j] * inv_n : 0.f; k1[j] = in ? sum_dzx[c + j] * inv_n : 0.f;
        }
        for (int e = r0 + rl; e < r1; e += RL) {
            if (e >= rows) {
                if (V == 4) *reinterpret_cast<float4*>(du + (size_t)e * ldu + c) = make_float4(0.f, 0.f, 0.f, 0.f);
                else
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (c + j < cols) du[(size_t)e * ldu + c + j] = 0.f;
                continue;
            }
            const int v = seg_of_row[e];
            float zv[4], dv[4], o[4]; int av[4];
            if (V == 4) {
                const float4 z4 = *reinterpret_cast<const float4*>(Z + (size_t)e * ldz + c);
                const float4 d4 = *reinterpret_cast<const float4*>(dout + (size_t)v * ldd + c);
                const int4 a4 = *reinterpret_cast<const int4*>(arg + (size_t)v * ld_arg + c);
                zv[0] = z4.x; zv[1] = z4.y; zv[2] = z4.z; zv[3] = z4.w; dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
                av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool in = c + j < cols;
                    zv[j] = in ? Z[(size_t)e * ldz + c + j] : 0.f; dv[j] = in ? dout[(size_t)v * ldd + c + j] : 0.f;
                    av[j] = in ? arg[(size_t)v * ld_arg + c + j] : -1;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (zv[j] - m[j]) * rs[j];
                const float dzv = av[j] == e ? dv[j] : 0.f;
                const float g = grs[j] * (dzv - k0[j] - xh * k1[j]);
                o[j] = (!relu || zv[j] > 0.f) ? g : 0.f;
                acc[0][j] += (double)o[j];
            }
            if (V == 4) *reinterpret_cast<float4*>(du + (size_t)e * ldu + c) = make_float4(o[0], o[1], o[2], o[3]);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (c + j < cols) du[(size_t)e * ldu + c + j] = o[j];
        }
    }
    if (part) stats_block_store(acc, quads, RL, q, rl, c0, cols, blockIdx.y, part);
}

// dA[v] = sum over the rows of segment v (fixed order); dB[src(e)] += dG[e] (atomics). One thread per (segment, V columns).
template <int V>
__global__ __launch_bounds__(256) void edge_scatter_bwd_kernel(const float* __restrict__ dG, int ldg, const int* __restrict__ rowptr,
                                                               const int* __restrict__ srcS, int n_nodes, int H,
                                                               float* __restrict__ dA, int lda, float* __restrict__ dB, int ldb) {
    const int qn = H / V;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n_nodes * qn) return;
    const int v = (int)(t / qn), c = (int)(t - (int64_t)v * qn) * V;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    VecF<V> s;
#pragma unroll
    for (int j = 0; j < V; ++j) s.v[j] = 0.f;
    for (int e = e0; e < e1; ++e) {
        const VecF<V> g = ldv<V>(dG + (size_t)e * ldg + c);
        float* pb = dB + (size_t)srcS[e] * ldb + c;
#pragma unroll
        for (int j = 0; j < V; ++j) { s.v[j] += g.v[j]; atomicAdd(pb + j, g.v[j]); }
    }
    stv<V>(dA + (size_t)v * lda + c, s);
}

// The backward of Z[e] = relu(A[dst_e] + B[src_e]) with the BatchNorm behind it, WITHOUT materialising the per-edge gradient and
// without atomics: d[e] = [y > 0] gamma rstd (g - sum_dz / n - xhat sum_dzx / n) is evaluated where it is summed, once in the CSR
// order of the targets (dA[v]) and once in the order of the TRANSPOSED graph (dB[u] = sum over perm[k], k in the segment of u in
// rowptr_t: the edges out of u in ascending row order). Both sums run in a fixed order. mean == NULL: d[e] = g[e] (the plain
// scatter). One thread per (vertex, V columns).
template <int V>
__global__ __launch_bounds__(256) void edge_bn_scatter_bwd_kernel(const float* __restrict__ dG, int ldg, const float* __restrict__ Y, int ldy,
                                                                  const int* __restrict__ rowptr, const int* __restrict__ rowptr_t,
                                                                  const int* __restrict__ perm_t, int n_nodes, int n_src, int H,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ sum_dz,
                                                                  const float* __restrict__ sum_dzx, float* __restrict__ dA, int lda,
                                                                  float* __restrict__ dB, int ldb,
                                                                  const float* __restrict__ ZA /* NULL, or: Y[e] = relu(ZA[dst e] + ZB[src e]) */,
                                                                  int ldza, const float* __restrict__ ZB, int ldzb,
                                                                  const int* __restrict__ srcS, const int* __restrict__ dstS) {
    const int qn = H / V;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nmax = max(n_nodes, n_src);
    if (t >= (int64_t)nmax * qn) return;
    const int v = (int)(t / qn), c = (int)(t - (int64_t)v * qn) * V;
    const int rows = rowptr[n_nodes];
    const float inv_n = rows > 0 ? 1.f / (float)rows : 0.f;
    VecF<V> m, rs, grs, k0, k1;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        m.v[j] = mean ? mean[c + j] : 0.f; rs.v[j] = mean ? rstd[c + j] : 0.f; grs.v[j] = mean ? gamma[c + j] * rs.v[j] : 1.f;
        k0.v[j] = mean ? sum_dz[c + j] * inv_n : 0.f; k1.v[j] = mean ? sum_dzx[c + j] * inv_n : 0.f;
    }
    // Y given by its operands: the ReLU input is rebuilt from the two n x H matrices it was made of -- one of them is the SAME row
    // for a whole segment (`fixed`), the other a gathered row that the caches hold (2 x 33 MB at H = 256) -- instead of a second and
    // third read of the edges x H buffer from HBM; same expression as morig_edge_gather_relu, so the same bits
    auto grad = [&](int e, VecF<V>& acc, const VecF<V>& fixed, const float* __restrict__ other, int ld_other, const int* __restrict__ other_row) {
        const VecF<V> g = ldv<V>(dG + (size_t)e * ldg + c);
        if (mean) {
            VecF<V> y;
            if (ZA) {
                const VecF<V> o = ldv<V>(other + (size_t)other_row[e] * ld_other + c);
#pragma unroll
                for (int j = 0; j < V; ++j) { const float s = fixed.v[j] + o.v[j]; y.v[j] = s > 0.f ? s : 0.f; }
            } else {
                y = ldv<V>(Y + (size_t)e * ldy + c);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float xh = (y.v[j] - m.v[j]) * rs.v[j];
                const float d = grs.v[j] * (g.v[j] - k0.v[j] - xh * k1.v[j]);
                acc.v[j] += y.v[j] > 0.f ? d : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) acc.v[j] += g.v[j];
        }
    };
    if (v < n_nodes) {
        VecF<V> a;
#pragma unroll
        for (int j = 0; j < V; ++j) a.v[j] = 0.f;
        const int e1 = rowptr[v + 1];
        VecF<V> fa = a;
        if (ZA && mean) fa = ldv<V>(ZA + (size_t)v * ldza + c);               // every edge of this segment has target v
        for (int e = rowptr[v]; e < e1; ++e) grad(e, a, fa, ZB, ldzb, srcS);
        stv<V>(dA + (size_t)v * lda + c, a);
    }
    if (v < n_src) {
        VecF<V> b;
#pragma unroll
        for (int j = 0; j < V; ++j) b.v[j] = 0.f;
        const int k1e = rowptr_t[v + 1];
        VecF<V> fb = b;
        if (ZA && mean) fb = ldv<V>(ZB + (size_t)v * ldzb + c);               // every edge of this segment has source v
        for (int k = rowptr_t[v]; k < k1e; ++k) grad(perm_t[k], b, fb, ZA, ldza, dstS);
        stv<V>(dB + (size_t)v * ldb + c, b);
    }
}

// The two BatchNorm sums of the FIRST edge layer without a pass over the edges. The gradient that reaches its BatchNorm is
// dh = du2 W2 (rows = edges), so   sum_e dh[e][c]            = sum_k db2[k] W2[k][c]                       (db2 = column sums of du2)
//                                  sum_e dh[e][c] xhat[e][c] = rstd[c] sum_k W2[k][c] (M[k][c] - db2[k] mean[c])
// with M = du2^T Z1, the product the weight gradient dW2 is made of anyway (xhat = (Z1 - mean) rstd). Both passes over dh and Z1
// (18 GB per JointNetMotion step) become a walk over two H x H matrices, in fp64.
__global__ __launch_bounds__(256) void edge_bn_sums_from_products_kernel(const float* __restrict__ M, int ldm, const float* __restrict__ db2,
                                                                         const float* __restrict__ W2, int ldw, const float* __restrict__ mean,
                                                                         const float* __restrict__ rstd, int h_out, int h_in,
                                                                         float* __restrict__ sum_dz, float* __restrict__ sum_dzx) {
    // 16 columns x 16 row lanes per workgroup (one thread per column walked 256 rows with a dependent fp64 add each: 50 us), then the
    // lanes in a fixed tree
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < h_in)
        for (int k = kl; k < h_out; k += 16) {
            const double w = (double)W2[(size_t)k * ldw + c];
            a += (double)db2[k] * w;
            b += w * (double)M[(size_t)k * ldm + c];
        }
    __shared__ double sh[2][16][16];
    sh[0][kl][cl] = a; sh[1][kl][cl] = b;
    __syncthreads();
    for (int h = 8; h > 0; h >>= 1) {
        if (kl < h) { sh[0][kl][cl] += sh[0][kl + h][cl]; sh[1][kl][cl] += sh[1][kl + h][cl]; }
        __syncthreads();
    }
    if (kl != 0 || c >= h_in) return;
    a = sh[0][0][cl]; b = sh[1][0][cl];
    sum_dz[c] = (float)a;
    sum_dzx[c] = (float)((double)rstd[c] * (b - (double)mean[c] * a));
}

// ---- C = A^T B over the rows -----------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32: A operand lane l = A'[i = l & 31][k = l >> 5], B operand lane l = B'[k = l >> 5][j = l & 31]. With
// A' = A^T (i = a column n of A, k = a row r) both operands are 32 consecutive floats of a row: the row-major tiles go into LDS
// as they are and every fragment is one conflict-free ds_read_b32. Workgroup tile 128 (n) x 128 (k), 4 waves x (2 x 2) MFMA
// tiles, 32 rows staged at a time; a workgroup walks one chunk of the row range and writes one partial tile. The grid is one-
// dimensional and XCD-aware: hardware block id i runs on XCD i % 8, so block i takes logical tile (i % 8) * per_xcd + i / 8 and
// the tiles of one row chunk -- which read the same rows of A and B -- share one XCD's L2 (in tile-major order every XCD
// streamed the whole of B: 2 GB of L2 misses for a 32768 x 1024 x 1800 weight gradient whose operands are 370 MB).
constexpr int TN_T = 128, TN_R = 16;      // (rows per stage; two stages: the LDS footprint of the old single 32-row stage, 4 workgroups per CU)
typedef float tn_f32x16 __attribute__((ext_vector_type(16)));
typedef float tn_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                      const float* __restrict__ bshift,
                                                      int rows_host, const int* __restrict__ rows_dev, int N, int K, int chunk_rows,
                                                      int n_tiles, int k_tiles, int chunks, int per_xcd,
                                                      float* __restrict__ part /* [chunks][N][K] */) {
    // two LDS stages: the rows of stage s + 1 are written (from the registers a fetch filled one stage earlier) BETWEEN the MFMAs of
    // stage s, one barrier per stage. With one stage and two barriers every wave stopped issuing MFMAs for the write phase: 27 % of
    // the kernel (measurement builds -DTN_NO_LDSW / TN_NO_GLOBAL / TN_NO_MFMA, profiles/r03ac_gemm_tn_ablations.txt).
    __shared__ __attribute__((aligned(16))) float sA[2][TN_R][TN_T + 4], sB[2][TN_R][TN_T + 4];
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || logical >= n_tiles * k_tiles * chunks) return;
    const int bx = logical % n_tiles, by = (logical / n_tiles) % k_tiles, bz = logical / (n_tiles * k_tiles);
    const int n0 = bx * TN_T, k0 = by * TN_T;
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
    const int r_begin = bz * chunk_rows, r_end = min(r_begin + chunk_rows, rows);
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
    // 32-column sub-tiles that lie wholly past N / K (narrow layers: H = 16, 32; K = 3 position inputs) are skipped (wave-uniform)
    const bool on_a[2] = {n0 + wn < N, n0 + wn + 32 < N}, on_b[2] = {k0 + wk < K, k0 + wk + 32 < K};
    tn_f32x4 va[TN_R / 8], vb[TN_R / 8];
    tn_f32x4 sh4 = {0.f, 0.f, 0.f, 0.f};                            // B - 1 shift^T: this thread's four columns (0 past K, 0 without a shift)
    if (bshift) { for (int q = 0; q < 4; ++q) if (k0 + lc + q < K) sh4[q] = bshift[k0 + lc + q]; }
    auto fetch = [&](int r0) {                                      // global -> registers (in flight under the MFMAs of a whole stage)
#pragma unroll
        for (int p = 0; p < TN_R / 8; ++p) {
            const int r = r0 + p * 8 + lr;
            va[p] = tn_f32x4{0.f, 0.f, 0.f, 0.f}; vb[p] = tn_f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < r_end) {
                const float* pa = A + (size_t)r * lda + n0 + lc;
                const float* pb = B + (size_t)r * ldb + k0 + lc;
                if (a_vec && n0 + lc + 4 <= N) va[p] = *reinterpret_cast<const tn_f32x4*>(pa);
                else { for (int q = 0; q < 4; ++q) if (n0 + lc + q < N) va[p][q] = pa[q]; }
                if (b_vec && k0 + lc + 4 <= K) vb[p] = *reinterpret_cast<const tn_f32x4*>(pb);
                else { for (int q = 0; q < 4; ++q) if (k0 + lc + q < K) vb[p][q] = pb[q]; }
                vb[p] -= sh4;
            }
        }
    };
    auto put = [&](int st, int p) __attribute__((always_inline)) {
#ifndef TN_NO_LDSW                                      // (measurement builds, tools/build_variant.sh)
        *reinterpret_cast<tn_f32x4*>(&sA[st][p * 8 + lr][lc]) = va[p];
        *reinterpret_cast<tn_f32x4*>(&sB[st][p * 8 + lr][lc]) = vb[p];
#endif
    };
    if (r_begin < r_end) {
        fetch(r_begin);
#pragma unroll
        for (int p = 0; p < TN_R / 8; ++p) put(0, p);
#ifndef TN_NO_GLOBAL
        if (r_begin + TN_R < r_end) fetch(r_begin + TN_R);
#endif
    }
    __syncthreads();
    int st = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += TN_R, st ^= 1) {
        const bool more = r0 + TN_R < r_end;                        // block-uniform
#pragma unroll
        for (int kk = 0; kk < TN_R; kk += 2) {
            float fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = on_a[a] ? sA[st][kk + hi][wn + a * 32 + l31] : 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = on_b[b] ? sB[st][kk + hi][wk + b * 32 + l31] : 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#ifndef TN_NO_MFMA
                    if (on_a[a] && on_b[b]) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
#else
                    acc[a][b][0] += fa[a] * fb[b];
#endif
            // the next stage's rows go to the other buffer behind the first k-steps' MFMAs (their issue slots are free while the
            // matrix pipe works), then the fetch of the stage after it
            if (more && kk < 2 * (TN_R / 8)) put(st ^ 1, kk >> 1);
#ifndef TN_NO_GLOBAL
            if (kk == 2 * (TN_R / 8) && more && r0 + 2 * TN_R < r_end) fetch(r0 + 2 * TN_R);
#endif
        }
        __syncthreads();                                             // stage st consumed by every wave, stage st ^ 1 written
    }
    float* o = part + (size_t)bz * N * K;
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

// ---- the same contraction on the matrix cores' 16-bit rate: C = A^T B with every operand split into two bf16 halves ------------------
// [r04] Gradients span the float32 exponent range (1e-8 .. 1e+4 inside one step), which rules the fp16 split of the forward path
// out; bf16 keeps float32's exponent, so x = hi + lo with hi = bf16(x), lo = bf16(x - hi), both rounded to nearest, carries
// ~16 mantissa bits with no range guard, and  hi*hi + hi*lo + lo*hi  on v_mfma_f32_32x32x16_bf16 (float32 accumulation) runs at
// 16/3 = 5.3x the rate of the exact-float32 MFMA above. (The weight gradient is a sum over thousands of rows: 2^-16 per
// product is far inside the 2e-4-of-scale criterion of the block tests; MORIG_TRAIN_BWD=f32 keeps the exact kernel.)
// With the contraction over ROWS both MFMA operands need 8 CONSECUTIVE ROWS of one column per lane, i.e. the transposed tile:
// a loader thread owns a 4 x 4 block (4 rows x 4 columns: one float4 per row, 128 contiguous bytes per row and 8 lanes), splits
// it, transposes it in registers and writes 4 + 4 eight-byte pieces into the LDS image  sT[column][32 rows hi | 32 rows lo]
// (row pitch 144 bytes: the 8-byte writes of a half-wave -- row group fastest -- and the 16-byte fragment reads of 16 lanes both
// fall on distinct banks). Tiles, chunking over the rows, XCD-aware tile order and the fixed-order reduction are gemm_tn's.
constexpr int TN16_R = 32;                // rows per stage = two 16-row MFMA steps
constexpr int TN16_P = 144;               // bytes per column of the transposed image: 64 hi + 64 lo + 16 pad
typedef __bf16 tn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned tn_u32x2 __attribute__((ext_vector_type(2)));

// two values -> (packed bf16 hi pair, packed bf16 lo pair): hi = bf16(x) and lo = bf16(x - hi), both rounded to nearest even by the
// hardware conversion (v_cvt_pk_bf16_f32 on gfx950: one instruction per PAIR; the bit-twiddled form cost 8 VALU per element and made
// the kernel VALU-bound); x - hi is exact in float32
typedef __bf16 tn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_bf16(float x0, float x1, unsigned& hi, unsigned& lo) {
    const tn_bf16x2 h = {(__bf16)x0, (__bf16)x1};
    hi = __builtin_bit_cast(unsigned, h);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    const tn_bf16x2 l = {(__bf16)r0, (__bf16)r1};
    lo = __builtin_bit_cast(unsigned, l);
}

// One tile of one row chunk. FULL: the 128 x 128 tile lies inside N x K and both operands take 16-byte loads -- no column tests
// anywhere, no tests per row in the stages that are whole, no branches around the MFMAs (the counters of the general form on the
// step's large shapes: 8.7 VALU and 3 branches per MFMA, MFMA pipe 32 % busy; the guards, not the split, were most of it).
// WNT x WKT MFMA tiles per wave, WVN x WVK waves: 2 x 2 / 2 x 2 = the 128 x 128 tile on 256 threads; 4 x 2 / 2 x 4 = a 256 x 256 tile on 512
// threads (a third less LDS traffic per MFMA: 12 fragment reads per 24 MFMAs instead of 8 per 12, half the conversion writes), for
// outputs that have enough 256-tiles to fill the chip. The loader needs tile side = threads / 2 on both operands.
template <bool FULL, int WNT, int WKT, int WVN, int WVK>
__device__ __forceinline__ void tn16_tile(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                          const float* __restrict__ bshift, int N, int K, int n0, int k0, int n_own, int k_own,
                                          int r_begin, int r_end, bool a_vec, bool b_vec, char (&sA)[2][32 * WNT * WVN * TN16_P],
                                          char (&sB)[2][32 * WKT * WVK * TN16_P], float* __restrict__ o) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    static_assert(32 * WNT * WVN == 32 * WKT * WVK && 32 * WNT * WVN == 32 * WVN * WVK, "square tile, side = threads / 2");
    const int wn = (wave / WVK) * (32 * WNT), wk = (wave % WVK) * (32 * WKT);
    tn_f32x16 acc[WNT][WKT];
#pragma unroll
    for (int a = 0; a < WNT; ++a)
#pragma unroll
        for (int b = 0; b < WKT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int rg = tid & 7, cg = tid >> 3;                            // loader: rows 4 rg .. + 3 of the stage, columns 4 cg .. + 3 of the tile
    bool on_a[WNT], on_b[WKT];
#pragma unroll
    for (int a = 0; a < WNT; ++a) on_a[a] = n0 + wn + 32 * a < N;
#pragma unroll
    for (int b = 0; b < WKT; ++b) on_b[b] = k0 + wk + 32 * b < K;
    tn_f32x4 va[4], vb[4];
    tn_f32x4 sh4 = {0.f, 0.f, 0.f, 0.f};                              // B - 1 shift^T: this thread's four columns (0 past K, 0 without a shift)
    if (bshift) { for (int q = 0; q < 4; ++q) if (k0 + 4 * cg + q < K) sh4[q] = bshift[k0 + 4 * cg + q]; }
    const float* pa0 = A + (size_t)(4 * rg) * lda + n0 + 4 * cg;
    const float* pb0 = B + (size_t)(4 * rg) * ldb + k0 + 4 * cg;
    auto fetch = [&](int r0) {                                        // global -> registers (in flight under the MFMAs of a whole stage)
        if (FULL && r0 + TN16_R <= r_end) {                           // (block-uniform)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                va[i] = *reinterpret_cast<const tn_f32x4*>(pa0 + (size_t)(r0 + i) * lda);
                vb[i] = *reinterpret_cast<const tn_f32x4*>(pb0 + (size_t)(r0 + i) * ldb) - sh4;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + 4 * rg + i;
            va[i] = tn_f32x4{0.f, 0.f, 0.f, 0.f}; vb[i] = tn_f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < r_end) {
                const float* pa = A + (size_t)r * lda + n0 + 4 * cg;
                const float* pb = B + (size_t)r * ldb + k0 + 4 * cg;
                if (a_vec && n0 + 4 * cg + 4 <= N) va[i] = *reinterpret_cast<const tn_f32x4*>(pa);
                else { for (int q = 0; q < 4; ++q) if (n0 + 4 * cg + q < N) va[i][q] = pa[q]; }
                if (b_vec && k0 + 4 * cg + 4 <= K) vb[i] = *reinterpret_cast<const tn_f32x4*>(pb);
                else { for (int q = 0; q < 4; ++q) if (k0 + 4 * cg + q < K) vb[i][q] = pb[q]; }
                vb[i] -= sh4;
            }
        }
    };
    // split, transpose the 4 x 4 block and write column q's four rows as one 8-byte piece each for hi and lo
    auto put_one = [&](char* base, const tn_f32x4 (&v)[4], int q) __attribute__((always_inline)) {
        unsigned h01, l01, h23, l23;
        split_pair_bf16(v[0][q], v[1][q], h01, l01);
        split_pair_bf16(v[2][q], v[3][q], h23, l23);
        char* col = base + (4 * cg + q) * TN16_P + 8 * rg;
        *reinterpret_cast<tn_u32x2*>(col) = tn_u32x2{h01, h23};
        *reinterpret_cast<tn_u32x2*>(col + 64) = tn_u32x2{l01, l23};
    };
    auto put = [&](int st, int q) __attribute__((always_inline)) { put_one(sA[st], va, q); put_one(sB[st], vb, q); };
    if (r_begin < r_end) {
        fetch(r_begin);
#pragma unroll
        for (int q = 0; q < 4; ++q) put(0, q);
        if (r_begin + TN16_R < r_end) fetch(r_begin + TN16_R);
    }
    __syncthreads();
    int st = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += TN16_R, st ^= 1) {
        const bool more = r0 + TN16_R < r_end;                        // block-uniform
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {                              // the stage's two 16-row steps
            tn_bf16x8 ah[WNT], al[WNT], bh[WKT], bl[WKT];
#pragma unroll
            for (int a = 0; a < WNT; ++a) {
                const char* pa = sA[st] + (wn + a * 32 + l31) * TN16_P + 32 * s2 + 16 * hi;
                ah[a] = *reinterpret_cast<const tn_bf16x8*>(pa);
                al[a] = *reinterpret_cast<const tn_bf16x8*>(pa + 64);
            }
#pragma unroll
            for (int b = 0; b < WKT; ++b) {
                const char* pb = sB[st] + (wk + b * 32 + l31) * TN16_P + 32 * s2 + 16 * hi;
                bh[b] = *reinterpret_cast<const tn_bf16x8*>(pb);
                bl[b] = *reinterpret_cast<const tn_bf16x8*>(pb + 64);
            }
#pragma unroll
            for (int a = 0; a < WNT; ++a)
#pragma unroll
                for (int b = 0; b < WKT; ++b)
                    if (FULL || (on_a[a] && on_b[b])) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    }
            // the next stage's block goes to the other buffer behind this step's MFMAs (two columns per step), then the fetch of the
            // stage after it
            if (more) { put(st ^ 1, 2 * s2); put(st ^ 1, 2 * s2 + 1); }
            if (s2 == 1 && more && r0 + 2 * TN16_R < r_end) fetch(r0 + 2 * TN16_R);
        }
        // stage st consumed by every wave, stage st ^ 1 written. A raw barrier: __syncthreads() would also drain vmcnt, i.e. wait for
        // the global loads of the stage AFTER the next one, which were issued a moment ago and have a whole stage to arrive
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int a = 0; a < WNT; ++a)
#pragma unroll
        for (int b = 0; b < WKT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, k = k0 + wk + b * 32 + l31;
                // a tile that was moved back inside the matrix recomputes part of its neighbour's block: it stores only what it owns
                // (n >= n_own, k >= k_own), so no element of `part` has two writers
                if ((FULL || (n < N && k < K)) && n >= n_own && k >= k_own) o[(size_t)n * K + k] = acc[a][b][r];
            }
}

template <int WNT, int WKT, int WVN, int WVK>
__global__ __launch_bounds__(64 * WVN * WVK) void gemm_tn16_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                        const float* __restrict__ bshift, int rows_host, const int* __restrict__ rows_dev, int N, int K, int chunk_rows,
                                                        int n_tiles, int k_tiles, int chunks, int per_xcd,
                                                        float* __restrict__ part /* [chunks][N][K] */) {
    constexpr int TT = 32 * WNT * WVN;                                // tile side (both operands)
    __shared__ __attribute__((aligned(16))) char sA[2][TT * TN16_P], sB[2][TT * TN16_P];
    const int rows = rows_dev ? *rows_dev : rows_host;
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || logical >= n_tiles * k_tiles * chunks) return;
    const int bx = logical % n_tiles, by = (logical / n_tiles) % k_tiles, bz = logical / (n_tiles * k_tiles);
    int n0 = bx * TT, k0 = by * TT;
    const int n_own = n0, k_own = k0;                                 // first row / column of the block this workgroup owns
    const int r_begin = bz * chunk_rows, r_end = min(r_begin + chunk_rows, rows);
    const bool a_vec = (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
    const bool b_vec = (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
    // a last tile that would stick out is moved back inside (when the extent allows 16-byte loads there): it recomputes some columns
    // of its neighbour -- same operands, same order, the same bits, written twice -- and takes the guard-free path like every other tile
    if (a_vec && N >= TT && (N & 3) == 0) n0 = min(n0, N - TT);
    if (b_vec && K >= TT && (K & 3) == 0) k0 = min(k0, K - TT);
    float* o = part + (size_t)bz * N * K;
    if (a_vec && b_vec && n0 + TT <= N && k0 + TT <= K) tn16_tile<true, WNT, WKT, WVN, WVK>(A, lda, B, ldb, bshift, N, K, n0, k0, n_own, k_own, r_begin, r_end, a_vec, b_vec, sA, sB, o);
    else tn16_tile<false, WNT, WKT, WVN, WVK>(A, lda, B, ldb, bshift, N, K, n0, k0, n_own, k_own, r_begin, r_end, a_vec, b_vec, sA, sB, o);
}

// partial tiles -> C: 32 output elements x 8 chunk lanes per block; lane j adds chunks j, j + 8, ... in order, the eight lane sums
// are combined as a fixed tree (deterministic; the chain per lane is <= 64 additions)
// N, K <= 32 (the 16-wide motion branches of every EdgeConvMotion: ~70 weight gradients per step over 0.26-0.56 M edge rows each):
// a 128 x 128 MFMA tile is 1-6 % full there and the call took 50-60 us for 30-70 MB of operands. Plain FMAs instead: a workgroup
// walks its chunk of the rows in 64-row stages through LDS; a thread keeps a 4 x 4 block of the output (two 16-byte LDS reads per 16
// FMAs) and the P x P / 16 threads that cover the padded output form one of G row groups, group g taking rows g, g + G, ... of a
// stage; the groups are added in a fixed order at the end. Products are exact float32, every order is fixed.
template <int P /* padded output side: 16 or 32 */>
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                            const float* __restrict__ bshift, int rows_host, const int* __restrict__ rows_dev, int N, int K, int chunk_rows,
                                                            float* __restrict__ part) {
    constexpr int TPG = (P / 4) * (P / 4);                           // threads per row group
    constexpr int G = 256 / TPG;                                     // row groups: 16 (P = 16) or 4 (P = 32)
    constexpr int LD = P + 4;
    const int rows = rows_dev ? *rows_dev : rows_host;
    __shared__ __attribute__((aligned(16))) float sA[64 * LD], sB[64 * LD];
    const int tid = threadIdx.x;
    const int g = tid / TPG, t = tid % TPG;
    const int bi = (t / (P / 4)) * 4, bj = (t % (P / 4)) * 4;        // this thread's 4 x 4 output block
    constexpr int CPT = 64 * P / 256;                                // loader: consecutive columns per thread (4 or 8)
    const int lr = tid / (P / CPT), lc = (tid % (P / CPT)) * CPT;
    const int r0 = blockIdx.x * chunk_rows, r1 = min(r0 + chunk_rows, rows);
    const bool vec = ((N | K | lda | ldb) & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float shv[CPT];                                                  // B - 1 shift^T: this thread's loader columns
#pragma unroll
    for (int j = 0; j < CPT; ++j) shv[j] = (bshift && lc + j < K) ? bshift[lc + j] : 0.f;
    for (int rb = r0; rb < r1; rb += 64) {
        const int r = rb + lr;
        const bool live = r < r1;
        if (vec) {                                                   // 16-byte pieces: each lies wholly inside or outside the live columns
#pragma unroll
            for (int j = 0; j < CPT; j += 4) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&sA[lr * LD + lc + j]) =
                    (live && lc + j < N) ? *reinterpret_cast<const float4*>(A + (size_t)r * lda + lc + j) : z;
                float4 bq = z;
                if (live && lc + j < K) {
                    bq = *reinterpret_cast<const float4*>(B + (size_t)r * ldb + lc + j);
                    bq.x -= shv[j]; bq.y -= shv[j + 1]; bq.z -= shv[j + 2]; bq.w -= shv[j + 3];
                }
                *reinterpret_cast<float4*>(&sB[lr * LD + lc + j]) = bq;
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                sA[lr * LD + lc + j] = (live && lc + j < N) ? A[(size_t)r * lda + lc + j] : 0.f;
                sB[lr * LD + lc + j] = (live && lc + j < K) ? B[(size_t)r * ldb + lc + j] - shv[j] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < 64 / G; ++qq) {
            const int q = g + qq * G;
            const float4 a = *reinterpret_cast<const float4*>(&sA[q * LD + bi]);
            const float4 b = *reinterpret_cast<const float4*>(&sB[q * LD + bj]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    // the G row groups' blocks, added in group order
    __shared__ float red[256 * 16];                                  // [G][TPG][16]
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(g * TPG + t) * 16 + i * 4 + j] = acc[i][j];
    }
    __syncthreads();
    for (int o = tid; o < TPG * 16; o += 256) {
        const int tt = o >> 4, e = o & 15;
        float sum = 0.f;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) sum += red[(gg * TPG + tt) * 16 + e];
        const int n = (tt / (P / 4)) * 4 + (e >> 2), k = (tt % (P / 4)) * 4 + (e & 3);
        if (n < N && k < K) part[((size_t)blockIdx.x * N + n) * K + k] = sum;
    }
}

// (256 / LN) output elements x LN chunk lanes per workgroup, then a fixed tree over the lanes. LN = 32 for many chunks (narrow
// outputs over many rows: <= 16 additions per lane at 512 chunks), 8 for the wide outputs that have a handful of chunks.
template <int LN>
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ part, int chunks, int N, int K,
                                                             float* __restrict__ out, int ldo) {
    constexpr int EL = 256 / LN;
    const int64_t total = (int64_t)N * K;
    const int el = threadIdx.x % EL, ln = threadIdx.x / EL;
    const int64_t i = (int64_t)blockIdx.x * EL + el;
    float s = 0.f;
    if (i < total)
        for (int c = ln; c < chunks; c += LN) s += part[(size_t)c * total + i];
    __shared__ float sh[LN][EL];
    sh[ln][el] = s;
    __syncthreads();
    for (int h = LN / 2; h > 0; h >>= 1) {
        if (ln < h) sh[ln][el] += sh[ln + h][el];
        __syncthreads();
    }
    if (ln != 0 || i >= total) return;
    const int64_t n = i / K; const int k = (int)(i - n * K);
    out[n * ldo + k] = sh[0][el];
}

static void launch_tn_reduce(const float* part, int chunks, int N, int K, float* out, int ldo, hipStream_t s) {
    const int64_t total = (int64_t)N * K;
    if (chunks >= 64) hipLaunchKernelGGL(gemm_tn_reduce_kernel<32>, dim3((int)((total + 7) / 8)), dim3(256), 0, s, part, chunks, N, K, out, ldo);
    else hipLaunchKernelGGL(gemm_tn_reduce_kernel<8>, dim3((int)((total + 31) / 32)), dim3(256), 0, s, part, chunks, N, K, out, ldo);
}

static int tn_chunks(int rows, int N, int K) {
    const int tiles = cdiv(N, TN_T) * cdiv(K, TN_T);
    // ~4 workgroups per CU; 2 per CU for outputs of <= 16 tiles, whose partial tiles (chunks x N x K floats, written and read back) are
    // otherwise a tenth of the operand traffic (swept on the step's shapes: 256 x 1024 over 32 768 rows 0.121 -> 0.084 ms)
    // (rounded DOWN: two 128 x 128 workgroups fit a CU, 512 slots -- 120 tiles x 9 chunks = 1 080 workgroups ran as two rounds and a third of
    // 56; x 8 = 960 stays inside two)
    int chunks = ((tiles <= 16 && tiles > 4) ? 512 : 1024) / tiles;   // (<= 4 tiles: 256 chunks = one round of 256 x 256 workgroups: 0.160 -> 0.150 ms
                                                                      //  on 230 000 x 256 x 256; the same for 5-16 tiles measured slower)
    if (chunks > 512) chunks = 512;                                   // ... and at most 64 additions per lane of the reduction
    const int max_chunks = cdiv(rows > 0 ? rows : 1, 128);            // at least 128 rows per chunk
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
    const int slab_rows = stats_slab_rows(rows, cols);
    const int slabs = cdiv(rows > 0 ? rows : 1, slab_rows);
    if (workspace_doubles < (int64_t)slabs * 2 * cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 8.0 * rows * (double)cols);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dz, ldz, y, ldy, rows, rows_dev, cols, slab_rows,
                       mean, rstd, workspace, (const int*)nullptr, 0);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(cols, STATS_FC)), dim3(256), 0, s, workspace, slabs, cols, sum_dz, y ? sum_dzx : nullptr);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_bn_relu_backward(const float* dz, int32_t ldz, const float* y, int32_t ldy, int32_t rows, const int32_t* rows_dev,
                                      int32_t cols, const float* mean, const float* rstd, const float* gamma, const float* sum_dz,
                                      const float* sum_dzx, float* du, int32_t ldu, double* workspace, int64_t workspace_doubles,
                                      float* sum_du, void* stream) {
    if (!dz || !y || !mean || !rstd || !gamma || !sum_dz || !sum_dzx || !du) return MORIG_E_INVALID;
    if (rows < 0 || cols <= 0 || ldz < cols || ldy < cols || ldu < cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool v4 = (cols & 3) == 0 && vec4_ptr(dz, ldz) && vec4_ptr(y, ldy) && vec4_ptr(du, ldu);
    if (sum_du) {                                      // du and its column sums (the bias gradient) from one pass
        const int slab_rows = stats_slab_rows(rows, cols);
        const int slabs = cdiv(rows > 0 ? rows : 1, slab_rows);
        if (!workspace || workspace_doubles < (int64_t)slabs * 2 * cols) return MORIG_E_INVALID;
        ProfScope ps(K_MISC, s, 0.0, 12.0 * rows * (double)cols);
        if (v4) hipLaunchKernelGGL(bn_relu_bwd_sum_kernel<4>, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dz, ldz, y, ldy, rows, rows_dev, cols,
                                   slab_rows, mean, rstd, gamma, sum_dz, sum_dzx, du, ldu, workspace);
        else hipLaunchKernelGGL(bn_relu_bwd_sum_kernel<1>, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dz, ldz, y, ldy, rows, rows_dev, cols,
                                slab_rows, mean, rstd, gamma, sum_dz, sum_dzx, du, ldu, workspace);
        MORIG_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(cols, STATS_FC)), dim3(256), 0, s, workspace, slabs, cols, sum_du, (float*)nullptr);
        MORIG_LAUNCH_CHECK();
        return MORIG_OK;
    }
    if (rows == 0) return MORIG_OK;
    int64_t blocks = ((int64_t)rows * (cols / (v4 ? 4 : 1)) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    ProfScope ps(K_MISC, s, 0.0, 12.0 * rows * (double)cols);
    if (v4) hipLaunchKernelGGL(bn_relu_bwd_kernel<4>, dim3((int)blocks), dim3(256), 0, s, dz, ldz, y, ldy, rows, rows_dev, cols, mean, rstd,
                               gamma, sum_dz, sum_dzx, du, ldu);
    else hipLaunchKernelGGL(bn_relu_bwd_kernel<1>, dim3((int)blocks), dim3(256), 0, s, dz, ldz, y, ldy, rows, rows_dev, cols, mean, rstd,
                            gamma, sum_dz, sum_dzx, du, ldu);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_segmax_affine_arg(const float* Z, int32_t ldz, const int32_t* rowptr, int32_t n_segments, int32_t H,
                                       const float* scale, const float* shift, float* out, int32_t ldo, int32_t* arg, int32_t ld_arg,
                                       float* zwin, int32_t ldw, void* stream) {
    if (!Z || !rowptr || !out || !arg || n_segments <= 0 || H <= 0 || ldz < H || ldo < H || ld_arg < H) return MORIG_E_INVALID;
    if ((scale == nullptr) != (shift == nullptr) || (zwin && ldw < H)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 0.0);
    const bool v4 = (H & 3) == 0 && vec4_ptr(Z, ldz) && vec4_ptr(out, ldo) && vec4_ptr(arg, ld_arg) && (!zwin || vec4_ptr(zwin, ldw));
    const long threads = (long)n_segments * (H / (v4 ? 4 : 1));
    // too few (segment, column group) pairs to fill the chip AND few segments: the segments are what is long (mesh pooling: one
    // segment per mesh and replica, <= 64 x 5 on the path). A small GRAPH (hundreds of vertices, ~10 rows each) has as few pairs
    // but short segments: one 256-thread tree per (segment, 16 columns) would be slower there than the per-thread kernel
    if (threads < 32768 && n_segments <= 400) {
        const dim3 grid(cdiv(H, 16), n_segments);
        if (v4) hipLaunchKernelGGL(segmax_arg_long_kernel<4>, grid, dim3(256), 0, s, Z, ldz, rowptr, H, scale, shift, out, ldo, arg, ld_arg,
                                   zwin, ldw);
        else hipLaunchKernelGGL(segmax_arg_long_kernel<1>, grid, dim3(256), 0, s, Z, ldz, rowptr, H, scale, shift, out, ldo, arg, ld_arg,
                                zwin, ldw);
        MORIG_LAUNCH_CHECK();
        return MORIG_OK;
    }
    const int blocks = cdiv(threads, 256);
    if (v4) hipLaunchKernelGGL(segmax_arg_kernel<4>, dim3(blocks), dim3(256), 0, s, Z, ldz, rowptr, n_segments, H, scale, shift, out, ldo,
                               arg, ld_arg, zwin, ldw);
    else hipLaunchKernelGGL(segmax_arg_kernel<1>, dim3(blocks), dim3(256), 0, s, Z, ldz, rowptr, n_segments, H, scale, shift, out, ldo,
                            arg, ld_arg, zwin, ldw);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_segmax_bn_backward_stats(const float* dout, int32_t ldd, const int32_t* arg, int32_t ld_arg, const float* Z,
                                              int32_t ldz, const float* zwin, int32_t ldw, int32_t n_segments, int32_t cols,
                                              const float* mean, const float* rstd, double* workspace, int64_t workspace_doubles,
                                              float* sum_dz, float* sum_dzx, void* stream) {
    if (!dout || !arg || (!Z && !zwin) || !mean || !rstd || !workspace || !sum_dz || !sum_dzx) return MORIG_E_INVALID;
    if (n_segments <= 0 || cols <= 0 || ldd < cols || ld_arg < cols || (zwin ? ldw < cols : ldz < cols)) return MORIG_E_INVALID;
    const int slab_rows = stats_slab_rows(n_segments, cols);
    const int slabs = cdiv(n_segments, slab_rows);
    if (workspace_doubles < (int64_t)slabs * 2 * cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 12.0 * n_segments * (double)cols);
    // with the winners' values at hand (the forward kept them per segment) these are the plain BatchNorm sums over the segment rows,
    // all reads coalesced; without, every (segment, column) fetches its own row of Z
    if (zwin) hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dout, ldd, zwin, ldw, n_segments,
                                 (const int*)nullptr, cols, slab_rows, mean, rstd, workspace, arg, ld_arg);
    else hipLaunchKernelGGL(segmax_bwd_partial_kernel, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dout, ldd, arg, ld_arg, Z, ldz, n_segments,
                       cols, slab_rows, mean, rstd, workspace);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(cols, STATS_FC)), dim3(256), 0, s, workspace, slabs, cols, sum_dz, sum_dzx);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_segmax_bn_relu_backward(const float* dout, int32_t ldd, const int32_t* arg, int32_t ld_arg, const float* Z,
                                             int32_t ldz, const int32_t* rowptr, int32_t n_segments, const int32_t* seg_of_row,
                                             int32_t row_capacity, int32_t cols, const float* mean, const float* rstd, const float* gamma,
                                             const float* sum_dz, const float* sum_dzx, int32_t relu, float* du, int32_t ldu,
                                             double* workspace, int64_t workspace_doubles, float* sum_du, void* stream) {
    if (!dout || !arg || !Z || !rowptr || !seg_of_row || !mean || !rstd || !gamma || !sum_dz || !sum_dzx || !du) return MORIG_E_INVALID;
    if (n_segments <= 0 || row_capacity <= 0 || cols <= 0 || ldd < cols || ld_arg < cols || ldz < cols || ldu < cols) return MORIG_E_INVALID;
    const int slab_rows = stats_slab_rows(row_capacity, cols);
    const int slabs = cdiv(row_capacity, slab_rows);
    if (sum_du && (!workspace || workspace_doubles < (int64_t)slabs * 2 * cols)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool v4 = (cols & 3) == 0 && vec4_ptr(Z, ldz) && vec4_ptr(du, ldu) && vec4_ptr(dout, ldd) && vec4_ptr(arg, ld_arg);
    ProfScope ps(K_MISC, s, 0.0, 16.0 * row_capacity * (double)cols);
    double* part = sum_du ? workspace : nullptr;
    if (v4) hipLaunchKernelGGL(segmax_bn_relu_bwd_kernel<4>, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dout, ldd, arg, ld_arg, Z, ldz, rowptr,
                               n_segments, seg_of_row, row_capacity, cols, slab_rows, mean, rstd, gamma, sum_dz, sum_dzx, relu, du, ldu, part);
    else hipLaunchKernelGGL(segmax_bn_relu_bwd_kernel<1>, dim3(cdiv(cols, 64), slabs), dim3(256), 0, s, dout, ldd, arg, ld_arg, Z, ldz, rowptr,
                            n_segments, seg_of_row, row_capacity, cols, slab_rows, mean, rstd, gamma, sum_dz, sum_dzx, relu, du, ldu, part);
    MORIG_LAUNCH_CHECK();
    if (sum_du) {
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(cols, STATS_FC)), dim3(256), 0, s, workspace, slabs, cols, sum_du, (float*)nullptr);
        MORIG_LAUNCH_CHECK();
    }
    return MORIG_OK;
}

extern "C" int morig_edge_scatter_backward(const float* dG, int32_t ldg, const int32_t* rowptr, const int32_t* src_sorted, int32_t n_nodes,
                                           int32_t n_src_nodes, int32_t H, float* dA, int32_t lda, float* dB, int32_t ldb, void* stream) {
    if (!dG || !rowptr || !src_sorted || !dA || !dB || n_nodes <= 0 || n_src_nodes <= 0 || H <= 0) return MORIG_E_INVALID;
    if (ldg < H || lda < H || ldb < H) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 0.0);
    MORIG_HIP_TRY(hipMemset2DAsync(dB, (size_t)ldb * sizeof(float), 0, (size_t)H * sizeof(float), (size_t)n_src_nodes, s));
    // one column per thread: a wave's atomics then fall into consecutive words (one float4 per thread -- four scattered atomic
    // instructions per row -- measured 4x slower)
    hipLaunchKernelGGL(edge_scatter_bwd_kernel<1>, dim3(cdiv((long)n_nodes * H, 256)), dim3(256), 0, s, dG, ldg, rowptr, src_sorted, n_nodes, H,
                       dA, lda, dB, ldb);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_edge_bn_scatter_backward(const float* dG, int32_t ldg, const float* Y, int32_t ldy, const int32_t* rowptr,
                                             const int32_t* rowptr_t, const int32_t* perm_t, int32_t n_nodes, int32_t n_src_nodes,
                                             int32_t H, const float* mean, const float* rstd, const float* gamma, const float* sum_dz,
                                             const float* sum_dzx, float* dA, int32_t lda, float* dB, int32_t ldb, const float* ZA,
                                             int32_t ldza, const float* ZB, int32_t ldzb, const int32_t* src_sorted,
                                             const int32_t* dst_sorted, void* stream) {
    if (!dG || !rowptr || !rowptr_t || !perm_t || !dA || !dB || n_nodes <= 0 || n_src_nodes <= 0 || H <= 0) return MORIG_E_INVALID;
    if (ldg < H || lda < H || ldb < H) return MORIG_E_INVALID;
    if (mean && (!rstd || !gamma || !sum_dz || !sum_dzx)) return MORIG_E_INVALID;
    if (mean && !ZA && (!Y || ldy < H)) return MORIG_E_INVALID;
    if (ZA && (!ZB || !src_sorted || !dst_sorted || ldza < H || ldzb < H)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 0.0);
    const bool v4 = (H & 3) == 0 && vec4_ptr(dG, ldg) && vec4_ptr(dA, lda) && vec4_ptr(dB, ldb) &&
                    (!mean || (ZA ? (vec4_ptr(ZA, ldza) && vec4_ptr(ZB, ldzb)) : vec4_ptr(Y, ldy)));
    const int nmax = n_nodes > n_src_nodes ? n_nodes : n_src_nodes;
    const int blocks = cdiv((long)nmax * (H / (v4 ? 4 : 1)), 256);
    if (v4) hipLaunchKernelGGL(edge_bn_scatter_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, dG, ldg, Y, ldy, rowptr, rowptr_t, perm_t, n_nodes,
                               n_src_nodes, H, mean, rstd, gamma, sum_dz, sum_dzx, dA, lda, dB, ldb, ZA, ldza, ZB, ldzb, src_sorted, dst_sorted);
    else hipLaunchKernelGGL(edge_bn_scatter_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, dG, ldg, Y, ldy, rowptr, rowptr_t, perm_t, n_nodes,
                            n_src_nodes, H, mean, rstd, gamma, sum_dz, sum_dzx, dA, lda, dB, ldb, ZA, ldza, ZB, ldzb, src_sorted, dst_sorted);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_edge_bn_sums_from_products(const float* M, int32_t ldm, const float* db2, const float* W2, int32_t ldw, const float* mean,
                                               const float* rstd, int32_t h_out, int32_t h_in, float* sum_dz, float* sum_dzx, void* stream) {
    if (!M || !db2 || !W2 || !mean || !rstd || !sum_dz || !sum_dzx || h_out <= 0 || h_in <= 0 || ldm < h_in || ldw < h_in) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 0.0);
    hipLaunchKernelGGL(edge_bn_sums_from_products_kernel, dim3(cdiv(h_in, 16)), dim3(256), 0, s, M, ldm, db2, W2, ldw, mean, rstd, h_out, h_in,
                       sum_dz, sum_dzx);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int64_t morig_gemm_tn_workspace(int32_t rows, int32_t N, int32_t K) {
    if (rows < 0 || N <= 0 || K <= 0) return 0;
    return (int64_t)tn_chunks(rows, N, K) * N * K;
}

extern "C" int morig_gemm_tn_shift(const float* A, int32_t lda, const float* B, int32_t ldb, const float* b_shift, int32_t rows,
                                   const int32_t* rows_dev, int32_t N, int32_t K, float* workspace, int64_t workspace_floats, float* out,
                                   int32_t ldo, void* stream);
extern "C" int morig_gemm_tn(const float* A, int32_t lda, const float* B, int32_t ldb, int32_t rows, const int32_t* rows_dev, int32_t N,
                             int32_t K, float* workspace, int64_t workspace_floats, float* out, int32_t ldo, void* stream) {
    return morig_gemm_tn_shift(A, lda, B, ldb, nullptr, rows, rows_dev, N, K, workspace, workspace_floats, out, ldo, stream);
}

// C = A^T (B - 1 b_shift^T): every row of B is centred on the vector b_shift [K] before it is split / multiplied (b_shift = NULL:
// plain A^T B). For products that are used as  M - colsum(A) (x) mean  afterwards (the BatchNorm sums of the first edge layer):
// with the rows centred on that mean the contraction itself carries no cancellation
extern "C" int morig_gemm_tn_shift(const float* A, int32_t lda, const float* B, int32_t ldb, const float* b_shift, int32_t rows,
                                   const int32_t* rows_dev, int32_t N, int32_t K, float* workspace, int64_t workspace_floats, float* out,
                                   int32_t ldo, void* stream) {
    if (!A || !B || !workspace || !out || rows < 0 || N <= 0 || K <= 0 || lda < N || ldb < K || ldo < K) return MORIG_E_INVALID;
    const int chunks = tn_chunks(rows, N, K);
    if (workspace_floats < (int64_t)chunks * N * K) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // MORIG_TRAIN_BWD=f32: the exact-float32 MFMA kernel; default: the bf16 x 3 split (gemm_tn16_kernel). Read per call: the tests flip it.
    const char* e_bwd = getenv("MORIG_TRAIN_BWD");
    const bool split16 = !(e_bwd && e_bwd[0] == 'f');
    ProfScope ps(K_MISC, s, 2.0 * rows * (double)N * K, 4.0 * rows * ((double)N + K));
    if (N <= 32 && K <= 32 && !getenv("MORIG_TN_NO_SMALL")) {
        const int chunk_rows = cdiv(cdiv(rows > 0 ? rows : 1, chunks), 64) * 64;
        if (N <= 16 && K <= 16)
            hipLaunchKernelGGL(gemm_tn_small_kernel<16>, dim3(chunks), dim3(256), 0, s, A, lda, B, ldb, b_shift, rows, rows_dev, N, K, chunk_rows, workspace);
        else
            hipLaunchKernelGGL(gemm_tn_small_kernel<32>, dim3(chunks), dim3(256), 0, s, A, lda, B, ldb, b_shift, rows, rows_dev, N, K, chunk_rows, workspace);
        MORIG_LAUNCH_CHECK();
        launch_tn_reduce(workspace, chunks, N, K, out, ldo, s);
        MORIG_LAUNCH_CHECK();
        return MORIG_OK;
    }
    const int RS = split16 ? TN16_R : TN_R;
    const int chunk_rows = cdiv(cdiv(rows > 0 ? rows : 1, chunks), RS) * RS;
    const int n_tiles = cdiv(N, TN_T), k_tiles = cdiv(K, TN_T);
    const int per_xcd = cdiv((long)n_tiles * k_tiles * chunks, 8);
    // large outputs: 256 x 256 tiles on 512 threads when they still make ~one workgroup per CU with the same chunking (the same partial
    // buffer and reduction): a third less LDS traffic per MFMA, which is what bounds the 128 x 128 form
    const int n_big = cdiv(N, 256), k_big = cdiv(K, 256);
    static const bool no_big = getenv("MORIG_TN_NO_BIG") != nullptr;
    // (one 512-thread workgroup fits a CU: the chunk count is cut to what makes ONE round of <= 256 workgroups -- 288 of them ran as a
    // full round plus a round of 32, at twice the time of 256)
    const int tiles_big = n_big * k_big;
    const int chunks_big = tiles_big <= 256 ? min(chunks, max(1, 256 / tiles_big)) : chunks;
    if (split16 && !no_big && N >= 256 && K >= 256 && (long)tiles_big * chunks_big >= 200) {
        const int chunk_rows_b = cdiv(cdiv(rows > 0 ? rows : 1, chunks_big), RS) * RS;
        const int per_xcd_b = cdiv((long)tiles_big * chunks_big, 8);
        hipLaunchKernelGGL((gemm_tn16_kernel<4, 2, 2, 4>), dim3(per_xcd_b * 8), dim3(512), 0, s, A, lda, B, ldb, b_shift, rows, rows_dev, N, K, chunk_rows_b,
                           n_big, k_big, chunks_big, per_xcd_b, workspace);
        MORIG_LAUNCH_CHECK();
        launch_tn_reduce(workspace, chunks_big, N, K, out, ldo, s);
        MORIG_LAUNCH_CHECK();
        return MORIG_OK;
    } else if (split16)
        hipLaunchKernelGGL((gemm_tn16_kernel<2, 2, 2, 2>), dim3(per_xcd * 8), dim3(256), 0, s, A, lda, B, ldb, b_shift, rows, rows_dev, N, K, chunk_rows, n_tiles,
                           k_tiles, chunks, per_xcd, workspace);
    else
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(per_xcd * 8), dim3(256), 0, s, A, lda, B, ldb, b_shift, rows, rows_dev, N, K, chunk_rows, n_tiles, k_tiles,
                           chunks, per_xcd, workspace);
    MORIG_LAUNCH_CHECK();
    launch_tn_reduce(workspace, chunks, N, K, out, ldo, s);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}
