// Graph preparation and small index-driven copies: HBM-bound integer/byte work.
//
// morig_csr_build: COO (int64, PyG layout) -> self-loop-normalised CSR by destination (int32).
//   pass 1  count   : in-degree of every target over non-loop edges               (atomics, 16 B/edge read)
//   pass 2  scan    : rowptr = exclusive scan of (in-degree + 1); 3 short launches  (8 B/node)
//   pass 3  fill    : every edge claims a slot in its target's segment (atomic cursor); one self loop
//                     per node is appended                                            (16 B/edge read, 8 B/edge write)
#include "common.h"

namespace morig {

constexpr int SCAN_T = 256;          // threads per scan block
constexpr int SCAN_I = 8;            // items per thread
constexpr int SCAN_B = SCAN_T * SCAN_I;

__global__ void csr_count_kernel(const int64_t* __restrict__ ei, int64_t E, int nsrc, int n, int skipneg,
                                 int* __restrict__ cnt, int* status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
        const int64_t s = ei[e], d = ei[E + e];
        if (skipneg && (s < 0 || d < 0)) continue;
        if (s < 0 || s >= nsrc || d < 0 || d >= n) { *status = 1; continue; }
        if (s != d) atomicAdd(&cnt[d], 1);
    }
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// block-wide exclusive scan of one int per thread (SCAN_T threads); returns exclusive prefix, total in *tot
__device__ __forceinline__ int block_excl_scan(int v, int* sh /* >= SCAN_T/64 + 1 ints */, int* tot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int inc = wave_incl_scan(v, lane);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int i = 0; i < SCAN_T / 64; ++i) { const int t = sh[i]; if (i < w) woff += t; total += t; }
    __syncthreads();
    *tot = total;
    return woff + inc - v;
}

// segment length of a target: its non-loop in-edges + one self loop, optionally rounded up to a multiple of 4
// (the padding slots repeat the self loop: max-aggregation is idempotent, and 4-aligned segments let the
// EdgeConv epilogue reduce each lane's 4 consecutive accumulator rows in registers)
__device__ __forceinline__ int seg_len(int cnt, int pad4) { const int d = cnt + 1; return pad4 ? ((d + 3) & ~3) : d; }

__global__ __launch_bounds__(SCAN_T) void scan_reduce_kernel(const int* __restrict__ cnt, int n, int pad4, int* __restrict__ bsum) {
    __shared__ int sh[SCAN_T / 64 + 1];
    const int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
    int v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) if (base + i < n) v += seg_len(cnt[base + i], pad4);
    int tot;
    (void)block_excl_scan(v, sh, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single block: in-place exclusive scan of nb block sums; grand total -> *total_out
__global__ __launch_bounds__(SCAN_T) void scan_blocksums_kernel(int* bsum, int nb, int* total_out) {
    __shared__ int sh[SCAN_T / 64 + 1];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += SCAN_T) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, sh, &tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_T) void scan_apply_kernel(const int* cnt_in, int n, int pad4, const int* __restrict__ bsum,
                                                           int* __restrict__ rowptr, int* cursor) {   // cursor may alias cnt_in
    __shared__ int sh[SCAN_T / 64 + 1];
    const int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
    int item[SCAN_I];
    int v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) { item[i] = (base + i < n) ? seg_len(cnt_in[base + i], pad4) : 0; v += item[i]; }
    int tot;
    int run = bsum[blockIdx.x] + block_excl_scan(v, sh, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        if (base + i < n) { rowptr[base + i] = run; cursor[base + i] = run; }   // cursor aliases cnt_in: own items only
        run += item[i];
    }
}

__global__ void csr_fill_kernel(const int64_t* __restrict__ ei, int64_t E, int nsrc, int n, int* __restrict__ cursor,
                                int* __restrict__ srcS, int* __restrict__ dstS) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t total = E + n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        int s, d;
        if (e < E) {
            const int64_t s64 = ei[e], d64 = ei[E + e];
            if (s64 < 0 || s64 >= nsrc || d64 < 0 || d64 >= n || s64 == d64) continue;
            s = (int)s64; d = (int)d64;
        } else {
            s = d = (int)(e - E);
        }
        const int pos = atomicAdd(&cursor[d], 1);
        srcS[pos] = s;
        dstS[pos] = d;
    }
}

// after the fill pass cursor[i] = end of node i's real entries: repeat the self loop up to rowptr[i+1]
__global__ void csr_pad_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ cursor,
                               int* __restrict__ srcS, int* __restrict__ dstS) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int e1 = rowptr[i + 1];
        for (int pos = cursor[i]; pos < e1; ++pos) { srcS[pos] = (int)i; dstS[pos] = (int)i; }
    }
}

__global__ void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows, int cols) {
    const int64_t n = (int64_t)rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / cols; const int c = (int)(i - r * cols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}

typedef __fp16 f16x2_t __attribute__((ext_vector_type(2)));
__global__ void copy2d_pad_kernel(const float* __restrict__ src, int lds, int rows, int cols, float* __restrict__ dst, int ldd,
                                  int slot, int split, int* ovf) {
    const int64_t n = (int64_t)rows * slot;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / slot; const int c = (int)(i - r * slot);
        const float v = c < cols ? src[r * lds + c] : 0.f;
        if (!split) { dst[r * ldd + c] = v; continue; }
        const __fp16 h = (__fp16)v;
        const __fp16 l = (__fp16)(v - (float)h);
        if (!(fabsf(v) < 65000.f)) *ovf = 1;
        __fp16* yh = reinterpret_cast<__fp16*>(dst + r * ldd) + (c >> 5) * 64 + (c & 31);
        yh[0] = h;
        yh[32] = l;
    }
}

// `replicas` copies in one launch: copy r reads the source window shifted by r * src_col_step columns and writes the
// destination window shifted by r * dst_row_step rows (keyframe features: column step 3, :86; position rows and the
// position-branch columns of the motion replicas: column step 0)
__global__ void copy2d_pad_rep_kernel(const float* __restrict__ src, int lds, int rows, int cols, int src_col_step,
                                      float* __restrict__ dst, int ldd, int slot, int replicas, int64_t dst_row_step, int split, int* ovf) {
    const int64_t n = (int64_t)rows * slot;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / slot; const int c = (int)(i - r * slot);
        float v = c < cols ? src[r * lds + c] : 0.f;
        for (int q = 0; q < replicas; ++q) {
            if (q > 0 && src_col_step != 0) v = c < cols ? src[r * lds + c + q * src_col_step] : 0.f;
            float* drow = dst + (r + q * dst_row_step) * ldd;
            if (!split) { drow[c] = v; continue; }
            const __fp16 h = (__fp16)v;
            const __fp16 l = (__fp16)(v - (float)h);
            if (!(fabsf(v) < 65000.f)) *ovf = 1;
            __fp16* yh = reinterpret_cast<__fp16*>(drow) + (c >> 5) * 64 + (c & 31);
            yh[0] = h;
            yh[32] = l;
        }
    }
}

// The same copies with four columns per thread (slot width a multiple of 4, 16-byte aligned destination rows): one 16-byte store
// per replica on the fp32 path, two 8-byte stores (hi pairs, lo pairs) in the split layout instead of eight 2-byte stores. The
// split is split_pair_f16 (common.h): hi truncated, lo = fp16(v - hi) rounded to nearest, as every GEMM epilogue writes it.
__global__ void copy2d_pad_rep4_kernel(const float* __restrict__ src, int lds, int rows, int cols, int src_col_step,
                                       float* __restrict__ dst, int ldd, int slot, int replicas, int64_t dst_row_step, int split, int* ovf) {
    const int q4 = slot >> 2;
    const int64_t n = (int64_t)rows * q4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / q4; const int c = (int)(i - r * q4) * 4;
        float v[4];
        for (int q = 0; q < replicas; ++q) {
            if (q == 0 || src_col_step != 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = c + j < cols ? src[r * lds + c + j + q * src_col_step] : 0.f;
            }
            float* drow = dst + (r + q * dst_row_step) * ldd;
            if (!split) { *reinterpret_cast<float4*>(drow + c) = make_float4(v[0], v[1], v[2], v[3]); continue; }
            float h0, h1, l0, l1;
            split_pair_f16(v[0], v[1], h0, l0);
            split_pair_f16(v[2], v[3], h1, l1);
            const float am = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            if (!(am < 65000.f)) *ovf = 1;
            char* yh = reinterpret_cast<char*>(drow) + 2 * ((c >> 5) * 64 + (c & 31));      // halves -> bytes; c % 4 == 0: 8-byte aligned
            *reinterpret_cast<float2*>(yh) = make_float2(h0, h1);
            *reinterpret_cast<float2*>(yh + 64) = make_float2(l0, l1);
        }
    }
}

__global__ void gather_cols_kernel(const float* __restrict__ src, int lds, const int* __restrict__ cols, int ncols,
                                   float* __restrict__ dst, int ldd, int rows) {
    const int64_t n = (int64_t)rows * ncols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / ncols; const int c = (int)(i - r * ncols);
        dst[r * ldd + c] = src[r * lds + cols[c]];
    }
}

__global__ void make_seg_kernel(const int64_t* __restrict__ batch, int n, int ng, int reps, int* __restrict__ seg) {
    const int64_t total = (int64_t)n * reps;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int r = (int)(i / n); const int v = (int)(i - (int64_t)r * n);
        seg[i] = r * ng + (int)batch[v];
    }
}

// cursor[0 .. n) = 0 and *status = 0 in ONE launch. (These were two hipMemsetAsync calls; as memset NODES of a captured HIP graph
// they made the second replay of the graph die with "write access to a read-only page" as soon as the allocator had handed out
// other memory in between -- tools/graph_probe2.py, ROCm 7.0; the larger 0xFF memset of the pooled GEMM replays fine. A kernel
// is also one launch instead of two.)
__global__ void csr_clear_kernel(int* __restrict__ cursor, int n, int* __restrict__ status) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) cursor[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *status = 0;
}

static inline int grid_for(int64_t n, int block = 256, int cap = 256 * 16) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

}  // namespace morig

using namespace morig;

extern "C" int morig_csr_build(const int64_t* edge_index, int64_t n_edges, int32_t n_nodes,
                               int32_t* rowptr, int32_t* src_sorted, int32_t* dst_sorted,
                               int32_t* cursor, int32_t* status, void* stream) {
    return morig_csr_build_bipartite(edge_index, n_edges, n_nodes, n_nodes, 0, rowptr, src_sorted, dst_sorted, cursor, status, stream);
}

extern "C" int morig_csr_build_bipartite(const int64_t* edge_index, int64_t n_edges, int32_t n_src_nodes, int32_t n_nodes,
                                         int32_t flags, int32_t* rowptr, int32_t* src_sorted, int32_t* dst_sorted,
                                         int32_t* cursor, int32_t* status, void* stream) {
    if (n_src_nodes < n_nodes) return MORIG_E_INVALID;
    const int skip_negative = flags & MORIG_CSR_SKIP_NEGATIVE ? 1 : 0;
    const int pad4 = flags & MORIG_CSR_PAD4 ? 1 : 0;
    if (!rowptr || !src_sorted || !dst_sorted || !cursor || !status) return MORIG_E_INVALID;
    if (n_edges < 0 || n_nodes <= 0 || (n_edges > 0 && !edge_index)) return MORIG_E_INVALID;
    if (n_edges + 4 * (int64_t)n_nodes > 0x7fffffffLL) return MORIG_E_UNSUPPORTED;   // int32 edge ids
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nb = cdiv(n_nodes, SCAN_B);
    int* bsum = dst_sorted;                       // scratch until the fill pass (capacity >= n_nodes >= nb)
    ProfScope ps(K_CSR, s, 0.0, 16.0 * n_edges * 2 + 8.0 * n_edges + 12.0 * n_nodes);
    hipLaunchKernelGGL(csr_clear_kernel, dim3(grid_for(n_nodes + 1)), dim3(256), 0, s, cursor, n_nodes + 1, status);
    MORIG_LAUNCH_CHECK();
    if (n_edges > 0) {
        hipLaunchKernelGGL(csr_count_kernel, dim3(grid_for(n_edges)), dim3(256), 0, s, edge_index, n_edges, n_src_nodes, n_nodes, skip_negative, cursor, status);
        MORIG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_T), 0, s, cursor, n_nodes, pad4, bsum);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(SCAN_T), 0, s, bsum, nb, rowptr + n_nodes);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_T), 0, s, cursor, n_nodes, pad4, bsum, rowptr, cursor);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_fill_kernel, dim3(grid_for(n_edges + n_nodes)), dim3(256), 0, s, edge_index, n_edges, n_src_nodes, n_nodes,
                       cursor, src_sorted, dst_sorted);
    MORIG_LAUNCH_CHECK();
    if (pad4) {
        hipLaunchKernelGGL(csr_pad_kernel, dim3(grid_for(n_nodes)), dim3(256), 0, s, n_nodes, rowptr, cursor, src_sorted, dst_sorted);
        MORIG_LAUNCH_CHECK();
    }
    return MORIG_OK;
}

// ---- CSR straight from a ball-query slot table: target k owns slots [k * max_nbrs, (k+1) * max_nbrs) of row 0 of the COO
// (unused = -1), so counting and filling need neither row 1 nor atomics: one wave per target, ballot + popcount ranks.
// Same normalisation as morig_csr_build_bipartite(MORIG_CSR_SKIP_NEGATIVE): pairs with source == target index dropped,
// one self loop (k, k) appended last.
__global__ __launch_bounds__(256) void csr_slots_count_kernel(const int64_t* __restrict__ src_slots, int n, int max_nbrs, int nsrc,
                                                              int* __restrict__ cnt, int* status) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int64_t v = lane < max_nbrs ? src_slots[(int64_t)k * max_nbrs + lane] : -1;
    if (v >= nsrc) *status = 1;
    const bool keep = v >= 0 && v < nsrc && v != k;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) cnt[k] = __popcll(mask);
}

__global__ __launch_bounds__(256) void csr_slots_fill_kernel(const int64_t* __restrict__ src_slots, int n, int max_nbrs, int nsrc,
                                                             const int* __restrict__ rowptr, int* __restrict__ srcS, int* __restrict__ dstS) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int64_t v = lane < max_nbrs ? src_slots[(int64_t)k * max_nbrs + lane] : -1;
    const bool keep = v >= 0 && v < nsrc && v != k;
    const unsigned long long mask = __ballot(keep);
    const int base = rowptr[k];
    if (keep) {
        const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
        srcS[pos] = (int)v; dstS[pos] = k;
    }
    if (lane == 0) { const int pos = base + __popcll(mask); srcS[pos] = k; dstS[pos] = k; }
}

extern "C" int morig_csr_from_slots(const int64_t* coo, int32_t n_nodes, int32_t max_nbrs, int32_t n_src_nodes,
                                    int32_t* rowptr, int32_t* src_sorted, int32_t* dst_sorted, int32_t* cursor, int32_t* status,
                                    void* stream) {
    if (!coo || !rowptr || !src_sorted || !dst_sorted || !cursor || !status) return MORIG_E_INVALID;
    if (n_nodes <= 0 || max_nbrs <= 0 || max_nbrs > 64 || n_src_nodes < n_nodes) return MORIG_E_INVALID;
    if ((int64_t)n_nodes * (max_nbrs + 1) > 0x7fffffffLL) return MORIG_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nb = cdiv(n_nodes, SCAN_B);
    int* bsum = dst_sorted;                       // scratch until the fill pass (capacity >= n_nodes >= nb)
    ProfScope ps(K_CSR, s, 0.0, 16.0 * n_nodes * (double)max_nbrs + 8.0 * n_nodes * (double)max_nbrs);
    hipLaunchKernelGGL(csr_clear_kernel, dim3(1), dim3(64), 0, s, cursor, 0, status);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_slots_count_kernel, dim3(cdiv(n_nodes, 4)), dim3(256), 0, s, coo, n_nodes, max_nbrs, n_src_nodes, cursor, status);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_T), 0, s, cursor, n_nodes, 0, bsum);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(SCAN_T), 0, s, bsum, nb, rowptr + n_nodes);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_T), 0, s, cursor, n_nodes, 0, bsum, rowptr, cursor);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_slots_fill_kernel, dim3(cdiv(n_nodes, 4)), dim3(256), 0, s, coo, n_nodes, max_nbrs, n_src_nodes, rowptr, src_sorted, dst_sorted);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_copy2d(const float* src, int32_t lds, float* dst, int32_t ldd, int32_t rows, int32_t cols, void* stream) {
    if (!src || !dst || rows < 0 || cols < 0 || lds < cols || ldd < cols) return MORIG_E_INVALID;
    if (rows == 0 || cols == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_COPY, s, 0.0, 8.0 * rows * cols);
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_copy2d_pad(const float* src, int32_t lds, int32_t rows, int32_t cols, float* dst, int32_t ldd,
                                int32_t slot_cols, int32_t split, int32_t* overflow, void* stream) {
    if (!src || !dst || rows < 0 || cols < 0 || slot_cols < cols || lds < cols || ldd < slot_cols) return MORIG_E_INVALID;
    if (split && (!overflow || (slot_cols & 31) || (ldd & 31) || (reinterpret_cast<uintptr_t>(dst) & 127))) return MORIG_E_INVALID;
    if (rows == 0 || slot_cols == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_COPY, s, 0.0, 4.0 * rows * (cols + slot_cols));
    hipLaunchKernelGGL(copy2d_pad_kernel, dim3(grid_for((int64_t)rows * slot_cols)), dim3(256), 0, s, src, lds, rows, cols, dst, ldd,
                       slot_cols, split, overflow);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_copy2d_pad_rep(const float* src, int32_t lds, int32_t rows, int32_t cols, int32_t src_col_step, float* dst,
                                    int32_t ldd, int32_t slot_cols, int32_t replicas, int64_t dst_row_step, int32_t split,
                                    int32_t* overflow, void* stream) {
    if (!src || !dst || rows < 0 || cols < 0 || slot_cols < cols || ldd < slot_cols || replicas < 1 || src_col_step < 0 || dst_row_step < 0)
        return MORIG_E_INVALID;
    if (lds < cols + (replicas - 1) * src_col_step) return MORIG_E_INVALID;
    if (split && (!overflow || (slot_cols & 31) || (ldd & 31) || (reinterpret_cast<uintptr_t>(dst) & 127))) return MORIG_E_INVALID;
    if (rows == 0 || slot_cols == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_COPY, s, 0.0, 4.0 * rows * ((double)cols + (double)slot_cols * replicas));
    if ((slot_cols & 3) == 0 && (ldd & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (dst_row_step * ldd) % 4 == 0)
        hipLaunchKernelGGL(copy2d_pad_rep4_kernel, dim3(grid_for((int64_t)rows * (slot_cols >> 2))), dim3(256), 0, s, src, lds, rows, cols,
                           src_col_step, dst, ldd, slot_cols, replicas, dst_row_step, split, overflow);
    else
        hipLaunchKernelGGL(copy2d_pad_rep_kernel, dim3(grid_for((int64_t)rows * slot_cols)), dim3(256), 0, s, src, lds, rows, cols, src_col_step,
                           dst, ldd, slot_cols, replicas, dst_row_step, split, overflow);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_gather_cols(const float* src, int32_t lds, const int32_t* cols, int32_t n_cols,
                                 float* dst, int32_t ldd, int32_t rows, void* stream) {
    if (!src || !dst || !cols || rows < 0 || n_cols <= 0 || ldd < n_cols) return MORIG_E_INVALID;
    if (rows == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_COPY, s, 0.0, 8.0 * rows * n_cols);
    hipLaunchKernelGGL(gather_cols_kernel, dim3(grid_for((int64_t)rows * n_cols)), dim3(256), 0, s, src, lds, cols, n_cols, dst, ldd, rows);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_make_seg(const int64_t* batch, int32_t n_nodes, int32_t n_graphs, int32_t replicas,
                              int32_t* seg_out, void* stream) {
    if (!batch || !seg_out || n_nodes <= 0 || n_graphs <= 0 || replicas <= 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_MISC, s, 0.0, 12.0 * n_nodes * replicas);
    hipLaunchKernelGGL(make_seg_kernel, dim3(grid_for((int64_t)n_nodes * replicas)), dim3(256), 0, s, batch, n_nodes, n_graphs, replicas, seg_out);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}
