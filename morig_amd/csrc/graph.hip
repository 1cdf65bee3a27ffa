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
// (pad4 == 2: the plain count -- the prefix sums of morig_geo_ball_graph's per-row member counts)
// (pad4 == 3, MORIG_CSR_MIN4: at least 4 rows per segment -- self-loop copies behind a short one -- so that a quad of 4 consecutive rows
// never holds more than two segments: what the mixed-quad epilogue of edge_ws.hip needs instead of 4-ALIGNED segments)
__device__ __forceinline__ int seg_len(int cnt, int pad4) {
    if (pad4 == 2) return cnt;
    const int d = cnt + 1;
    if (pad4 == 3) return d < 4 ? 4 : d;
    return pad4 ? ((d + 3) & ~3) : d;
}

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

// ---- dual build: the plain and the 4-aligned CSR of ONE graph from one count pass (the rig networks use both: the narrow layers
// run on the plain one, the 128 / 256-wide kernels on the padded one). Segment lengths differ (d + 1 vs round4(d + 1)), edge ranks
// inside a segment are shared: one atomic per edge claims rank k, the edge lands at rowptr[d] + k and rowptr4[d] + k.
__global__ __launch_bounds__(SCAN_T) void scan2_reduce_kernel(const int* __restrict__ cnt, int n, int* __restrict__ bsum, int* __restrict__ bsum4, int mode0) {
    __shared__ int sh[SCAN_T / 64 + 1];
    const int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
    int v = 0, v4 = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) if (base + i < n) { const int c = cnt[base + i]; v += seg_len(c, mode0); v4 += seg_len(c, 1); }
    int tot, tot4;
    (void)block_excl_scan(v, sh, &tot);
    (void)block_excl_scan(v4, sh, &tot4);
    if (threadIdx.x == 0) { bsum[blockIdx.x] = tot; bsum4[blockIdx.x] = tot4; }
}

__global__ __launch_bounds__(SCAN_T) void scan2_blocksums_kernel(int* bsum, int* bsum4, int nb, int* total_out, int* total4_out) {
    __shared__ int sh[SCAN_T / 64 + 1];
    int carry = 0, carry4 = 0;
    for (int b0 = 0; b0 < nb; b0 += SCAN_T) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0, v4 = i < nb ? bsum4[i] : 0;
        int tot, tot4;
        const int ex = block_excl_scan(v, sh, &tot);
        const int ex4 = block_excl_scan(v4, sh, &tot4);
        if (i < nb) { bsum[i] = carry + ex; bsum4[i] = carry4 + ex4; }
        carry += tot; carry4 += tot4;
    }
    if (threadIdx.x == 0) { *total_out = carry; *total4_out = carry4; }
}

__global__ __launch_bounds__(SCAN_T) void scan2_apply_kernel(const int* __restrict__ cnt, int n, const int* __restrict__ bsum,
                                                            const int* __restrict__ bsum4, int* __restrict__ rowptr, int* __restrict__ rowptr4, int mode0) {
    __shared__ int sh[SCAN_T / 64 + 1];
    const int base = blockIdx.x * SCAN_B + threadIdx.x * SCAN_I;
    int item[SCAN_I], item4[SCAN_I];
    int v = 0, v4 = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        const int c = (base + i < n) ? cnt[base + i] : 0;
        item[i] = (base + i < n) ? seg_len(c, mode0) : 0; item4[i] = (base + i < n) ? seg_len(c, 1) : 0;
        v += item[i]; v4 += item4[i];
    }
    int tot;
    int run = bsum[blockIdx.x] + block_excl_scan(v, sh, &tot);
    int run4 = bsum4[blockIdx.x] + block_excl_scan(v4, sh, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        if (base + i < n) { rowptr[base + i] = run; rowptr4[base + i] = run4; }
        run += item[i]; run4 += item4[i];
    }
}

__global__ void csr_fill_dual_kernel(const int64_t* __restrict__ ei, int64_t E, int n, const int* __restrict__ cnt, int* __restrict__ rank,
                                     const int* __restrict__ rowptr, int* __restrict__ srcS, int* __restrict__ dstS,
                                     const int* __restrict__ rowptr4, int* __restrict__ srcS4, int* __restrict__ dstS4, int mode0) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t total = E + n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        if (e < E) {
            const int64_t s64 = ei[e], d64 = ei[E + e];
            if (s64 < 0 || s64 >= n || d64 < 0 || d64 >= n || s64 == d64) continue;
            const int s = (int)s64, d = (int)d64;
            const int k = atomicAdd(&rank[d], 1);
            const int p = rowptr[d] + k, p4 = rowptr4[d] + k;
            srcS[p] = s; dstS[p] = d;
            srcS4[p4] = s; dstS4[p4] = d;
        } else {
            // node i: its self loop behind the cnt[i] real in-edges; the 4-aligned copy repeats it up to the segment's end
            const int i = (int)(e - E);
            const int c = cnt[i];
            const int e0 = rowptr[i] + seg_len(c, mode0);          // (MORIG_CSR_MIN4: copies of the self loop up to 4 rows)
            for (int q = rowptr[i] + c; q < e0; ++q) { srcS[q] = i; dstS[q] = i; }
            const int e4 = rowptr4[i] + seg_len(c, 1);
            for (int q = rowptr4[i] + c; q < e4; ++q) { srcS4[q] = i; dstS4[q] = i; }
        }
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

// K tails of several concatenations in one launch (morig_pack_tails): a thread = (row v, tail t, 4 adjacent columns) -> two 8-byte stores
// (hi pairs, lo pairs) of the row's ONE split chunk; columns [0, wa) from src[v][col_a[t] ..], [wa, wa + wb) from src[v][col_b[t] ..], zeros behind
struct TailCols { int a[MORIG_MAX_TAILS], b[MORIG_MAX_TAILS]; };
__global__ void pack_tails_kernel(const float* __restrict__ src, int lds, int rows, TailCols tc, int wa, int wb, int n_tails,
                                  float* __restrict__ dst, int64_t tail_stride, int* ovf) {
    const int64_t n = (int64_t)rows * n_tails * 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i & 7) * 4;
        const int64_t rv = i >> 3;
        const int64_t r = rv / n_tails; const int t = (int)(rv - r * n_tails);     // row-major: a source row is read once, by neighbouring threads
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cc = c + j;
            v[j] = cc < wa ? src[r * lds + tc.a[t] + cc] : (cc < wa + wb ? src[r * lds + tc.b[t] + (cc - wa)] : 0.f);
        }
        float h0, h1, l0, l1;
        split_pair_f16(v[0], v[1], h0, l0);
        split_pair_f16(v[2], v[3], h1, l1);
        const float am = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        if (!(am < 65000.f)) *ovf = 1;                     // (NaN too)
        char* yh = reinterpret_cast<char*>(dst + t * tail_stride + r * 32) + 2 * c;
        *reinterpret_cast<float2*>(yh) = make_float2(h0, h1);
        *reinterpret_cast<float2*>(yh + 64) = make_float2(l0, l1);
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

// ---------------------------------------------------------------------------------------------------------------------------
// Geodesic-ball graph on the device (data_proc/common_ops.py:214-226 get_geo_edges; SURVEY 8(d) synthetic recipe): for every
// vertex i the members j != i of its mesh with dist(i, j) <= radius, in index order; a row with more than max_nn members keeps a
// uniformly random subset of exactly max_nn (the reference: np.random.choice(members, max_nn, replace=False)); rows [i, member].
//
// Positions variant (Euclidean distance standing in for the geodesic one, as the synthetic recipe does): LDS-tiled brute force.
// A 256-thread block owns 4 x CPW consecutive centres; the candidates of their mesh(es) pass through LDS in tiles of GEO_TILE
// points (SoA, 12 KB), every wave tests 64 candidates per step against each of its CPW centres: ballot + popcount give the
// member ranks, the reservoir (Algorithm R, as morig_radius_sample: slot t lives in lane t's register, member number t >= max_nn
// replaces slot u = floor(U (t + 1)) when u < max_nn, U from a counter hash of (seed, row, t)) keeps the subset. No atomics, no
// distance matrix in memory: 12 B read per vertex per tile pass, max_nn int32 written per row.
// Distance-matrix variant (true geodesics: an n x n float64 matrix, the reference's own input): one wave per row streams the
// row (HBM-bound: 8 n^2 bytes), diagonal + 10 as common_ops.py:218 does.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int GEO_TILE = 1024;

__device__ __forceinline__ unsigned geo_mix32(unsigned a) {
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

// one 64-candidate step of a row's reservoir: `mask` = member lanes (index order = lane order), idx0 = candidate index of lane 0
__device__ __forceinline__ void geo_reservoir_step(unsigned long long mask, int idx0, int lane, int row, int max_nn, unsigned seed,
                                                   int& slot, int& seen) {
    const int nh = __popcll(mask);
    if (nh == 0) return;
    if (seen + nh <= max_nn) {                            // still filling: member number t goes to slot t (one assignment per lane)
        const int t = lane - seen;                        // which member of this step lands in my slot
        if (t >= 0 && t < nh) {
            unsigned long long m = mask;
            for (int q = 0; q < t; ++q) m &= m - 1ull;    // drop the t lowest members (nh <= max_nn <= 64: short)
            slot = idx0 + __builtin_ctzll(m);
        }
        seen += nh;
        return;
    }
    for (unsigned long long m = mask; m; m &= m - 1ull) { // wave-uniform walk over the members of this step
        const int idx = idx0 + __builtin_ctzll(m);
        const int t = seen;
        if (t < max_nn) { if (lane == t) slot = idx; }
        else {
            const unsigned h = geo_mix32(seed ^ geo_mix32((unsigned)row * 0x9E3779B9u + (unsigned)t));
            const unsigned u = (unsigned)(((unsigned long long)h * (unsigned long long)(t + 1)) >> 32);   // uniform in [0, t]
            if ((int)u < max_nn && lane == (int)u) slot = idx;
        }
        ++seen;
    }
}

__device__ __forceinline__ int geo_mesh_of(const int* __restrict__ mesh_ptr, int n_meshes, int v) {
    int lo = 0, hi = n_meshes;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (mesh_ptr[mid] <= v) lo = mid; else hi = mid; }
    return lo;
}

template <int CPW>
__global__ __launch_bounds__(256) void geo_ball_graph_kernel(const float* __restrict__ pos, int ldp, const int* __restrict__ mesh_ptr,
                                                             int n_meshes, int n, float r2, int max_nn, unsigned seed,
                                                             int* __restrict__ slots, int* __restrict__ counts, int* __restrict__ members) {
    __shared__ float sx[GEO_TILE], sy[GEO_TILE], sz[GEO_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk0 = blockIdx.x * 4 * CPW;
    const int blk1 = min(n, blk0 + 4 * CPW) - 1;          // last centre of the block (blk0 < n by the grid size)
    const int xs = mesh_ptr[geo_mesh_of(mesh_ptr, n_meshes, blk0)];
    const int xe = mesh_ptr[geo_mesh_of(mesh_ptr, n_meshes, blk1) + 1];
    int c[CPW], cs[CPW], ce[CPW], slot[CPW], seen[CPW];
    float cx[CPW], cy[CPW], cz[CPW];
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
        c[q] = blk0 + wave * CPW + q;
        slot[q] = -1; seen[q] = 0; cs[q] = ce[q] = 0; cx[q] = cy[q] = cz[q] = 0.f;
        if (c[q] < n) {
            const int mq = geo_mesh_of(mesh_ptr, n_meshes, c[q]);
            cs[q] = mesh_ptr[mq]; ce[q] = mesh_ptr[mq + 1];
            cx[q] = pos[(size_t)c[q] * ldp]; cy[q] = pos[(size_t)c[q] * ldp + 1]; cz[q] = pos[(size_t)c[q] * ldp + 2];
        }
    }
    for (int t0 = xs; t0 < xe; t0 += GEO_TILE) {
        __syncthreads();                                  // the previous tile is consumed
#pragma unroll
        for (int i = tid; i < GEO_TILE; i += 256) {
            const int j = t0 + i;
            if (j < xe) { const float* pj = pos + (size_t)j * ldp; sx[i] = pj[0]; sy[i] = pj[1]; sz[i] = pj[2]; }
        }
        __syncthreads();
        const int nt = min(GEO_TILE, xe - t0);
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            if (c[q] >= n || t0 >= ce[q] || t0 + nt <= cs[q]) continue;     // wave-uniform: this tile holds none of the centre's mesh
            for (int b = 0; b < nt; b += 64) {
                const int i = b + lane, j = t0 + i;
                bool hit = false;
                if (i < nt && j >= cs[q] && j < ce[q] && j != c[q]) {
                    const float dx = cx[q] - sx[i], dy = cy[q] - sy[i], dz = cz[q] - sz[i];
                    hit = (dx * dx + dy * dy) + dz * dz <= r2;
                }
                geo_reservoir_step(__ballot(hit), t0 + b, lane, c[q], max_nn, seed, slot[q], seen[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
        if (c[q] >= n) continue;
        if (lane < max_nn) slots[(int64_t)c[q] * max_nn + lane] = lane < seen[q] ? slot[q] : -1;
        if (lane == 0) { counts[c[q]] = min(seen[q], max_nn); if (members) members[c[q]] = seen[q]; }
    }
}

// distance-matrix variant: one wave per row, 4 x 64 candidates in flight per step
__global__ __launch_bounds__(256) void geo_ball_graph_dist_kernel(const double* __restrict__ dist, int64_t ldd, int n, double radius,
                                                                  int max_nn, unsigned seed, int* __restrict__ slots,
                                                                  int* __restrict__ counts, int* __restrict__ members) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* drow = dist + (int64_t)row * ldd;
    int slot = -1, seen = 0;
    for (int b = 0; b < n; b += 256) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = b + u * 64 + lane; v[u] = j < n ? drow[j] : 1e300; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = b + u * 64 + lane;
            const double d = j == row ? v[u] + 10.0 : v[u];               // surface_geodesic += 10 * eye (common_ops.py:218)
            geo_reservoir_step(__ballot(j < n && d <= radius), b + u * 64, lane, row, max_nn, seed, slot, seen);
        }
    }
    if (lane < max_nn) slots[(int64_t)row * max_nn + lane] = lane < seen ? slot : -1;
    if (lane == 0) { counts[row] = min(seen, max_nn); if (members) members[row] = seen; }
}

// rows [i, member] in row order, members in slot order; then (optionally) the n self loops the datasets append
// (datasets/dataset_rig.py:121-122). coo: int64 [2][n_out], n_out = offsets[n] (+ n)
__global__ __launch_bounds__(256) void geo_ball_fill_kernel(const int* __restrict__ slots, const int* __restrict__ offsets, int n, int max_nn,
                                                            int self_loops, int64_t* __restrict__ coo, int64_t n_out) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += stride) {
        const int o = offsets[row], cnt = offsets[row + 1] - o;
        if (lane < cnt) { coo[o + lane] = row; coo[n_out + o + lane] = slots[row * max_nn + lane]; }
    }
    if (self_loops) {
        const int64_t e0 = offsets[n];
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            coo[e0 + i] = i; coo[n_out + e0 + i] = i;
        }
    }
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
    if ((flags & MORIG_CSR_PAD4) && (flags & MORIG_CSR_MIN4)) return MORIG_E_INVALID;
    const int pad4 = flags & MORIG_CSR_PAD4 ? 1 : (flags & MORIG_CSR_MIN4 ? 3 : 0);       // seg_len's mode
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

extern "C" int morig_csr_build_dual(const int64_t* edge_index, int64_t n_edges, int32_t n_nodes, int32_t* rowptr, int32_t* src_sorted,
                                    int32_t* dst_sorted, int32_t* rowptr4, int32_t* src_sorted4, int32_t* dst_sorted4, int32_t* ws,
                                    int32_t flags, int32_t* status, void* stream) {
    if (flags & ~MORIG_CSR_MIN4) return MORIG_E_INVALID;
    if (!rowptr || !src_sorted || !dst_sorted || !rowptr4 || !src_sorted4 || !dst_sorted4 || !ws || !status) return MORIG_E_INVALID;
    if (n_edges < 0 || n_nodes <= 0 || (n_edges > 0 && !edge_index)) return MORIG_E_INVALID;
    if (n_edges + 4 * (int64_t)n_nodes > 0x7fffffffLL) return MORIG_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nb = cdiv(n_nodes, SCAN_B);
    int* cnt = ws;                                 // [n + 1]
    int* rank = ws + (n_nodes + 1);                // [n]      (cleared together with cnt: one contiguous range)
    int* bsum = dst_sorted;                        // scratch until the fill pass (capacity >= n_nodes >= nb)
    int* bsum4 = dst_sorted4;
    ProfScope ps(K_CSR, s, 0.0, 16.0 * n_edges * 2 + 16.0 * n_edges + 24.0 * n_nodes);
    hipLaunchKernelGGL(csr_clear_kernel, dim3(grid_for(2 * n_nodes + 1)), dim3(256), 0, s, ws, 2 * n_nodes + 1, status);
    MORIG_LAUNCH_CHECK();
    if (n_edges > 0) {
        hipLaunchKernelGGL(csr_count_kernel, dim3(grid_for(n_edges)), dim3(256), 0, s, edge_index, n_edges, n_nodes, n_nodes, 0, cnt, status);
        MORIG_LAUNCH_CHECK();
    }
    const int mode0 = (flags & MORIG_CSR_MIN4) ? 3 : 0;
    hipLaunchKernelGGL(scan2_reduce_kernel, dim3(nb), dim3(SCAN_T), 0, s, cnt, n_nodes, bsum, bsum4, mode0);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan2_blocksums_kernel, dim3(1), dim3(SCAN_T), 0, s, bsum, bsum4, nb, rowptr + n_nodes, rowptr4 + n_nodes);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan2_apply_kernel, dim3(nb), dim3(SCAN_T), 0, s, cnt, n_nodes, bsum, bsum4, rowptr, rowptr4, mode0);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_fill_dual_kernel, dim3(grid_for(n_edges + n_nodes)), dim3(256), 0, s, edge_index, n_edges, n_nodes, cnt, rank,
                       rowptr, src_sorted, dst_sorted, rowptr4, src_sorted4, dst_sorted4, mode0);
    MORIG_LAUNCH_CHECK();
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

// ---- geodesic-ball graph (kernels above) -------------------------------------------------------------------------------
static int geo_offsets(const int* counts, int n, int* offsets, int* scan_ws, hipStream_t s) {
    const int nb = cdiv(n, SCAN_B);
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_T), 0, s, counts, n, 2, scan_ws);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_blocksums_kernel, dim3(1), dim3(SCAN_T), 0, s, scan_ws, nb, offsets + n);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_T), 0, s, counts, n, 2, scan_ws, offsets, offsets);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_geo_ball_graph(const float* pos, int32_t ldp, const int32_t* mesh_ptr, int32_t n_meshes, int32_t n_nodes,
                                    float radius, int32_t max_nn, uint32_t seed, int32_t* slots, int32_t* counts, int32_t* members,
                                    int32_t* offsets, int32_t* scan_ws, void* stream) {
    if (!pos || !mesh_ptr || !slots || !counts || !offsets || !scan_ws) return MORIG_E_INVALID;
    if (ldp < 3 || n_meshes <= 0 || n_nodes <= 0 || max_nn <= 0 || max_nn > 64 || !(radius >= 0.f)) return MORIG_E_INVALID;
    if ((int64_t)n_nodes * max_nn > 0x7fffffffLL) return MORIG_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int CPW = 4;
    // algorithmic bytes: positions once, the slot table + counts + offsets written once
    ProfScope ps(K_GEOGRAPH, s, 0.0, 12.0 * n_nodes + 4.0 * n_nodes * (double)max_nn + 8.0 * n_nodes);
    hipLaunchKernelGGL(geo_ball_graph_kernel<CPW>, dim3(cdiv(n_nodes, 4 * CPW)), dim3(256), 0, s, pos, ldp, mesh_ptr, n_meshes, n_nodes,
                       radius * radius, max_nn, seed, slots, counts, members);
    MORIG_LAUNCH_CHECK();
    return geo_offsets(counts, n_nodes, offsets, scan_ws, s);
}

extern "C" int morig_geo_ball_graph_dist(const double* dist, int64_t ldd, int32_t n_nodes, double radius, int32_t max_nn, uint32_t seed,
                                         int32_t* slots, int32_t* counts, int32_t* members, int32_t* offsets, int32_t* scan_ws,
                                         void* stream) {
    if (!dist || !slots || !counts || !offsets || !scan_ws) return MORIG_E_INVALID;
    if (n_nodes <= 0 || ldd < n_nodes || max_nn <= 0 || max_nn > 64 || !(radius >= 0.0)) return MORIG_E_INVALID;
    if ((int64_t)n_nodes * max_nn > 0x7fffffffLL) return MORIG_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_GEOGRAPH, s, 0.0, 8.0 * n_nodes * (double)n_nodes + 4.0 * n_nodes * (double)max_nn);
    hipLaunchKernelGGL(geo_ball_graph_dist_kernel, dim3(cdiv(n_nodes, 4)), dim3(256), 0, s, dist, ldd, n_nodes, radius, max_nn, seed,
                       slots, counts, members);
    MORIG_LAUNCH_CHECK();
    return geo_offsets(counts, n_nodes, offsets, scan_ws, s);
}

extern "C" int morig_geo_ball_fill(const int32_t* slots, const int32_t* offsets, int32_t n_nodes, int32_t max_nn, int32_t self_loops,
                                   int64_t* coo, int64_t n_out, void* stream) {
    if (!slots || !offsets || !coo || n_nodes <= 0 || max_nn <= 0 || max_nn > 64 || n_out < 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_GEOGRAPH, s, 0.0, 4.0 * n_nodes * (double)max_nn + 16.0 * (double)n_out);
    hipLaunchKernelGGL(geo_ball_fill_kernel, dim3(grid_for((int64_t)n_nodes * 64)), dim3(256), 0, s, slots, offsets, n_nodes, max_nn,
                       self_loops, coo, n_out);
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
    // [r06] four columns per thread and 8- / 16-byte stores where the slot allows it (the one-replica case of copy2d_pad_rep4_kernel): the
    // element-per-thread form below pays a 64-bit division and two 2-byte stores per element -- 75 us for the 1.3 M x 32 feature slot
    if ((slot_cols & 3) == 0 && (ldd & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0)
        hipLaunchKernelGGL(copy2d_pad_rep4_kernel, dim3(grid_for((int64_t)rows * (slot_cols >> 2))), dim3(256), 0, s, src, lds, rows, cols, 0,
                           dst, ldd, slot_cols, 1, (int64_t)0, split, overflow);
    else
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

extern "C" int morig_pack_tails(const float* src, int32_t lds, int32_t rows, const int32_t* col_a, const int32_t* col_b, int32_t wa, int32_t wb,
                                int32_t n_tails, float* dst, int64_t tail_stride, int32_t* overflow, void* stream) {
    if (!src || !dst || !col_a || !col_b || !overflow || rows < 0 || n_tails < 1 || n_tails > MORIG_MAX_TAILS || wa < 0 || wb < 0 || wa + wb > 32 ||
        wa + wb == 0 || (tail_stride & 31) || tail_stride < (int64_t)rows * 32 || (reinterpret_cast<uintptr_t>(dst) & 127)) return MORIG_E_INVALID;
    TailCols tc = {};
    for (int t = 0; t < n_tails; ++t) {
        if (col_a[t] < 0 || col_b[t] < 0 || col_a[t] + wa > lds || col_b[t] + wb > lds) return MORIG_E_INVALID;
        tc.a[t] = col_a[t]; tc.b[t] = col_b[t];
    }
    if (rows == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_COPY, s, 0.0, 4.0 * rows * n_tails * (double)(wa + wb + 32));
    hipLaunchKernelGGL(pack_tails_kernel, dim3(grid_for((int64_t)rows * n_tails * 8)), dim3(256), 0, s, src, lds, rows, tc, wa, wb, n_tails,
                       dst, tail_stride, overflow);
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
