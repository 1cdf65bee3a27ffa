// Joint extraction that follows the hot path (evaluate/eval_rigging.py:80-95): voxel inside-test
// (utils/mst_utils.py:15-29), bandwidth = mean distance to the k-th nearest neighbour (sklearn estimate_bandwidth),
// weighted mean-shift (utils/cluster_utils.py:14-38), non-maximum suppression (utils/cluster_utils.py:41-66).
// The reference runs these in numpy float64 on a few thousand points per mesh; every kernel here is an O(n^2) fp64
// scan of one point set (VALU fp64, candidates streamed through LDS) and keeps numpy's operation order where a
// discrete decision depends on it (distance = ((dx^2 + dy^2) + dz^2), no FMA contraction; sqrt before "<= bandwidth").
#include "common.h"
#include <stdlib.h>

// Every kernel in this file makes index decisions (arg-max, "< r^2", k nearest, voxel cells) or reproduces a summation order, so
// products and sums round separately, as on the CPU: no fma contraction anywhere below. (hipcc's default contracts, and the
// __fmul_rn / __dadd_rn ... helpers do not prevent it: they are header functions with plain operators, compiled under the
// default -- hence the macros, which put the same operators under this pragma.)
#pragma clang fp contract(off)
#define __fmul_rn(a, b) ((a) * (b))
#define __fadd_rn(a, b) ((a) + (b))
#define __fsub_rn(a, b) ((a) - (b))
#define __dmul_rn(a, b) ((a) * (b))
#define __dadd_rn(a, b) ((a) + (b))
#define __dsub_rn(a, b) ((a) - (b))
#define __ddiv_rn(a, b) ((a) / (b))

namespace morig {

__device__ __forceinline__ double sqdist3d(double ax, double ay, double az, double bx, double by, double bz) {
    const double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by), dz = __dsub_rn(az, bz);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// ---- inside_check (mst_utils.py:15-29): vc = round((p - translate) / scale * dims0) half-to-even; inside the
// hard-coded 88^3 grid and on a filled voxel ----
__global__ void inside_check_kernel(const double* __restrict__ pts, int n, const unsigned char* __restrict__ vox, double tx, double ty,
                                    double tz, double scale, double dims0, unsigned char* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double t[3] = {tx, ty, tz};
    long vc[3]; bool in_grid = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v = __dmul_rn(__ddiv_rn(__dsub_rn(pts[(size_t)i * 3 + a], t[a]), scale), dims0);
        vc[a] = (long)rint(v);
        in_grid = in_grid && vc[a] >= 0 && vc[a] < 88;
        vc[a] = vc[a] < 0 ? 0 : (vc[a] > 87 ? 87 : vc[a]);
    }
    keep[i] = (in_grid && vox[(vc[0] * 88 + vc[1]) * 88 + vc[2]]) ? 1 : 0;
}

// ---- distance to the k-th nearest neighbour (the point itself included), one workgroup per point: radix select over
// the bit pattern of the squared distances (non-negative doubles order like their bits), 8 bits per pass, distances
// recomputed in every pass instead of stored ----
// Batched form of every kernel below: point sets of several meshes concatenated, mesh b = rows [ptr[b], ptr[b + 1]); blockIdx.y = b
// (ptr == nullptr: ONE set of n points). Per mesh the arithmetic and its order are those of the one-set launch.
__device__ __forceinline__ void mesh_range(const int* __restrict__ ptr, int n, int& s, int& e) {
    if (ptr) { s = ptr[blockIdx.y]; e = ptr[blockIdx.y + 1]; } else { s = 0; e = n; }
}

constexpr int KTH_SMALL = 8;
// k of mesh b: the given k, or (quantile >= 0) sklearn's int(n_b * quantile) floored at 1 (estimate_bandwidth's n_neighbors).
// EARLY: stop the radix passes as soon as the selected bin holds <= KTH_SMALL keys and pick the k-th among them from a list made
// by one more pass (pass 5 + the list pass instead of 8 passes on the 4 % quantile of 8192 points: 38.6 -> 24.4 ms for 64 meshes).
// Measured and not kept (profiles/r03m_kth_select_modes.txt): one aggregated atomic per wave and digit in the two top passes, where
// nearly every key falls into one bin (the ballot loop costs more than the LDS unit's own same-address serialisation: 66.9 ms);
// the points as three arrays instead of [n][3] (48.0 ms); 8 rows per workgroup sharing LDS tiles of candidates (53.7 ms).
template <bool EARLY>
__global__ __launch_bounds__(256) void kth_nn_kernel(const double* __restrict__ pts_all, const int* __restrict__ ptr, int n_all, int k_given,
                                                     double quantile, double* __restrict__ kth_all) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_k;
    __shared__ unsigned long long s_list[KTH_SMALL];
    __shared__ int s_nlist, s_stop;
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    if ((int)blockIdx.x >= n) return;
    const double* pts = pts_all + (size_t)s0 * 3;
    double* kth = kth_all + s0;
    int k = k_given;
    if (quantile >= 0.0) { k = (int)((double)n * quantile); if (k < 1) k = 1; }
    const int row = blockIdx.x, tid = threadIdx.x;
    const double px = pts[(size_t)row * 3], py = pts[(size_t)row * 3 + 1], pz = pts[(size_t)row * 3 + 2];
    if (tid == 0) { s_prefix = 0ull; s_k = k; s_nlist = 0; s_stop = -1; }
    for (int pass = 7; pass >= 0; --pass) {
        hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        const unsigned long long hi_mask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
        for (int j = tid; j < n; j += 256) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(
                sqdist3d(pts[(size_t)j * 3], pts[(size_t)j * 3 + 1], pts[(size_t)j * 3 + 2], px, py, pz));
            if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 0xffu], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int need = s_k; unsigned b = 0;
            while (b < 255u && (int)hist[b] < need) { need -= (int)hist[b]; ++b; }
            s_k = need; s_prefix = prefix | ((unsigned long long)b << (8 * pass));
            if (pass == 0 || (EARLY && (int)hist[b] <= KTH_SMALL)) s_stop = pass;          // (pass 0: whatever is left is all equal)
        }
        __syncthreads();
        if (s_stop >= 0) break;                                             // block-uniform
    }
    if (!EARLY) {                                                           // all 8 passes ran: the prefix IS the key
        if (tid == 0) kth[row] = sqrt(__longlong_as_double((long long)s_prefix));
        return;
    }
    const int sp = s_stop;
    const unsigned long long lmask = sp <= 0 ? ~0ull : (~0ull << (8 * sp));
    const unsigned long long prefix = s_prefix;
    for (int j = tid; j < n; j += 256) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(
            sqdist3d(pts[(size_t)j * 3], pts[(size_t)j * 3 + 1], pts[(size_t)j * 3 + 2], px, py, pz));
        if ((key & lmask) == prefix) {
            const int slot = atomicAdd(&s_nlist, 1);
            if (slot < KTH_SMALL) s_list[slot] = key;                       // (more than KTH_SMALL only when all of them are equal)
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int m = s_nlist < KTH_SMALL ? s_nlist : KTH_SMALL;
        unsigned long long v[KTH_SMALL];
        for (int i = 0; i < KTH_SMALL; ++i) v[i] = i < m ? s_list[i] : ~0ull;
        for (int i = 1; i < KTH_SMALL; ++i)
            for (int q = i; q > 0 && v[q] < v[q - 1]; --q) { const unsigned long long t = v[q]; v[q] = v[q - 1]; v[q - 1] = t; }
        int idx = s_k - 1;
        if (idx < 0) idx = 0;
        if (idx >= m) idx = m - 1;
        kth[row] = sqrt(__longlong_as_double((long long)v[idx]));
    }
}

// ---- the same selection, four points per workgroup and 12 key bits per pass ----
// What the one-row kernel pays for: (1) its two top passes are degenerate -- squared distances inside a normalised mesh share
// their sign and top exponent bits, so nearly every key lands in one bin and the LDS unit serialises 64 same-address atomics per
// wave instruction; (2) every row streams the mesh's 24-byte points again (n x 24 B per row and pass out of L2: the address unit's
// rate, not HBM's). Here a workgroup selects for FOUR rows at once -- a loaded candidate meets four row points held in
// registers -- and the first pass looks at key bits 46..57 of the keys whose top six bits are 0b001111 (squared distances in
// [2^-31, 2): six exponent bits + six mantissa bits = 64 bins per octave, so neighbouring distances spread over hundreds of bins);
// keys below that range (the point itself, exact duplicates) are only counted, keys above need no count. At the 4 % quantile of
// 8192 points the selected bin holds 2-5 keys, so the usual row takes ONE histogram pass + one list pass instead of three + one.
// Bins are 16-bit halves of a word shared by two rows (n <= 65535; larger sets run the one-row kernel), padded by one word per
// 64 so that the scan -- one wave per row, a lane sums 64 consecutive bins -- reads conflict-free. A row whose k-th key lies
// outside the range of the first pass (k <= number of duplicates of the point; sets far larger than the unit box) is selected by
// the generic 8 x 8-bit passes below, by the whole workgroup.
constexpr int K4_R = 4, K4_LIST = 64, K4_BINS = 4096, K4_WORDS = K4_BINS + K4_BINS / 64;
__device__ __forceinline__ int k4_idx(int b) { return b + (b >> 6); }
__device__ __forceinline__ unsigned k4_half(unsigned w, int r) { return (r & 1) ? (w >> 16) : (w & 0xffffu); }

__global__ __launch_bounds__(256) void kth_nn4_kernel(const double* __restrict__ pts_all, const int* __restrict__ ptr, int n_all, int k_given,
                                                      double quantile, double* __restrict__ kth_all) {
    __shared__ unsigned hist[K4_R / 2][K4_WORDS];
    __shared__ unsigned long long s_list[K4_R][K4_LIST];
    __shared__ unsigned long long s_prefix[K4_R];
    __shared__ int s_k[K4_R], s_nlist[K4_R], s_state[K4_R], s_shift[K4_R];      // state: 0 selecting, 1 list pass next, 2 done, 3 generic
    __shared__ unsigned s_lo[K4_R];
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    if ((int)blockIdx.x * K4_R >= n) return;
    const double* pts = pts_all + (size_t)s0 * 3;
    double* kth = kth_all + s0;
    int k = k_given;
    if (quantile >= 0.0) { k = (int)((double)n * quantile); if (k < 1) k = 1; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double px[K4_R], py[K4_R], pz[K4_R];
#pragma unroll
    for (int r = 0; r < K4_R; ++r) {                                           // rows past the end repeat the last one (never written)
        const int row = min((int)blockIdx.x * K4_R + r, n - 1);
        px[r] = pts[(size_t)row * 3]; py[r] = pts[(size_t)row * 3 + 1]; pz[r] = pts[(size_t)row * 3 + 2];
    }
    if (tid < K4_R) { s_prefix[tid] = 0ull; s_k[tid] = k; s_nlist[tid] = 0; s_state[tid] = 0; s_shift[tid] = 0; s_lo[tid] = 0u; }
    unsigned long long pre[K4_R];
    for (int p = 0; p < 5; ++p) {
        const int shift = p < 4 ? 46 - 12 * p : 0, nbits = p < 4 ? 12 : 10, pshift = shift + nbits;
        const unsigned bmask = (1u << nbits) - 1u;
        for (int i = tid; i < (K4_R / 2) * K4_WORDS; i += 256) (&hist[0][0])[i] = 0u;
        __syncthreads();
        unsigned act = 0u;
#pragma unroll
        for (int r = 0; r < K4_R; ++r) { if (s_state[r] == 0) act |= 1u << r; pre[r] = s_prefix[r]; }
        for (int j = tid; j < n; j += 256) {
            const double qx = pts[(size_t)j * 3], qy = pts[(size_t)j * 3 + 1], qz = pts[(size_t)j * 3 + 2];
#pragma unroll
            for (int r = 0; r < K4_R; ++r) {
                if (!((act >> r) & 1u)) continue;                              // block-uniform
                const unsigned long long key = (unsigned long long)__double_as_longlong(sqdist3d(qx, qy, qz, px[r], py[r], pz[r]));
                if (p == 0) {
                    const unsigned top = (unsigned)(key >> 58);
                    if (top == 0xFu) atomicAdd(&hist[r >> 1][k4_idx((int)((key >> 46) & 0xfffu))], 1u << (16 * (r & 1)));
                    else if (top < 0xFu) atomicAdd(&s_lo[r], 1u);
                } else if ((key >> pshift) == pre[r]) {
                    atomicAdd(&hist[r >> 1][k4_idx((int)((unsigned)(key >> shift) & bmask))], 1u << (16 * (r & 1)));
                }
            }
        }
        __syncthreads();
        if (wave < K4_R && ((act >> wave) & 1u)) {                             // wave r scans row r's bins
            const int r = wave;
            int need = s_k[r];
            bool generic = false;
            if (p == 0) { const int lo = (int)s_lo[r]; if (need <= lo) generic = true; else need -= lo; }
            const int per = (1 << nbits) / 64;
            unsigned tot = 0u;
            for (int c = 0; c < per; ++c) tot += k4_half(hist[r >> 1][k4_idx(lane * per + c)], r);
            unsigned inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
            const unsigned total = __shfl(inc, 63);
            if (!generic && need > (int)total) generic = true;                 // the k-th key lies above the range of pass 0
            if (generic) { if (lane == 0) s_state[r] = 3; }
            else {
                const unsigned long long m = __ballot((int)inc >= need);
                const int L = __ffsll((long long)m) - 1;
                if (lane == L) {
                    int rem = need - (int)(inc - tot), b = lane * per;
                    unsigned v = 0u;
                    for (int c = 0; c < per; ++c) {
                        v = k4_half(hist[r >> 1][k4_idx(lane * per + c)], r);
                        if (rem <= (int)v) { b = lane * per + c; break; }
                        rem -= (int)v;
                    }
                    s_k[r] = rem;
                    s_prefix[r] = (p == 0 ? (0xFull << 12) : (pre[r] << nbits)) | (unsigned long long)b;
                    s_shift[r] = shift;
                    s_state[r] = p == 4 ? 2 : ((int)v <= K4_LIST ? 1 : 0);     // (after the last pass the prefix IS the key)
                }
            }
        }
        __syncthreads();
        bool more = false;
#pragma unroll
        for (int r = 0; r < K4_R; ++r) more = more || s_state[r] == 0;
        if (!more) break;                                                      // block-uniform
    }
    // list pass: the keys of the selected bins, ranked by one wave per row
    unsigned lst = 0u;
    int sh[K4_R];
#pragma unroll
    for (int r = 0; r < K4_R; ++r) { if (s_state[r] == 1) lst |= 1u << r; pre[r] = s_prefix[r]; sh[r] = s_shift[r]; }
    if (lst) {
        for (int j = tid; j < n; j += 256) {
            const double qx = pts[(size_t)j * 3], qy = pts[(size_t)j * 3 + 1], qz = pts[(size_t)j * 3 + 2];
#pragma unroll
            for (int r = 0; r < K4_R; ++r) {
                if (!((lst >> r) & 1u)) continue;
                const unsigned long long key = (unsigned long long)__double_as_longlong(sqdist3d(qx, qy, qz, px[r], py[r], pz[r]));
                if ((key >> sh[r]) == pre[r]) {
                    const int slot = atomicAdd(&s_nlist[r], 1);
                    if (slot < K4_LIST) s_list[r][slot] = key;
                }
            }
        }
        __syncthreads();
        if (wave < K4_R && ((lst >> wave) & 1u)) {
            const int r = wave, m = min(s_nlist[r], K4_LIST);
            const unsigned long long v = lane < m ? s_list[r][lane] : ~0ull;
            int rank = 0;
            for (int j = 0; j < m; ++j) { const unsigned long long c = s_list[r][j]; rank += (c < v || (c == v && j < lane)) ? 1 : 0; }
            int idx = s_k[r] - 1;
            idx = idx < 0 ? 0 : (idx >= m ? m - 1 : idx);
            if (lane < m && rank == idx) s_prefix[r] = v;
        }
        __syncthreads();
    }
    // rows outside pass 0's range: 8 passes of 8 bits from the top, one row at a time
    for (int r = 0; r < K4_R; ++r) {
        if (s_state[r] != 3) continue;                                         // block-uniform
        unsigned* h = &hist[0][0];
        __syncthreads();
        if (tid == 0) { s_prefix[r] = 0ull; s_k[r] = k; }
        for (int pass = 7; pass >= 0; --pass) {
            h[tid] = 0u;
            __syncthreads();
            const unsigned long long prefix = s_prefix[r];
            const unsigned long long hi_mask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
            for (int j = tid; j < n; j += 256) {
                const unsigned long long key = (unsigned long long)__double_as_longlong(
                    sqdist3d(pts[(size_t)j * 3], pts[(size_t)j * 3 + 1], pts[(size_t)j * 3 + 2], px[r], py[r], pz[r]));
                if ((key & hi_mask) == prefix) atomicAdd(&h[(key >> (8 * pass)) & 0xffu], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int need = s_k[r]; unsigned b = 0;
                while (b < 255u && (int)h[b] < need) { need -= (int)h[b]; ++b; }
                s_k[r] = need; s_prefix[r] = prefix | ((unsigned long long)b << (8 * pass));
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid < K4_R) {
        const int row = (int)blockIdx.x * K4_R + tid;
        if (row < n) kth[row] = sqrt(__longlong_as_double((long long)s_prefix[tid]));
    }
}

// fixed-order sum (one workgroup): the bandwidth is deterministic from run to run
// (per mesh: mean of its kth distances -> out[b])
__global__ __launch_bounds__(256) void mean_f64_kernel(const double* __restrict__ x_all, const int* __restrict__ ptr, int n_all,
                                                       double* __restrict__ out) {
    __shared__ double part[256];
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    const double* x = x_all + s0;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.y] = n > 0 ? part[0] * (1.0 / (double)n) : 0.0;
}

// ---- one mean-shift step (cluster_utils.py:24-35), sources i streamed through LDS: moved_j = p_j + 0.3 (sum_i k_ij w_i p_i / (sum_i k_ij w_i + 1e-10) - p_j), k_ij = max(h^2 - d_ij^2, 0).
// state[t] accumulates sum |moved - p|^2 of step t; a step runs only while sqrt(state[t-1]) > 1e-3 (the reference's
// loop condition evaluated on the device: no host round trip per step), otherwise it passes the points through ----
// A workgroup owns 32 targets; its 256 threads are 8 source slices x 32 targets (one thread alone would walk all n sources:
// a dependent fp64 chain of ~n x 20 operations, 0.4 ms per step at n = 8192). Slice s takes sources 32 s .. 32 s + 31 of
// every 256-source tile; the 8 partial sums of a target are added in slice order (deterministic).
constexpr int MS_TILE = 256, MS_TGT = 32, MS_SL = MS_TILE / MS_TGT;
__global__ __launch_bounds__(MS_TILE) void meanshift_step_kernel(const double* __restrict__ src_all, const float* __restrict__ w_all,
                                                                 const int* __restrict__ ptr, int n_all,
                                                                 const double* __restrict__ bandwidth_all, int t, int max_iter,
                                                                 double* __restrict__ state_all, double* __restrict__ dst_all) {
    __shared__ double sx[MS_TILE], sy[MS_TILE], sz[MS_TILE], sw[MS_TILE];
    __shared__ double part[MS_SL][4][MS_TGT];
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    if ((int)blockIdx.x * MS_TGT >= n) return;
    const double* src = src_all + (size_t)s0 * 3;
    double* dst = dst_all + (size_t)s0 * 3;
    const float* w = w_all ? w_all + s0 : nullptr;
    const double* bandwidth = bandwidth_all + blockIdx.y;
    double* state = state_all + (size_t)blockIdx.y * max_iter;
    const int tg = threadIdx.x & (MS_TGT - 1), sl = threadIdx.x / MS_TGT;
    const int j = blockIdx.x * MS_TGT + tg;
    const bool live = j < n;
    const bool active = sqrt(state[t - 1]) > 1e-3;              // block-uniform
    const double px = live ? src[(size_t)j * 3] : 0.0, py = live ? src[(size_t)j * 3 + 1] : 0.0, pz = live ? src[(size_t)j * 3 + 2] : 0.0;
    if (!active) {
        if (live && sl == 0) { dst[(size_t)j * 3] = px; dst[(size_t)j * 3 + 1] = py; dst[(size_t)j * 3 + 2] = pz; }
        return;
    }
    const double h = bandwidth[0], h2 = __dmul_rn(h, h);
    double ax = 0.0, ay = 0.0, az = 0.0, aw = 0.0;
    for (int base = 0; base < n; base += MS_TILE) {
        const int i = base + threadIdx.x;
        __syncthreads();
        if (i < n) { sx[threadIdx.x] = src[(size_t)i * 3]; sy[threadIdx.x] = src[(size_t)i * 3 + 1]; sz[threadIdx.x] = src[(size_t)i * 3 + 2];
                     sw[threadIdx.x] = w ? (double)w[i] : 1.0; }
        __syncthreads();
        const int lo = sl * MS_TGT, hi = min(lo + MS_TGT, n - base);
        for (int r = lo; r < hi; ++r) {
            double kk = __dsub_rn(h2, sqdist3d(sx[r], sy[r], sz[r], px, py, pz));
            kk = kk > 0.0 ? kk : 0.0;
            kk = __dmul_rn(kk, sw[r]);
            aw = __dadd_rn(aw, kk);
            ax = __dadd_rn(ax, __dmul_rn(kk, sx[r])); ay = __dadd_rn(ay, __dmul_rn(kk, sy[r])); az = __dadd_rn(az, __dmul_rn(kk, sz[r]));
        }
    }
    part[sl][0][tg] = ax; part[sl][1][tg] = ay; part[sl][2][tg] = az; part[sl][3][tg] = aw;
    __syncthreads();
    double d2 = 0.0;
    if (sl == 0) {
        ax = ay = az = aw = 0.0;
#pragma unroll
        for (int q = 0; q < MS_SL; ++q) { ax += part[q][0][tg]; ay += part[q][1][tg]; az += part[q][2][tg]; aw += part[q][3][tg]; }
        if (live) {
            const double den = aw + 1e-10;
            const double mx = 0.3 * (ax / den - px) + px, my = 0.3 * (ay / den - py) + py, mz = 0.3 * (az / den - pz) + pz;
            dst[(size_t)j * 3] = mx; dst[(size_t)j * 3 + 1] = my; dst[(size_t)j * 3 + 2] = mz;
            d2 = (mx - px) * (mx - px) + (my - py) * (my - py) + (mz - pz) * (mz - pz);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) d2 += __shfl_xor(d2, o);
        if (tg == 0) atomicAdd(&state[t], d2);
    }
}

// ---- mean-shift over SPATIALLY SORTED point sets (batched path): the kernel k_ij = max(h^2 - d_ij^2, 0) has compact support, and
// with 4 % bandwidth quantiles ~95 % of the pairs contribute exactly 0. After a Morton sort (morig_morton_keys + a device sort,
// once per call: points move by fractions of the bandwidth per step) 32 consecutive targets and 256 consecutive sources are
// spatially compact, so a whole source tile is skipped when its bounding box is farther than h from the target block's: every
// skipped pair has k_ij = 0 exactly, the sums only lose zero terms. Tile boxes are rebuilt before every step (points moved).
__global__ __launch_bounds__(256) void tile_bbox_kernel(const double* __restrict__ src_all, const int* __restrict__ ptr, int n_all, int max_tiles,
                                                        double* __restrict__ bbox) {
    // one box per 64 consecutive sources (= the sources one wave of the step kernel takes from a 256-source tile)
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    const int base = blockIdx.x * MS_TILE;
    if (base >= n) return;
    const int i = base + threadIdx.x;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    if (i < n) {
#pragma unroll
        for (int a = 0; a < 3; ++a) lo[a] = hi[a] = src_all[(size_t)(s0 + i) * 3 + a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o)); }
    }
    if ((threadIdx.x & 63) == 0) {
        double* bx = bbox + (((size_t)blockIdx.y * max_tiles + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 6;
#pragma unroll
        for (int a = 0; a < 3; ++a) { bx[a] = lo[a]; bx[3 + a] = hi[a]; }
    }
}

__global__ __launch_bounds__(MS_TILE) void meanshift_step_sorted_kernel(const double* __restrict__ src_all, const float* __restrict__ w_all,
                                                                        const int* __restrict__ ptr, int n_all,
                                                                        const double* __restrict__ bandwidth_all, int t, int max_iter,
                                                                        const double* __restrict__ bbox_all, int max_tiles,
                                                                        double* __restrict__ state_all, double* __restrict__ dst_all) {
    __shared__ double sx[MS_TILE], sy[MS_TILE], sz[MS_TILE], sw[MS_TILE];
    __shared__ double part[MS_SL][4][MS_TGT];
    __shared__ double tbox[6];
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    if ((int)blockIdx.x * MS_TGT >= n) return;
    const double* src = src_all + (size_t)s0 * 3;
    double* dst = dst_all + (size_t)s0 * 3;
    const float* w = w_all ? w_all + s0 : nullptr;
    const double* bbox = bbox_all + (size_t)blockIdx.y * max_tiles * 24;      // 4 boxes of 64 sources per tile
    double* state = state_all + (size_t)blockIdx.y * max_iter;
    const int tg = threadIdx.x & (MS_TGT - 1), sl = threadIdx.x / MS_TGT;
    const int j = blockIdx.x * MS_TGT + tg;
    const bool live = j < n;
    const bool active = sqrt(state[t - 1]) > 1e-3;              // block-uniform
    const int jc = live ? j : n - 1;                            // dead lanes shadow the last point: they do not widen the box
    const double px = src[(size_t)jc * 3], py = src[(size_t)jc * 3 + 1], pz = src[(size_t)jc * 3 + 2];
    if (!active) {
        if (live && sl == 0) { dst[(size_t)j * 3] = px; dst[(size_t)j * 3 + 1] = py; dst[(size_t)j * 3 + 2] = pz; }
        return;
    }
    // bounding box of this block's 32 targets (the 32 tg lanes of slice 0 = lanes 0..31 of wave 0)
    if (sl == 0) {
        double l0 = px, l1 = py, l2 = pz, h0 = px, h1 = py, h2b = pz;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            l0 = fmin(l0, __shfl_xor(l0, o)); l1 = fmin(l1, __shfl_xor(l1, o)); l2 = fmin(l2, __shfl_xor(l2, o));
            h0 = fmax(h0, __shfl_xor(h0, o)); h1 = fmax(h1, __shfl_xor(h1, o)); h2b = fmax(h2b, __shfl_xor(h2b, o));
        }
        if (tg == 0) { tbox[0] = l0; tbox[1] = l1; tbox[2] = l2; tbox[3] = h0; tbox[4] = h1; tbox[5] = h2b; }
    }
    __syncthreads();
    const double h = bandwidth_all[blockIdx.y], h2 = __dmul_rn(h, h);
    double ax = 0.0, ay = 0.0, az = 0.0, aw = 0.0;
    const int n_tiles = (n + MS_TILE - 1) / MS_TILE;
    const int wv = threadIdx.x >> 6;                               // this wave's 64 sources of a tile = slices 2 wv, 2 wv + 1
    for (int tile = 0; tile < n_tiles; ++tile) {
        // box-to-box distances of the tile's four 64-source boxes (block-uniform): farther than h -> every pair has k = 0
        bool near_any = false, near_mine = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double* bx = bbox + ((size_t)tile * 4 + q) * 6;
            double gap2 = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double g = fmax(fmax(bx[a] - tbox[3 + a], tbox[a] - bx[3 + a]), 0.0);
                gap2 += g * g;
            }
            const bool nr = gap2 <= h2;
            near_any = near_any || nr;
            if (q == wv) near_mine = nr;
        }
        if (!near_any) continue;
        const int base = tile * MS_TILE;
        const int i = base + threadIdx.x;
        __syncthreads();
        if (i < n) { sx[threadIdx.x] = src[(size_t)i * 3]; sy[threadIdx.x] = src[(size_t)i * 3 + 1]; sz[threadIdx.x] = src[(size_t)i * 3 + 2];
                     sw[threadIdx.x] = w ? (double)w[i] : 1.0; }
        __syncthreads();
        if (!near_mine) continue;                                  // wave-uniform: this wave's 64 sources are all out of reach
        const int lo = sl * MS_TGT, hi = min(lo + MS_TGT, n - base);
        for (int r = lo; r < hi; ++r) {
            double kk = __dsub_rn(h2, sqdist3d(sx[r], sy[r], sz[r], px, py, pz));
            kk = kk > 0.0 ? kk : 0.0;
            kk = __dmul_rn(kk, sw[r]);
            aw = __dadd_rn(aw, kk);
            ax = __dadd_rn(ax, __dmul_rn(kk, sx[r])); ay = __dadd_rn(ay, __dmul_rn(kk, sy[r])); az = __dadd_rn(az, __dmul_rn(kk, sz[r]));
        }
    }
    part[sl][0][tg] = ax; part[sl][1][tg] = ay; part[sl][2][tg] = az; part[sl][3][tg] = aw;
    __syncthreads();
    double d2 = 0.0;
    if (sl == 0) {
        ax = ay = az = aw = 0.0;
#pragma unroll
        for (int q = 0; q < MS_SL; ++q) { ax += part[q][0][tg]; ay += part[q][1][tg]; az += part[q][2][tg]; aw += part[q][3][tg]; }
        if (live) {
            const double den = aw + 1e-10;
            const double mx = 0.3 * (ax / den - px) + px, my = 0.3 * (ay / den - py) + py, mz = 0.3 * (az / den - pz) + pz;
            dst[(size_t)j * 3] = mx; dst[(size_t)j * 3 + 1] = my; dst[(size_t)j * 3 + 2] = mz;
            d2 = (mx - px) * (mx - px) + (my - py) * (my - py) + (mz - pz) * (mz - pz);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) d2 += __shfl_xor(d2, o);
        if (tg == 0) atomicAdd(&state[t], d2);
    }
}

// ---- the sorted step at box granularity 32, without workgroup barriers in the pair loop ----
// The step above tests four boxes of every 256-source tile in every thread (24 loads + ~50 operations per tile and thread: as much
// work as the pairs that survive), stages whole tiles behind two barriers, and leaves a wave idle whenever its quarter of the tile
// is out of reach. Here a box holds 32 consecutive (Morton-sorted) points, for sources and targets alike -- the workgroup's 32
// targets ARE box blockIdx.x, and the workgroup writes the box of their NEW positions for the next step (two box buffers alternate;
// no separate box kernel after the first step). One thread tests one source box, the near ones are compacted in order into a list
// and dealt round-robin to the four waves. A wave stages a box in its own 1 KB of LDS (the next box's points are in flight in
// registers meanwhile) and its lanes are 32 targets x 2 halves of 16 sources. Sources past the end of the set carry weight 0 and
// finite coordinates: they add +0.0. Partial sums meet in a fixed order (deterministic; another order than the tile kernel's).
constexpr int MSB = 32;
__global__ __launch_bounds__(256) void box32_kernel(const double* __restrict__ src_all, const int* __restrict__ ptr, int n_all, int max_boxes,
                                                    double* __restrict__ bbox) {
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    const int base = blockIdx.x * 256;
    if (base >= n) return;
    const int i = min(base + (int)threadIdx.x, n - 1);                        // lanes past the end shadow the last point
    double lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = src_all[(size_t)(s0 + i) * 3 + a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o)); }
    }
    const int box = (base + (int)threadIdx.x) / MSB;
    if ((threadIdx.x & (MSB - 1)) == 0 && box * MSB < n) {
        double* bx = bbox + ((size_t)blockIdx.y * max_boxes + box) * 6;
#pragma unroll
        for (int a = 0; a < 3; ++a) { bx[a] = lo[a]; bx[3 + a] = hi[a]; }
    }
}

__global__ __launch_bounds__(256) void meanshift_step_box_kernel(const double* __restrict__ src_all, const float* __restrict__ w_all,
                                                                 const int* __restrict__ ptr, int n_all,
                                                                 const double* __restrict__ bandwidth_all, int t, int max_iter,
                                                                 const double* __restrict__ bbox_in_all, double* __restrict__ bbox_out_all,
                                                                 int max_boxes, double* __restrict__ state_all, double* __restrict__ dst_all) {
    __shared__ double stage[4][MSB][4];
    __shared__ double part[8][4][MSB];
    __shared__ unsigned long long s_mask[4];
    __shared__ unsigned short s_near[256];
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    if ((int)blockIdx.x * MSB >= n) return;
    const double* src = src_all + (size_t)s0 * 3;
    double* dst = dst_all + (size_t)s0 * 3;
    const float* w = w_all ? w_all + s0 : nullptr;
    const double* bbox_in = bbox_in_all + (size_t)blockIdx.y * max_boxes * 6;
    double* bbox_out = bbox_out_all + (size_t)blockIdx.y * max_boxes * 6;
    double* state = state_all + (size_t)blockIdx.y * max_iter;
    const int tid = threadIdx.x, tg = tid & (MSB - 1), sl = tid / MSB, wv = tid >> 6, lane = tid & 63, half = (tid >> 5) & 1;
    const int j = blockIdx.x * MSB + tg;
    const bool live = j < n;
    const bool active = sqrt(state[t - 1]) > 1e-3;              // block-uniform
    const int jc = live ? j : n - 1;                            // dead lanes shadow the last point
    const double px = src[(size_t)jc * 3], py = src[(size_t)jc * 3 + 1], pz = src[(size_t)jc * 3 + 2];
    if (!active) {
        if (live && sl == 0) { dst[(size_t)j * 3] = px; dst[(size_t)j * 3 + 1] = py; dst[(size_t)j * 3 + 2] = pz; }
        if (tid < 6) bbox_out[(size_t)blockIdx.x * 6 + tid] = bbox_in[(size_t)blockIdx.x * 6 + tid];
        return;
    }
    double tb[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) tb[a] = bbox_in[(size_t)blockIdx.x * 6 + a];
    const double h = bandwidth_all[blockIdx.y], h2 = __dmul_rn(h, h);
    double ax = 0.0, ay = 0.0, az = 0.0, aw = 0.0;
    const int n_boxes = (n + MSB - 1) / MSB;
    for (int cb = 0; cb < n_boxes; cb += 256) {
        const int b = cb + tid;
        bool near = false;
        if (b < n_boxes) {
            const double* bx = bbox_in + (size_t)b * 6;
            double gap2 = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double g = fmax(fmax(bx[a] - tb[3 + a], tb[a] - bx[3 + a]), 0.0);
                gap2 += g * g;
            }
            near = gap2 <= h2;                                                 // farther than h box to box: every pair has k = 0
        }
        const unsigned long long mask = __ballot(near);
        if (cb) __syncthreads();                                               // the previous chunk's list is consumed
        if (lane == 0) s_mask[wv] = mask;
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = __popcll(s_mask[q]); if (q < wv) off += c; total += c; }
        if (near) s_near[off + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)tid;
        __syncthreads();
        // this wave's boxes: list entries wv, wv + 4, ...
        int e = wv;
        double rx = px, ry = py, rz = pz, rw = 0.0;
        if (e < total && lane < MSB) {
            const int i = (cb + (int)s_near[e]) * MSB + lane;
            if (i < n) { rx = src[(size_t)i * 3]; ry = src[(size_t)i * 3 + 1]; rz = src[(size_t)i * 3 + 2]; rw = w ? (double)w[i] : 1.0; }
        }
        while (e < total) {
            if (lane < MSB) { stage[wv][lane][0] = rx; stage[wv][lane][1] = ry; stage[wv][lane][2] = rz; stage[wv][lane][3] = rw; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            e += 4;
            rx = px; ry = py; rz = pz; rw = 0.0;
            if (e < total && lane < MSB) {
                const int i = (cb + (int)s_near[e]) * MSB + lane;
                if (i < n) { rx = src[(size_t)i * 3]; ry = src[(size_t)i * 3 + 1]; rz = src[(size_t)i * 3 + 2]; rw = w ? (double)w[i] : 1.0; }
            }
#pragma unroll
            for (int r = 0; r < MSB / 2; ++r) {
                const double* sp = stage[wv][half * (MSB / 2) + r];
                const double qx = sp[0], qy = sp[1], qz = sp[2], qw = sp[3];
                double kk = __dsub_rn(h2, sqdist3d(qx, qy, qz, px, py, pz));
                kk = kk > 0.0 ? kk : 0.0;
                kk = __dmul_rn(kk, qw);
                aw = __dadd_rn(aw, kk);
                ax = __dadd_rn(ax, __dmul_rn(kk, qx)); ay = __dadd_rn(ay, __dmul_rn(kk, qy)); az = __dadd_rn(az, __dmul_rn(kk, qz));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    part[sl][0][tg] = ax; part[sl][1][tg] = ay; part[sl][2][tg] = az; part[sl][3][tg] = aw;
    __syncthreads();
    if (sl == 0) {
        ax = ay = az = aw = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { ax += part[q][0][tg]; ay += part[q][1][tg]; az += part[q][2][tg]; aw += part[q][3][tg]; }
        const double den = aw + 1e-10;
        const double mx = 0.3 * (ax / den - px) + px, my = 0.3 * (ay / den - py) + py, mz = 0.3 * (az / den - pz) + pz;
        double d2 = 0.0;
        if (live) {
            dst[(size_t)j * 3] = mx; dst[(size_t)j * 3 + 1] = my; dst[(size_t)j * 3 + 2] = mz;
            d2 = (mx - px) * (mx - px) + (my - py) * (my - py) + (mz - pz) * (mz - pz);
        }
        double l0 = mx, l1 = my, l2 = mz, h0 = mx, h1 = my, h2b = mz;          // (dead lanes moved with the last point)
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            d2 += __shfl_xor(d2, o);
            l0 = fmin(l0, __shfl_xor(l0, o)); l1 = fmin(l1, __shfl_xor(l1, o)); l2 = fmin(l2, __shfl_xor(l2, o));
            h0 = fmax(h0, __shfl_xor(h0, o)); h1 = fmax(h1, __shfl_xor(h1, o)); h2b = fmax(h2b, __shfl_xor(h2b, o));
        }
        if (tg == 0) {
            atomicAdd(&state[t], d2);
            double* bo = bbox_out + (size_t)blockIdx.x * 6;
            bo[0] = l0; bo[1] = l1; bo[2] = l2; bo[3] = h0; bo[4] = h1; bo[5] = h2b;
        }
    }
}

// Morton key of a point inside [-2, 2)^3 (10 bits per axis; coordinates outside are clamped), mesh index in the bits above
__device__ __forceinline__ unsigned part1by2(unsigned v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu; v = (v | (v << 4)) & 0x030c30c3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ void morton_keys_kernel(const double* __restrict__ pts, const int* __restrict__ ptr, int n_meshes, int n_all, long long* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_all) return;
    int lo = 0, hi = n_meshes;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ptr[mid] <= i) lo = mid; else hi = mid; }
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double v = (pts[(size_t)i * 3 + a] + 2.0) * 256.0;
        v = v < 0.0 ? 0.0 : (v > 1023.0 ? 1023.0 : v);
        q[a] = (unsigned)v;
    }
    keys[i] = ((long long)lo << 30) | (long long)(part1by2(q[0]) | (part1by2(q[1]) << 1) | (part1by2(q[2]) << 2));
}

__global__ void meanshift_state_init_kernel(double* __restrict__ state, int max_iter, int n_meshes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < max_iter * n_meshes) state[i] = (i % max_iter) == 0 ? 1e20 : 0.0;
}

// "sqrt(d2) <= h" (numpy compares the ROOTED distance, cluster_utils.py:49, 57) without a square root per pair: correctly rounded
// sqrt is monotone, so the pairs that pass are exactly those with d2 <= T, T = the largest double whose root is <= h. h * h is within
// a few ulps of T; walk to it.
__device__ __forceinline__ double sqrt_le_threshold(double h) {
    double T = __dmul_rn(h, h);
    for (int it = 0; it < 64 && sqrt(T) > h; ++it) T = __longlong_as_double(__double_as_longlong(T) - 1);
    for (int it = 0; it < 64; ++it) {
        const double u = __longlong_as_double(__double_as_longlong(T) + 1);
        if (!(sqrt(u) <= h)) break;
        T = u;
    }
    return T;
}

// ---- NMS (cluster_utils.py:48-51): neighbour counts within the bandwidth (the point itself included) ----
__global__ __launch_bounds__(256) void nms_counts_kernel(const double* __restrict__ pts_all, const int* __restrict__ ptr, int n_all,
                                                         const double* __restrict__ bandwidth_all, int* __restrict__ counts_all) {
    __shared__ double sx[256], sy[256], sz[256];
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    if ((int)blockIdx.x * 256 >= n) return;
    const double* pts = pts_all + (size_t)s0 * 3;
    int* counts = counts_all + s0;
    const double* bandwidth = bandwidth_all + blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool live = j < n;
    const double px = live ? pts[(size_t)j * 3] : 0.0, py = live ? pts[(size_t)j * 3 + 1] : 0.0, pz = live ? pts[(size_t)j * 3 + 2] : 0.0;
    const double T = sqrt_le_threshold(bandwidth[0]);
    int c = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        __syncthreads();
        if (i < n) { sx[threadIdx.x] = pts[(size_t)i * 3]; sy[threadIdx.x] = pts[(size_t)i * 3 + 1]; sz[threadIdx.x] = pts[(size_t)i * 3 + 2]; }
        __syncthreads();
        const int cnt = min(256, n - base);
        for (int r = 0; r < cnt; ++r) c += sqdist3d(sx[r], sy[r], sz[r], px, py, pz) <= T ? 1 : 0;
    }
    if (live) counts[j] = c;
}

// the same counts over Morton-sorted point sets with box culling (the layout of meanshift_step_box_kernel: 32 targets per workgroup,
// one thread tests one 32-point source box, near boxes dealt to the waves). The modes of a converged mean-shift sit in a few dozen
// tight clusters, so a target block meets a few per cent of the boxes. A box farther than h (rooted gap, the same monotone chain of
// roundings as a pair's distance: never larger than any of its pairs') holds no neighbour.
__global__ __launch_bounds__(256) void nms_counts_box_kernel(const double* __restrict__ pts_all, const int* __restrict__ ptr, int n_all,
                                                             const double* __restrict__ bandwidth_all, const double* __restrict__ bbox_all,
                                                             int max_boxes, int* __restrict__ counts_all) {
    __shared__ double stage[4][MSB][3];
    __shared__ int part[8][MSB];
    __shared__ unsigned long long s_mask[4];
    __shared__ unsigned short s_near[256];
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    if ((int)blockIdx.x * MSB >= n) return;
    const double* src = pts_all + (size_t)s0 * 3;
    const double* bbox = bbox_all + (size_t)blockIdx.y * max_boxes * 6;
    const int tid = threadIdx.x, tg = tid & (MSB - 1), sl = tid / MSB, wv = tid >> 6, lane = tid & 63, half = (tid >> 5) & 1;
    const int j = blockIdx.x * MSB + tg;
    const int jc = j < n ? j : n - 1;
    const double px = src[(size_t)jc * 3], py = src[(size_t)jc * 3 + 1], pz = src[(size_t)jc * 3 + 2];
    double tb[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) tb[a] = bbox[(size_t)blockIdx.x * 6 + a];
    const double T = sqrt_le_threshold(bandwidth_all[blockIdx.y]);
    int c = 0;
    const int n_boxes = (n + MSB - 1) / MSB;
    for (int cb = 0; cb < n_boxes; cb += 256) {
        const int b = cb + tid;
        bool near = false;
        if (b < n_boxes) {
            const double* bx = bbox + (size_t)b * 6;
            double gap2 = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double g = fmax(fmax(bx[a] - tb[3 + a], tb[a] - bx[3 + a]), 0.0);
                gap2 += g * g;
            }
            near = gap2 <= T;
        }
        const unsigned long long mask = __ballot(near);
        if (cb) __syncthreads();
        if (lane == 0) s_mask[wv] = mask;
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int k = __popcll(s_mask[q]); if (q < wv) off += k; total += k; }
        if (near) s_near[off + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)tid;
        __syncthreads();
        int e = wv;
        double rx = 1e300, ry = 1e300, rz = 1e300;                             // past the end: infinitely far
        if (e < total && lane < MSB) {
            const int i = (cb + (int)s_near[e]) * MSB + lane;
            if (i < n) { rx = src[(size_t)i * 3]; ry = src[(size_t)i * 3 + 1]; rz = src[(size_t)i * 3 + 2]; }
        }
        while (e < total) {
            if (lane < MSB) { stage[wv][lane][0] = rx; stage[wv][lane][1] = ry; stage[wv][lane][2] = rz; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            e += 4;
            rx = ry = rz = 1e300;
            if (e < total && lane < MSB) {
                const int i = (cb + (int)s_near[e]) * MSB + lane;
                if (i < n) { rx = src[(size_t)i * 3]; ry = src[(size_t)i * 3 + 1]; rz = src[(size_t)i * 3 + 2]; }
            }
#pragma unroll
            for (int r = 0; r < MSB / 2; ++r) {
                const double* sp = stage[wv][half * (MSB / 2) + r];
                c += sqdist3d(sp[0], sp[1], sp[2], px, py, pz) <= T ? 1 : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    part[sl][tg] = c;
    __syncthreads();
    if (sl == 0 && j < n) {
        int ct = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) ct += part[q][tg];
        counts_all[s0 + j] = ct;
    }
}

// ---- NMS greedy pass (cluster_utils.py:54-64), one 1024-thread workgroup: points visited in `order`; a point still
// alive suppresses everything within the bandwidth (itself included) and is restored only if its neighbourhood is
// well attended (float32 compare, as numpy compares a float32 with a Python float) or dense ----
constexpr int NMS_T = 1024;
// (batched: one workgroup per mesh; `order` holds indices LOCAL to the mesh)
__global__ __launch_bounds__(NMS_T) void nms_greedy_kernel(const double* __restrict__ pts_all, const float* __restrict__ attn_all,
                                                           const int* __restrict__ ptr, int n_all,
                                                           const double* __restrict__ bandwidth_all, const int* __restrict__ order_all,
                                                           double thrd_density, float thrd_attn, unsigned char* __restrict__ alive_all) {
    __shared__ int s_cnt[NMS_T / 64];
    __shared__ float s_att[NMS_T / 64];
    __shared__ int s_first[NMS_T / 64];
    int s0, e0;
    mesh_range(ptr, n_all, s0, e0);
    const int n = e0 - s0;
    const double* pts = pts_all + (size_t)s0 * 3;
    const float* attn = attn_all + s0;
    const int* order = order_all + s0;
    unsigned char* alive = alive_all + s0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double T = sqrt_le_threshold(bandwidth_all[blockIdx.y]);
    for (int r = tid; r < n; r += NMS_T) alive[r] = 1;
    __syncthreads();
    // Only a few dozen of the n visits find their point still alive. The next one is found NMS_T order entries at a time (one per
    // thread; the flags only change inside a visit) instead of one entry per barrier pair.
    int s = 0;
    while (s < n) {
        const int cand = s + tid;
        const bool al = cand < n && alive[order[cand]] != 0;
        const unsigned long long m = __ballot(al);
        if (lane == 0) s_first[wv] = m ? wv * 64 + (__ffsll((long long)m) - 1) : NMS_T;
        __syncthreads();
        int first = NMS_T;
#pragma unroll
        for (int q = 0; q < NMS_T / 64; ++q) first = min(first, s_first[q]);
        __syncthreads();                                   // s_first is rewritten by the next round
        if (first == NMS_T) { s += NMS_T; continue; }      // block-uniform
        s += first;
        const int i = order[s];
        const double cx = pts[(size_t)i * 3], cy = pts[(size_t)i * 3 + 1], cz = pts[(size_t)i * 3 + 2];
        int c = 0; float am = -INFINITY;
        for (int r = tid; r < n; r += NMS_T) {
            if (sqdist3d(pts[(size_t)r * 3], pts[(size_t)r * 3 + 1], pts[(size_t)r * 3 + 2], cx, cy, cz) <= T) {
                ++c; am = fmaxf(am, attn[r]); alive[r] = 0;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); am = fmaxf(am, __shfl_xor(am, o)); }
        if (lane == 0) { s_cnt[wv] = c; s_att[wv] = am; }
        __syncthreads();
        if (tid == 0) {
            int ct = 0; float at = -INFINITY;
            for (int q = 0; q < NMS_T / 64; ++q) { ct += s_cnt[q]; at = fmaxf(at, s_att[q]); }
            if (at > thrd_attn || (double)ct / (double)n > thrd_density) alive[i] = 1;
        }
        __syncthreads();
        s += 1;
    }
}

}  // namespace morig

using namespace morig;

extern "C" int morig_inside_check(const double* pts, int32_t n, const uint8_t* vox88, const double* translate, double scale,
                                  double dims0, uint8_t* keep, void* stream) {
    if (!pts || !vox88 || !translate || !keep || n < 0 || !(scale != 0.0)) return MORIG_E_INVALID;
    if (n == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    hipLaunchKernelGGL(inside_check_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, pts, n, vox88, translate[0], translate[1], translate[2],
                       scale, dims0, keep);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

// four rows per workgroup (16-bit bins: sets of at most 65535 points); MORIG_KTH_ROWS=1 keeps the one-row kernel (A/B switch)
static bool kth_four_rows(int max_n) {
    static const int one = [] { const char* e = getenv("MORIG_KTH_ROWS"); return e && atoi(e) == 1 ? 1 : 0; }();
    return !one && max_n <= 65535;
}

extern "C" int morig_knn_bandwidth(const double* pts, int32_t n, int32_t k, double* kth_ws, double* bandwidth, void* stream) {
    if (!pts || !kth_ws || !bandwidth || n <= 0 || k < 1 || k > n) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    if (kth_four_rows(n)) hipLaunchKernelGGL(kth_nn4_kernel, dim3(cdiv(n, K4_R), 1), dim3(256), 0, s, pts, nullptr, n, k, -1.0, kth_ws);
    else hipLaunchKernelGGL(kth_nn_kernel<true>, dim3(n, 1), dim3(256), 0, s, pts, nullptr, n, k, -1.0, kth_ws);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_f64_kernel, dim3(1, 1), dim3(256), 0, s, kth_ws, nullptr, n, bandwidth);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

static int meanshift_run(const double* pts, const float* weights, const int* ptr, int n_all, int n_meshes, int max_n,
                         const double* bandwidth, int max_iter, double* buf_a, double* buf_b, double* state, int32_t* result_in_a,
                         hipStream_t s) {
    // per mesh: state[0] = 1e20 (the reference's diff = 1e10, squared), state[1 .. max_iter-1] = 0
    hipLaunchKernelGGL(meanshift_state_init_kernel, dim3(cdiv((long)max_iter * n_meshes, 256)), dim3(256), 0, s, state, max_iter, n_meshes);
    MORIG_LAUNCH_CHECK();
    MORIG_HIP_TRY(hipMemcpyAsync(buf_a, pts, sizeof(double) * 3 * (size_t)n_all, hipMemcpyDeviceToDevice, s));
    double* cur = buf_a; double* nxt = buf_b;
    for (int t = 1; t < max_iter; ++t) {                  // num_iter runs 1 .. max_iter - 1 (cluster_utils.py:23)
        hipLaunchKernelGGL(meanshift_step_kernel, dim3(cdiv(max_n, MS_TGT), n_meshes), dim3(MS_TILE), 0, s, cur, weights, ptr, n_all,
                           bandwidth, t, max_iter, state, nxt);
        MORIG_LAUNCH_CHECK();
        double* tmp = cur; cur = nxt; nxt = tmp;
    }
    *result_in_a = (cur == buf_a) ? 1 : 0;
    return MORIG_OK;
}

extern "C" int morig_meanshift(const double* pts, const float* weights, int32_t n, const double* bandwidth, int32_t max_iter,
                               double* buf_a, double* buf_b, double* state, int32_t* result_in_a, void* stream) {
    if (!pts || !bandwidth || !buf_a || !buf_b || !state || !result_in_a || n <= 0 || max_iter < 1) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    return meanshift_run(pts, weights, nullptr, n, 1, n, bandwidth, max_iter, buf_a, buf_b, state, result_in_a, s);
}

extern "C" int morig_nms_counts(const double* pts, int32_t n, const double* bandwidth, int32_t* counts, void* stream) {
    if (!pts || !bandwidth || !counts || n <= 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    hipLaunchKernelGGL(nms_counts_kernel, dim3(cdiv(n, 256), 1), dim3(256), 0, s, pts, nullptr, n, bandwidth, counts);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_nms_greedy(const double* pts, const float* attn, int32_t n, const double* bandwidth, const int32_t* order,
                                double thrd_density, float thrd_attn, uint8_t* alive, void* stream) {
    if (!pts || !attn || !bandwidth || !order || !alive || n <= 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(1, 1), dim3(NMS_T), 0, s, pts, attn, nullptr, n, bandwidth, order, thrd_density, thrd_attn, alive);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

// ---- batched over meshes (evaluate/eval_rigging.py:80-95 is a loop over models; here every stage takes all point sets at once) ----
extern "C" int morig_knn_bandwidth_batched(const double* pts, const int32_t* ptr, int32_t n_meshes, int32_t n_all, int32_t max_n,
                                           double quantile, double* kth_ws, double* bandwidth, void* stream) {
    if (!pts || !ptr || !kth_ws || !bandwidth || n_meshes <= 0 || n_all <= 0 || max_n <= 0 || !(quantile >= 0.0)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    if (kth_four_rows(max_n))
        hipLaunchKernelGGL(kth_nn4_kernel, dim3(cdiv(max_n, K4_R), n_meshes), dim3(256), 0, s, pts, ptr, n_all, 0, quantile, kth_ws);
    else hipLaunchKernelGGL(kth_nn_kernel<true>, dim3(max_n, n_meshes), dim3(256), 0, s, pts, ptr, n_all, 0, quantile, kth_ws);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_f64_kernel, dim3(1, n_meshes), dim3(256), 0, s, kth_ws, ptr, n_all, bandwidth);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_meanshift_batched(const double* pts, const float* weights, const int32_t* ptr, int32_t n_meshes, int32_t n_all,
                                       int32_t max_n, const double* bandwidth, int32_t max_iter, double* buf_a, double* buf_b,
                                       double* state, int32_t* result_in_a, void* stream) {
    if (!pts || !ptr || !bandwidth || !buf_a || !buf_b || !state || !result_in_a || n_meshes <= 0 || n_all <= 0 || max_n <= 0 || max_iter < 1)
        return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    return meanshift_run(pts, weights, ptr, n_all, n_meshes, max_n, bandwidth, max_iter, buf_a, buf_b, state, result_in_a, s);
}

extern "C" int morig_nms_counts_batched(const double* pts, const int32_t* ptr, int32_t n_meshes, int32_t n_all, int32_t max_n,
                                        const double* bandwidth, int32_t* counts, void* stream) {
    if (!pts || !ptr || !bandwidth || !counts || n_meshes <= 0 || n_all <= 0 || max_n <= 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    hipLaunchKernelGGL(nms_counts_kernel, dim3(cdiv(max_n, 256), n_meshes), dim3(256), 0, s, pts, ptr, n_all, bandwidth, counts);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

// counts over Morton-sorted point sets (the order morig_meanshift_sorted works in): box culling, see nms_counts_box_kernel
extern "C" int morig_nms_counts_sorted(const double* pts, const int32_t* ptr, int32_t n_meshes, int32_t n_all, int32_t max_n,
                                       const double* bandwidth, double* bbox_ws, int32_t* counts, void* stream) {
    if (!pts || !ptr || !bandwidth || !bbox_ws || !counts || n_meshes <= 0 || n_all <= 0 || max_n <= 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    const int max_boxes = cdiv(max_n, MSB);
    hipLaunchKernelGGL(box32_kernel, dim3(cdiv(max_n, 256), n_meshes), dim3(256), 0, s, pts, ptr, n_all, max_boxes, bbox_ws);
    MORIG_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_counts_box_kernel, dim3(max_boxes, n_meshes), dim3(256), 0, s, pts, ptr, n_all, bandwidth, bbox_ws, max_boxes, counts);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_nms_greedy_batched(const double* pts, const float* attn, const int32_t* ptr, int32_t n_meshes, int32_t n_all,
                                        const double* bandwidth, const int32_t* order_local, double thrd_density, float thrd_attn,
                                        uint8_t* alive, void* stream) {
    if (!pts || !attn || !ptr || !bandwidth || !order_local || !alive || n_meshes <= 0 || n_all <= 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(1, n_meshes), dim3(NMS_T), 0, s, pts, attn, ptr, n_all, bandwidth, order_local, thrd_density,
                       thrd_attn, alive);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_morton_keys(const double* pts, const int32_t* ptr, int32_t n_meshes, int32_t n_all, int64_t* keys, void* stream) {
    if (!pts || !ptr || !keys || n_meshes <= 0 || n_all <= 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    hipLaunchKernelGGL(morton_keys_kernel, dim3(cdiv(n_all, 256)), dim3(256), 0, s, pts, ptr, n_meshes, n_all, reinterpret_cast<long long*>(keys));
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_meanshift_sorted(const double* pts, const float* weights, const int32_t* ptr, int32_t n_meshes, int32_t n_all,
                                      int32_t max_n, const double* bandwidth, int32_t max_iter, double* buf_a, double* buf_b,
                                      double* state, double* bbox_ws, int32_t* result_in_a, void* stream) {
    if (!pts || !ptr || !bandwidth || !buf_a || !buf_b || !state || !bbox_ws || !result_in_a || n_meshes <= 0 || n_all <= 0 || max_n <= 0 ||
        max_iter < 1) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_JOINTS, s, 0.0, 0.0);
    static const int tiles = [] { const char* e = getenv("MORIG_MS_TILES"); return e && atoi(e) == 1 ? 1 : 0; }();   // A/B switch
    const int max_tiles = cdiv(max_n, MS_TILE), max_boxes = cdiv(max_n, MSB);
    hipLaunchKernelGGL(meanshift_state_init_kernel, dim3(cdiv((long)max_iter * n_meshes, 256)), dim3(256), 0, s, state, max_iter, n_meshes);
    MORIG_LAUNCH_CHECK();
    MORIG_HIP_TRY(hipMemcpyAsync(buf_a, pts, sizeof(double) * 3 * (size_t)n_all, hipMemcpyDeviceToDevice, s));
    double* cur = buf_a; double* nxt = buf_b;
    if (tiles) {
        for (int t = 1; t < max_iter; ++t) {
            hipLaunchKernelGGL(tile_bbox_kernel, dim3(max_tiles, n_meshes), dim3(MS_TILE), 0, s, cur, ptr, n_all, max_tiles, bbox_ws);
            MORIG_LAUNCH_CHECK();
            hipLaunchKernelGGL(meanshift_step_sorted_kernel, dim3(cdiv(max_n, MS_TGT), n_meshes), dim3(MS_TILE), 0, s, cur, weights, ptr, n_all,
                               bandwidth, t, max_iter, bbox_ws, max_tiles, state, nxt);
            MORIG_LAUNCH_CHECK();
            double* tmp = cur; cur = nxt; nxt = tmp;
        }
    } else {
        double* box_cur = bbox_ws; double* box_nxt = bbox_ws + (size_t)n_meshes * max_boxes * 6;
        if (max_iter > 1) {
            hipLaunchKernelGGL(box32_kernel, dim3(cdiv(max_n, 256), n_meshes), dim3(256), 0, s, cur, ptr, n_all, max_boxes, box_cur);
            MORIG_LAUNCH_CHECK();
        }
        for (int t = 1; t < max_iter; ++t) {
            hipLaunchKernelGGL(meanshift_step_box_kernel, dim3(max_boxes, n_meshes), dim3(256), 0, s, cur, weights, ptr, n_all, bandwidth, t,
                               max_iter, box_cur, box_nxt, max_boxes, state, nxt);
            MORIG_LAUNCH_CHECK();
            double* tmp = cur; cur = nxt; nxt = tmp;
            tmp = box_cur; box_cur = box_nxt; box_nxt = tmp;
        }
    }
    *result_in_a = (cur == buf_a) ? 1 : 0;
    return MORIG_OK;
}
