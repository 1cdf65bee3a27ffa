// CorrNet point-branch operators (models/corrnet.py:50-73, models/basic_modules.py:66-138): farthest point
// sampling, ball query, k-NN (k<=3) inverse-distance interpolation, row gather (cosine k-NN: deform.hip).
// All are distance scans over one cloud at a time: HBM/LDS-bound integer+fp32 work, no GEMM shape.
//
// Squared distances are evaluated as ((dx*dx + dy*dy) + dz*dz) with every product and sum rounded to
// fp32 separately (__fmul_rn/__fadd_rn, no FMA contraction), so arg-max / arg-min / "< r^2" decisions
// are bit-identical to the fp32 restatement they are tested against (ties -> lowest index).
#include "common.h"

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

// (x - y)^2 summed left to right with every product and sum rounded: the value the CPU restatement (oracle/pyg_primitives.py) and
// torch's (diff * diff).sum(-1) produce. hipcc contracts a * b + c into an fma by default, and __fmul_rn / __fadd_rn do NOT stop
// it (header functions with plain operators, compiled under the default): plain operators under the pragma do.
// [r02: the scalar callers were being contracted until now; the packed FPS / k-NN loops never were]
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

// Wave-wide max of a non-negative float / min of an int WITHOUT the LDS crossbar: 4 DPP steps give every lane its
// 16-lane row's result (quad swaps, half-row mirror, row mirror), 4 readlanes + scalar ops combine the rows.
// (__shfl_xor is a ds_bpermute, ~100 cycles of latency per level: 12 dependent levels per FPS sample.)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = __float_as_int(v);                              // v >= 0: integer order == float order
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false));    // row_half_mirror
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false));    // row_mirror
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const int c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return __int_as_float(max(max(a, b), max(c, d)));
}
__device__ __forceinline__ int wave_min_int(int x) {
    x = min(x, __builtin_amdgcn_update_dpp(0x7fffffff, x, 0xB1, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(0x7fffffff, x, 0x4E, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(0x7fffffff, x, 0x141, 0xF, 0xF, false));
    x = min(x, __builtin_amdgcn_update_dpp(0x7fffffff, x, 0x140, 0xF, 0xF, false));
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const int c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return min(min(a, b), min(c, d));
}

// ---------------------------------------------------------------------------------------------------
// Farthest point sampling: one 1024-thread workgroup per cloud; every thread keeps PPT points and their
// running min-distance in registers; per sample one block-wide arg-max (value desc, index asc).
// ---------------------------------------------------------------------------------------------------
constexpr int FPS_T = 1024;
#ifndef MORIG_FPS_T_DEFAULT
#define MORIG_FPS_T_DEFAULT 1024
#endif

template <int PPT>
__global__ __launch_bounds__(FPS_T) void fps_kernel(const float* __restrict__ pos, int ldp, const int* __restrict__ ptr,
                                                    const int* __restrict__ out_ptr, const int* __restrict__ start,
                                                    int* __restrict__ idx_out) {
    __shared__ float s_val[FPS_T / 64];
    __shared__ int s_idx[FPS_T / 64];
    __shared__ float s_cur[3];
    const int b = blockIdx.x;
    const int p0 = ptr[b], n = ptr[b + 1] - p0;
    const int o0 = out_ptr[b], m = out_ptr[b + 1] - o0;
    if (n <= 0 || m <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = tid + i * FPS_T;
        if (j < n) {
            const float* q = pos + (size_t)(p0 + j) * ldp;
            px[i] = q[0]; py[i] = q[1]; pz[i] = q[2];
            dist[i] = INFINITY;
        } else { px[i] = py[i] = pz[i] = 0.f; dist[i] = -1.f; }        // never selected
    }
    int cur = start ? start[b] : 0;
    if (cur < 0 || cur >= n) cur = 0;
    for (int s = 0; s < m; ++s) {
        // the thread that holds point `cur` in registers broadcasts its coordinates through LDS (a global load
        // of the selected point would put an L2 round trip on the critical path of every one of the m steps)
        if (tid == (cur & (FPS_T - 1))) {
            const int slot = cur / FPS_T;
            float cx = px[0], cy = py[0], cz = pz[0];
#pragma unroll
            for (int i = 1; i < PPT; ++i) if (slot == i) { cx = px[i]; cy = py[i]; cz = pz[i]; }
            s_cur[0] = cx; s_cur[1] = cy; s_cur[2] = cz;
            idx_out[o0 + s] = p0 + cur;
        }
        __syncthreads();
        if (s + 1 == m) break;
        const float cx = s_cur[0], cy = s_cur[1], cz = s_cur[2];
        float bv = 0.f; int bi = 0x7fffffff;               // distances are >= 0; lanes without points keep (0, INT_MAX)
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int j = tid + i * FPS_T;
            if (j < n) {
                const float d = sqdist3(px[i], py[i], pz[i], cx, cy, cz);
                dist[i] = fminf(dist[i], d);
                if (dist[i] > bv || bi == 0x7fffffff) { bv = dist[i]; bi = j; }   // j ascending: first max kept
            }
        }
        // arg-max as (max value, then lowest index among the lanes that hold it), both wave-wide via DPP
        const float wv = wave_max_nonneg(bv);
        const int wi = wave_min_int(bv == wv ? bi : 0x7fffffff);
        if (lane == 0) { s_val[w] = wv; s_idx[w] = wi; }
        __syncthreads();
        // every wave combines the 16 per-wave candidates itself: no third barrier
        const float v = lane < FPS_T / 64 ? s_val[lane] : 0.f;
        const int ix = lane < FPS_T / 64 ? s_idx[lane] : 0x7fffffff;
        const float gv = wave_max_nonneg(v);
        cur = wave_min_int((v == gv) ? ix : 0x7fffffff);
    }
}

// ---------------------------------------------------------------------------------------------------
// Clouds of up to 8192 points (every CorrNet level at the benchmark sizes). A cloud's scan is bound by the VALU rate of
// the ONE CU it runs on (~12 ops per point per sample in the kernel above), so this variant spends fewer instructions:
//   * coordinates of all points also sit in LDS (96 KB): the selected point is a broadcast LDS read, no owner thread,
//     no barrier for the hand-over -- ONE barrier per sample;
//   * the distance update runs on pairs of points in packed fp32 math (v_pk_add/mul_f32 round every component exactly
//     like the scalar ops; contraction is off, so the value is sqdist3's);
//   * threads track only their max VALUE (max3); the index is worked out by the few lanes that hold the wave's max and
//     merged across waves by one 64-bit LDS atomic max on (value bits, ~index): value descending, index ascending.
// Three key slots rotate so a slot is cleared one barrier after its last read and one barrier before its next use.
// ---------------------------------------------------------------------------------------------------
typedef float fps_f2 __attribute__((ext_vector_type(2)));

template <int PPT>                                        // even, <= 8
__global__ __launch_bounds__(FPS_T) void fps_lds_kernel(const float* __restrict__ pos, int ldp, const int* __restrict__ ptr,
                                                        const int* __restrict__ out_ptr, const int* __restrict__ start,
                                                        int* __restrict__ idx_out) {
    constexpr int NP = PPT * FPS_T, H = PPT / 2;
    __shared__ float sx[NP], sy[NP], sz[NP];
    __shared__ unsigned long long s_key[3];
    const int b = blockIdx.x;
    const int p0 = ptr[b], n = ptr[b + 1] - p0;
    const int o0 = out_ptr[b], m = out_ptr[b + 1] - o0;
    if (n <= 0 || m <= 0) return;
    const int tid = threadIdx.x;
    fps_f2 px[H], py[H], pz[H], dist[H];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = tid + i * FPS_T;
        float x = 0.f, y = 0.f, z = 0.f, d = -1.f;                     // d = -1: never selected
        if (j < n) {
            const float* q = pos + (size_t)(p0 + j) * ldp;
            x = q[0]; y = q[1]; z = q[2]; d = INFINITY;
        }
        px[i >> 1][i & 1] = x; py[i >> 1][i & 1] = y; pz[i >> 1][i & 1] = z; dist[i >> 1][i & 1] = d;
        sx[j] = x; sy[j] = y; sz[j] = z;
    }
    if (tid < 3) s_key[tid] = 0ull;
    int cur = start ? start[b] : 0;
    if (cur < 0 || cur >= n) cur = 0;
    __syncthreads();
    for (int s = 0; s < m; ++s) {
        if (tid == 0) { idx_out[o0 + s] = p0 + cur; s_key[(s + 1) % 3] = 0ull; }
        if (s + 1 == m) break;
        const float cx = sx[cur], cy = sy[cur], cz = sz[cur];
        const fps_f2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
        float best = -1.f;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const fps_f2 dx = px[h] - c2x, dy = py[h] - c2y, dz = pz[h] - c2z;
                const fps_f2 d = (dx * dx + dy * dy) + dz * dz;
                dist[h][0] = fminf(dist[h][0], d[0]); dist[h][1] = fminf(dist[h][1], d[1]);
                best = fmaxf(best, fmaxf(dist[h][0], dist[h][1]));
            }
        }
        const float wv = wave_max_nonneg(best);            // int-ordered max: correct as soon as one lane is >= 0
        if (best == wv && wv >= 0.f) {
            unsigned loc = 0x7fffffffu;
#pragma unroll
            for (int i = PPT - 1; i >= 0; --i) if (dist[i >> 1][i & 1] == wv) loc = (unsigned)(tid + i * FPS_T);
            atomicMax(&s_key[s % 3], ((unsigned long long)__float_as_uint(wv) << 32) | (unsigned long long)(0xffffffffu - loc));
        }
        __syncthreads();
        cur = (int)(0xffffffffu - (unsigned)(s_key[s % 3] & 0xffffffffull));
    }
}

// ---------------------------------------------------------------------------------------------------
// [r05] Bucketed variant: the SAME sample sequence with most of the distance updates skipped. A sample changes dist[p] only where
// |p - c|^2 < dist[p]; after the first few dozen samples that is a small neighbourhood of c, yet the kernel above still updates all
// 8192 points per sample (VALU-bound: ~0.9 us per sample on the one CU a cloud owns, 3.6 ms for SA1's 4096 samples). Here the points
// are first counting-sorted by the Morton code of an 8 x 8 x 8 grid over the cloud's bounding box (LDS, once), so that the 128 points
// a wave holds in one register PAIR (2 x 64 lanes) are neighbours: a bucket. Every bucket keeps its bounding box and its current
// max min-distance (lanes 0..H-1 of the wave); per sample, lanes 0..H-1 evaluate the lower bound of |p - c|^2 over their box -- with
// the per-point operation order, so by monotonicity of rounding it is <= every point's computed value -- and only buckets with
// bound < max are updated (exactly: a skipped bucket provably changes nothing). A wave without updates re-submits its cached
// candidate. Arg-max ties go to the lowest ORIGINAL index, as before: key = (value, ~orig index, sorted position).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned morton3(unsigned x, unsigned y, unsigned z) {       // 3 bits each -> 9 bits
    auto spread = [](unsigned v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4); };
    return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}
__device__ __forceinline__ float wave_min_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// BT [r06]: threads per cloud. 1024 = 16 waves (4 per SIMD, 8 points per lane at 8192 points); 256 = ONE wave per SIMD with 32 points per lane:
// the per-sample chain then meets a 4-wave barrier and no wave shares its SIMD's issue port (MORIG_BT selects, see morig_fps)
template <int PPT, int BT = FPS_T>                        // PPT even, PPT * BT <= 8192
__global__ __launch_bounds__(BT) void fps_bkt_kernel(const float* __restrict__ pos, int ldp, const int* __restrict__ ptr,
                                                        const int* __restrict__ out_ptr, const int* __restrict__ start,
                                                        int* __restrict__ idx_out) {
    constexpr int NP = PPT * BT, H = PPT / 2, NCELL = 512;
    __shared__ float sx[NP], sy[NP], sz[NP];              // coordinates in SORTED order
    __shared__ unsigned short sorig[NP];                  // original index of a sorted position
    __shared__ int s_hist[NCELL + 1];
    __shared__ float s_box[6 * (BT / 64)];
    __shared__ unsigned long long s_key[3];
    __shared__ int s_start;
    const int b = blockIdx.x;
    const int p0 = ptr[b], n = ptr[b + 1] - p0;
    const int o0 = out_ptr[b], m = out_ptr[b + 1] - o0;
    if (n <= 0 || m <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int cur0 = start ? start[b] : 0;
    if (cur0 < 0 || cur0 >= n) cur0 = 0;

    // ---- cloud bounding box ----
    float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
    float qx[PPT], qy[PPT], qz[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = tid + i * BT;
        qx[i] = qy[i] = qz[i] = 0.f;
        if (j < n) {
            const float* q = pos + (size_t)(p0 + j) * ldp;
            qx[i] = q[0]; qy[i] = q[1]; qz[i] = q[2];
            lx = fminf(lx, qx[i]); ly = fminf(ly, qy[i]); lz = fminf(lz, qz[i]);
            hx = fmaxf(hx, qx[i]); hy = fmaxf(hy, qy[i]); hz = fmaxf(hz, qz[i]);
        }
    }
    lx = wave_min_f(lx); ly = wave_min_f(ly); lz = wave_min_f(lz); hx = wave_max_f(hx); hy = wave_max_f(hy); hz = wave_max_f(hz);
    if (lane == 0) { s_box[w * 6] = lx; s_box[w * 6 + 1] = ly; s_box[w * 6 + 2] = lz; s_box[w * 6 + 3] = hx; s_box[w * 6 + 4] = hy; s_box[w * 6 + 5] = hz; }
    for (int i = tid; i <= NCELL; i += BT) s_hist[i] = 0;
    if (tid < 3) s_key[tid] = 0ull;
    __syncthreads();
    for (int k = 0; k < BT / 64; ++k) {
        lx = fminf(lx, s_box[k * 6]); ly = fminf(ly, s_box[k * 6 + 1]); lz = fminf(lz, s_box[k * 6 + 2]);
        hx = fmaxf(hx, s_box[k * 6 + 3]); hy = fmaxf(hy, s_box[k * 6 + 4]); hz = fmaxf(hz, s_box[k * 6 + 5]);
    }
    // cell of a point: any monotone map works (the grid only decides the ORDER of the points, never a result); non-finite
    // coordinates land in cell 0
    const float gx = 8.f / fmaxf(hx - lx, 1e-30f), gy = 8.f / fmaxf(hy - ly, 1e-30f), gz = 8.f / fmaxf(hz - lz, 1e-30f);
    auto cell_of = [&](float x, float y, float z) {
        const int cx = min(max((int)((x - lx) * gx), 0), 7), cy = min(max((int)((y - ly) * gy), 0), 7), cz = min(max((int)((z - lz) * gz), 0), 7);
        return (int)morton3((unsigned)cx, (unsigned)cy, (unsigned)cz);
    };
    // ---- counting sort by cell: histogram, exclusive scan (wave 0), scatter ----
    int cell[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = tid + i * BT;
        cell[i] = j < n ? cell_of(qx[i], qy[i], qz[i]) : -1;
        if (j < n) atomicAdd(&s_hist[cell[i]], 1);
    }
    __syncthreads();
    if (w == 0) {                                          // 512 counters: 8 per lane, then a wave scan of the lane sums
        int c[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { c[k] = s_hist[lane * 8 + k]; sum += c[k]; }
        int incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        int run = incl - sum;
#pragma unroll
        for (int k = 0; k < 8; ++k) { s_hist[lane * 8 + k] = run; run += c[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = tid + i * BT;
        if (j < n) {
            const int d = atomicAdd(&s_hist[cell[i]], 1);
            sx[d] = qx[i]; sy[d] = qy[i]; sz[d] = qz[i]; sorig[d] = (unsigned short)j;
            if (j == cur0) s_start = d;
        }
    }
    __syncthreads();
    // ---- this thread's points: sorted positions w * 64 * PPT + i * 64 + lane (register pair h = positions of i = 2h, 2h + 1) ----
    fps_f2 px[H], py[H], pz[H], dist[H];
    unsigned short org[PPT];
    float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY, pmax = -1.f;   // lane h: bucket h
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float l0 = INFINITY, l1 = INFINITY, l2 = INFINITY, h0 = -INFINITY, h1 = -INFINITY, h2 = -INFINITY;
        bool any = false;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = w * 64 * PPT + (2 * h + e) * 64 + lane;
            float x = 0.f, y = 0.f, z = 0.f, dd = -1.f;                 // dd = -1: never selected
            unsigned short og = 0xffff;
            if (d < n) {
                x = sx[d]; y = sy[d]; z = sz[d]; og = sorig[d]; dd = INFINITY; any = true;
                l0 = fminf(l0, x); l1 = fminf(l1, y); l2 = fminf(l2, z); h0 = fmaxf(h0, x); h1 = fmaxf(h1, y); h2 = fmaxf(h2, z);
            }
            px[h][e] = x; py[h][e] = y; pz[h][e] = z; dist[h][e] = dd; org[2 * h + e] = og;
        }
        l0 = wave_min_f(l0); l1 = wave_min_f(l1); l2 = wave_min_f(l2); h0 = wave_max_f(h0); h1 = wave_max_f(h1); h2 = wave_max_f(h2);
        const bool live = __ballot(any) != 0ull;
        if (lane == h) { bx0 = l0; by0 = l1; bz0 = l2; bx1 = h0; by1 = h1; bz1 = h2; pmax = live ? INFINITY : -1.f; }
    }
    // per lane: the low key word of the best point THIS LANE holds among the wave's maxima (0: none), and the wave's max value; both
    // survive the samples in which the wave updates nothing (the lanes that hold the maximum re-submit it)
    unsigned mykey = 0u;
    float wv = -1.f;
    int cur = s_start;
    for (int s = 0; s < m; ++s) {
        if (tid == 0) { idx_out[o0 + s] = p0 + (int)sorig[cur]; s_key[(s + 1) % 3] = 0ull; }
        if (s + 1 == m) break;
        const float cx = sx[cur], cy = sy[cur], cz = sz[cur];
        // lower bound of |p - c|^2 over bucket `lane`'s box, in the per-point operation order
        unsigned long long need;
        {
#pragma clang fp contract(off)
            const float ex = fmaxf(fmaxf(bx0 - cx, cx - bx1), 0.f), ey = fmaxf(fmaxf(by0 - cy, cy - by1), 0.f), ez = fmaxf(fmaxf(bz0 - cz, cz - bz1), 0.f);
            const float lb = (ex * ex + ey * ey) + ez * ez;
            need = __ballot(lane < H && lb < pmax);
        }
        if (need != 0ull) {                                // wave-uniform
            const fps_f2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
            // [r06] only the buckets that NEED the update are touched: their distances, their wave maximum (one DPP chain each) and
            // their record in lane h. (The round-5 form reduced all H pair maxima and compared all PPT points whenever the wave had
            // one bucket to update; measured on the thread-count sweep, profiles/r06j_*: the per-sample time follows this per-lane
            // work -- 4.4 / 6.2 / 9.7 ms at 8 / 16 / 32 points per lane -- not the barrier.) An untouched bucket's record is already its
            // maximum, so the results are the same bits.
#pragma unroll
            for (int h = 0; h < H; ++h) {
                if (need & (1ull << h)) {                  // wave-uniform
#pragma clang fp contract(off)
                    const fps_f2 dx = px[h] - c2x, dy = py[h] - c2y, dz = pz[h] - c2z;
                    const fps_f2 d = (dx * dx + dy * dy) + dz * dz;
                    dist[h][0] = fminf(dist[h][0], d[0]); dist[h][1] = fminf(dist[h][1], d[1]);
                    int b = __float_as_int(fmaxf(fmaxf(dist[h][0], dist[h][1]), 0.f));     // >= 0 as int bits (lanes without a point: 0)
                    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, false));
                    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, false));
                    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, false));
                    b = max(b, __builtin_amdgcn_update_dpp(0, b, 0x140, 0xF, 0xF, false));
                    const int m4 = max(max(__builtin_amdgcn_readlane(b, 0), __builtin_amdgcn_readlane(b, 16)),
                                       max(__builtin_amdgcn_readlane(b, 32), __builtin_amdgcn_readlane(b, 48)));
                    if (lane == h) pmax = __int_as_float(m4);                              // (a bucket that needs an update has points)
                }
            }
            // the wave's maximum = the largest bucket record (empty buckets hold -1.0: negative as int), and this lane's best point there
            int wvi = -1;
            int pm[H];
#pragma unroll
            for (int h = 0; h < H; ++h) { pm[h] = __builtin_amdgcn_readlane(__float_as_int(pmax), h); wvi = max(wvi, pm[h]); }
            wv = wvi >= 0 ? __int_as_float(wvi) : -1.f;
            mykey = 0u;                                    // valid bit | (~orig & 0x1fff) << 13 | sorted position: larger = lower original index
            if (wvi >= 0) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    if (pm[h] == wvi) {                    // wave-uniform: only a bucket whose record IS the maximum can hold it
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int i = 2 * h + e;
                            if (dist[h][e] == wv)
                                mykey = max(mykey, (1u << 26) | ((0x1fffu - (unsigned)org[i]) << 13) | (unsigned)(w * 64 * PPT + i * 64 + lane));
                        }
                    }
                }
            }
        }
        if (mykey != 0u) atomicMax(&s_key[s % 3], ((unsigned long long)__float_as_uint(wv) << 32) | (unsigned long long)mykey);
        __syncthreads();
        cur = (int)((unsigned)(s_key[s % 3] & 0xffffffffull) & 0x1fffu);
    }
}

// ---------------------------------------------------------------------------------------------------
// Ball query (torch_cluster.radius CUDA semantics): one wave per centre scans its cloud in index order,
// 64 points per step; ballot + popcount keeps the first `max_nbrs` hits with d^2 < r^2 in order.
// Writes an int64 COO (row 0 = source point, row 1 = centre), unused slots = -1, ready for morig_csr_build.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ x, int ldx, const int* __restrict__ ptr_x,
                                                         const float* __restrict__ y, int ldy, const int* __restrict__ ptr_y,
                                                         int n_clouds, int n_centres, float r2, int max_nbrs,
                                                         int64_t* __restrict__ coo) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= n_centres) return;
    // cloud of this centre: binary search in ptr_y
    int lo = 0, hi = n_clouds;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ptr_y[mid] <= k) lo = mid; else hi = mid; }
    const int xs = ptr_x[lo], xe = ptr_x[lo + 1];
    const float cx = y[(size_t)k * ldy], cy = y[(size_t)k * ldy + 1], cz = y[(size_t)k * ldy + 2];
    const int64_t E = (int64_t)n_centres * max_nbrs;
    int64_t* src = coo + (int64_t)k * max_nbrs;
    int64_t* dst = coo + E + (int64_t)k * max_nbrs;
    int count = 0;
    for (int base = xs; base < xe && count < max_nbrs; base += 64) {
        const int j = base + lane;
        bool hit = false;
        if (j < xe) {
            const float* q = x + (size_t)j * ldx;
            hit = sqdist3(cx, cy, cz, q[0], q[1], q[2]) < r2;      // (y - x)^2: same value as (x - y)^2
        }
        const unsigned long long mask = __ballot(hit);
        const int rank = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && rank < max_nbrs) { src[rank] = j; dst[rank] = k; }
        count += __popcll(mask);
    }
    if (count > max_nbrs) count = max_nbrs;
    for (int s = count + lane; s < max_nbrs; s += 64) { src[s] = -1; dst[s] = -1; }
}

// ---------------------------------------------------------------------------------------------------
// radius_cpu (models/basic_modules.py:9-29): for each y all x with dist <= r (INCLUSIVE, no batch vector); a row with more
// than max_nbrs hits keeps a uniformly random subset of exactly max_nbrs (the reference draws it with torch.multinomial on
// the 0/1 validity row: every subset equally likely). One wave per y row, 64 candidates per step, Algorithm R reservoir:
// slot j lives in lane j's register (max_nbrs <= 64), hit number t >= max replaces slot u = floor(U * (t + 1)) when
// u < max, U from a counter-based hash of (seed, row, t). Rows with <= max_nbrs hits keep all hits in index order.
// Output: the slot table of ball_query (unused slots -1) + the uncapped hit count per row.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix32(unsigned a) {
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}
__global__ __launch_bounds__(256) void radius_sample_kernel(const float* __restrict__ x, int ldx, int nx,
                                                            const float* __restrict__ y, int ldy, int ny, float r2, int max_nbrs,
                                                            unsigned seed, int64_t* __restrict__ coo, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= ny) return;
    const float cx = y[(size_t)k * ldy], cy = y[(size_t)k * ldy + 1], cz = y[(size_t)k * ldy + 2];
    int slot = -1;                                        // reservoir entry `lane`
    int seen = 0;                                         // hits so far (wave-uniform)
    for (int base = 0; base < nx; base += 64) {
        const int j = base + lane;
        bool hit = false;
        if (j < nx) {
            const float* q = x + (size_t)j * ldx;
            hit = sqdist3(cx, cy, cz, q[0], q[1], q[2]) <= r2;
        }
        unsigned long long mask = __ballot(hit);
        const int nh = __popcll(mask);
        if (nh == 0) continue;
        if (seen + nh <= max_nbrs) {                      // still filling: hit number t goes to slot t
            for (unsigned long long m = mask; m; m &= m - 1ull) {    // slot t receives hit number t (wave-uniform walk)
                const int src_lane = __builtin_ctzll(m);
                const int t = seen + __popcll(mask & ((1ull << src_lane) - 1ull));
                const int idx = base + src_lane;
                if (lane == t) slot = idx;
            }
            seen += nh;
        } else {
            for (unsigned long long m = mask; m; m &= m - 1ull) {      // wave-uniform walk over the hits of this step
                const int src_lane = __builtin_ctzll(m);
                const int idx = base + src_lane;
                const int t = seen;
                if (t < max_nbrs) { if (lane == t) slot = idx; }
                else {
                    const unsigned h = mix32(seed ^ mix32((unsigned)k * 0x9E3779B9u + (unsigned)t));
                    const unsigned u = (unsigned)(((unsigned long long)h * (unsigned long long)(t + 1)) >> 32);   // uniform in [0, t]
                    if ((int)u < max_nbrs && lane == (int)u) slot = idx;
                }
                ++seen;
            }
        }
    }
    const int64_t E = (int64_t)ny * max_nbrs;
    if (lane < max_nbrs) {
        const bool used = lane < seen;
        coo[(int64_t)k * max_nbrs + lane] = used ? slot : -1;
        coo[E + (int64_t)k * max_nbrs + lane] = used ? k : -1;
    }
    if (lane == 0) counts[k] = seen;
}

// ---------------------------------------------------------------------------------------------------
// k-NN (k <= 3) of every target among the sources of its cloud + inverse-squared-distance weights.
// One thread per target; sources staged through LDS in tiles of 1024 points. Order: (d^2, index) ascending.
// ---------------------------------------------------------------------------------------------------
constexpr int KNN_TILE = 1024;
__global__ __launch_bounds__(256) void knn3_kernel(const float* __restrict__ xs, int ldx, const int* __restrict__ ptr_x,
                                                   const float* __restrict__ yt, int ldy, const int* __restrict__ ptr_y,
                                                   int n_clouds, int k, int* __restrict__ idx, float* __restrict__ wgt) {
    // sources in LDS as three coordinate arrays: a pair of adjacent sources is one 8-byte broadcast read per coordinate and the
    // distance of the pair is 8 packed-fp32 instructions (v_pk_add/mul_f32 round each component exactly like the scalar ops of
    // sqdist3, contraction off) instead of 16 scalar ones -- the scan is VALU-bound. An odd tail is padded with a point at
    // infinity (distance +inf: never below the running third-best).
    __shared__ __attribute__((aligned(8))) float sx[KNN_TILE], sy[KNN_TILE], sz[KNN_TILE];
    // blocks are assigned per (cloud, chunk of 256 targets): blockIdx.y = cloud
    const int c = blockIdx.y;
    const int ys = ptr_y[c], ye = ptr_y[c + 1];
    const int x0 = ptr_x[c], x1 = ptr_x[c + 1];
    const int t = ys + blockIdx.x * blockDim.x + threadIdx.x;
    if (ys + (int)(blockIdx.x * blockDim.x) >= ye) return;          // block-uniform
    const bool live = t < ye;
    float tx = 0.f, ty = 0.f, tz = 0.f;
    if (live) { const float* q = yt + (size_t)t * ldy; tx = q[0]; ty = q[1]; tz = q[2]; }
    const fps_f2 t2x = {tx, tx}, t2y = {ty, ty}, t2z = {tz, tz};
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = -1, i1 = -1, i2 = -1;
    auto offer = [&](float d, int j) {
        if (d < d2) {                                // strict: on ties the earlier (lower) index stays ahead
            if (d < d1) {
                d2 = d1; i2 = i1;
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; } else { d1 = d; i1 = j; }
            } else { d2 = d; i2 = j; }
        }
    };
    for (int base = x0; base < x1; base += KNN_TILE) {
        const int cnt = min(KNN_TILE, x1 - base);
        __syncthreads();
        for (int i = threadIdx.x; i < ((cnt + 1) & ~1); i += blockDim.x) {
            float qx = INFINITY, qy = INFINITY, qz = INFINITY;
            if (i < cnt) { const float* q = xs + (size_t)(base + i) * ldx; qx = q[0]; qy = q[1]; qz = q[2]; }
            sx[i] = qx; sy[i] = qy; sz[i] = qz;
        }
        __syncthreads();
        if (live) {
#pragma clang fp contract(off)
            for (int i = 0; i < cnt; i += 2) {
                const fps_f2 dx = *reinterpret_cast<const fps_f2*>(sx + i) - t2x;
                const fps_f2 dy = *reinterpret_cast<const fps_f2*>(sy + i) - t2y;
                const fps_f2 dz = *reinterpret_cast<const fps_f2*>(sz + i) - t2z;
                const fps_f2 d = (dx * dx + dy * dy) + dz * dz;                  // sqdist3's order: (x - y)^2, source minus target
                offer(d[0], base + i);
                offer(d[1], base + i + 1);
            }
        }
    }
    if (!live) return;
    const float dd[3] = {d0, d1, d2};
    const int ii[3] = {i0, i1, i2};
    for (int s = 0; s < 3; ++s) {
        const bool ok = s < k && ii[s] >= 0;
        idx[(size_t)t * 3 + s] = ok ? ii[s] : -1;
        wgt[(size_t)t * 3 + s] = ok ? 1.0f / fmaxf(dd[s], 1e-16f) : 0.f;
    }
}

// out[t][c] = sum_s w_s * x[idx_s][c] / sum_s w_s   (scatter_add order s = 0,1,2; division last)
__global__ __launch_bounds__(256) void knn_interp_kernel(const float* __restrict__ x, int ldx, int C, const int* __restrict__ idx,
                                                         const float* __restrict__ wgt, int n_targets, float* __restrict__ out, int ldo) {
    const int64_t total = (int64_t)n_targets * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t t = i / C; const int c = (int)(i - t * C);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int j = idx[t * 3 + s];
            if (j >= 0) {
                const float w = wgt[t * 3 + s];
                num = __fadd_rn(num, __fmul_rn(x[(size_t)j * ldx + c], w));
                den = __fadd_rn(den, w);
            }
        }
        out[t * ldo + c] = num / den;
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, int lds, const int* __restrict__ idx, int rows, int cols,
                                   float* __restrict__ dst, int ldd) {
    const int64_t total = (int64_t)rows * cols;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / cols; const int c = (int)(i - r * cols);
        const int j = idx[r];
        dst[r * ldd + c] = j >= 0 ? src[(size_t)j * lds + c] : 0.f;
    }
}

}  // namespace morig

using namespace morig;

extern "C" int morig_fps(const float* pos, int32_t ldp, const int32_t* ptr, const int32_t* out_ptr, const int32_t* start,
                         int32_t n_clouds, int32_t max_cloud_points, int32_t* idx_out, void* stream) {
    if (!pos || !ptr || !out_ptr || !idx_out || n_clouds <= 0 || ldp < 3 || max_cloud_points <= 0) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ppt = cdiv(max_cloud_points, FPS_T);
    ProfScope ps(K_FPS, s, 0.0, 0.0);
#define MORIG_FPS_CASE(P) hipLaunchKernelGGL((fps_kernel<P>), dim3(n_clouds), dim3(FPS_T), 0, s, pos, ldp, ptr, out_ptr, start, idx_out)
#define MORIG_FPS_LDS_CASE(P) hipLaunchKernelGGL((fps_lds_kernel<P>), dim3(n_clouds), dim3(FPS_T), 0, s, pos, ldp, ptr, out_ptr, start, idx_out)
    static const bool old_fps = getenv("MORIG_FPS_OLD") != nullptr;
    static const bool bkt = [] { const char* e = getenv("MORIG_FPS_BKT"); return !(e && e[0] == '0'); }();
#define MORIG_FPS_BKT_CASE(P) hipLaunchKernelGGL((fps_bkt_kernel<P>), dim3(n_clouds), dim3(FPS_T), 0, s, pos, ldp, ptr, out_ptr, start, idx_out)
#define MORIG_FPS_BKT_T(P, T) hipLaunchKernelGGL((fps_bkt_kernel<P, T>), dim3(n_clouds), dim3(T), 0, s, pos, ldp, ptr, out_ptr, start, idx_out)
    // threads per cloud of the bucketed kernel: MORIG_FPS_T = 256 | 512 | 1024 (A/B switch)
    static const int fps_t = [] { const char* e = getenv("MORIG_FPS_T"); const int v = e ? atoi(e) : MORIG_FPS_T_DEFAULT; return (v == 256 || v == 512) ? v : 1024; }();
    if (max_cloud_points <= 8192 && !old_fps && bkt && fps_t != 1024) {
        const int pp = cdiv(max_cloud_points, fps_t);
        if (fps_t == 256) {
            if (pp <= 4) MORIG_FPS_BKT_T(4, 256); else if (pp <= 8) MORIG_FPS_BKT_T(8, 256); else if (pp <= 16) MORIG_FPS_BKT_T(16, 256); else MORIG_FPS_BKT_T(32, 256);
        } else {
            if (pp <= 2) MORIG_FPS_BKT_T(2, 512); else if (pp <= 4) MORIG_FPS_BKT_T(4, 512); else if (pp <= 8) MORIG_FPS_BKT_T(8, 512); else MORIG_FPS_BKT_T(16, 512);
        }
    }
    else if (ppt <= 8 && !old_fps && bkt) {
        if (ppt <= 2) MORIG_FPS_BKT_CASE(2);
        else if (ppt <= 4) MORIG_FPS_BKT_CASE(4);
        else MORIG_FPS_BKT_CASE(8);
    }
    else if (ppt <= 8 && !old_fps) {
        if (ppt <= 2) MORIG_FPS_LDS_CASE(2);
        else if (ppt <= 4) MORIG_FPS_LDS_CASE(4);
        else MORIG_FPS_LDS_CASE(8);
    }
    else if (ppt <= 1) MORIG_FPS_CASE(1);
    else if (ppt <= 2) MORIG_FPS_CASE(2);
    else if (ppt <= 4) MORIG_FPS_CASE(4);
    else if (ppt <= 8) MORIG_FPS_CASE(8);
    else if (ppt <= 16) MORIG_FPS_CASE(16);
    else if (ppt <= 32) MORIG_FPS_CASE(32);
    else return MORIG_E_UNSUPPORTED;               // > 32768 points per cloud
#undef MORIG_FPS_CASE
#undef MORIG_FPS_LDS_CASE
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_ball_query(const float* x, int32_t ldx, const int32_t* ptr_x, const float* y, int32_t ldy,
                                const int32_t* ptr_y, int32_t n_clouds, int32_t n_centres, float radius,
                                int32_t max_nbrs, int64_t* coo, void* stream) {
    if (!x || !y || !ptr_x || !ptr_y || !coo || n_clouds <= 0 || n_centres < 0 || max_nbrs <= 0 || ldx < 3 || ldy < 3) return MORIG_E_INVALID;
    if (n_centres == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float r2 = (float)((double)radius * (double)radius);
    ProfScope ps(K_BALL, s, 0.0, 16.0 * n_centres * max_nbrs);
    hipLaunchKernelGGL(ball_query_kernel, dim3(cdiv(n_centres, 4)), dim3(256), 0, s, x, ldx, ptr_x, y, ldy, ptr_y, n_clouds,
                       n_centres, r2, max_nbrs, coo);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_radius_sample(const float* x, int32_t ldx, int32_t nx, const float* y, int32_t ldy, int32_t ny, float radius,
                                   int32_t max_nbrs, uint32_t seed, int64_t* coo, int32_t* counts, void* stream) {
    if (!x || !y || !coo || !counts || nx <= 0 || ny < 0 || ldx < 3 || ldy < 3) return MORIG_E_INVALID;
    if (max_nbrs <= 0 || max_nbrs > 64) return MORIG_E_UNSUPPORTED;
    if (ny == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float r2 = (float)((double)radius * (double)radius);
    ProfScope ps(K_BALL, s, 0.0, 12.0 * ny * (double)nx / 64.0);
    hipLaunchKernelGGL(radius_sample_kernel, dim3(cdiv(ny, 4)), dim3(256), 0, s, x, ldx, nx, y, ldy, ny, r2, max_nbrs, seed, coo, counts);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_knn_search(const float* pos_x, int32_t ldx, const int32_t* ptr_x, const float* pos_y, int32_t ldy,
                                const int32_t* ptr_y, int32_t n_clouds, int32_t n_targets, int32_t max_targets_per_cloud, int32_t k,
                                int32_t* idx, float* wgt, void* stream) {
    if (!pos_x || !pos_y || !ptr_x || !ptr_y || !idx || !wgt) return MORIG_E_INVALID;
    if (k < 1 || k > 3) return MORIG_E_UNSUPPORTED;
    if (n_clouds <= 0 || n_targets < 0 || max_targets_per_cloud <= 0 || ldx < 3 || ldy < 3) return MORIG_E_INVALID;
    if (n_targets == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_KNN_INTERP, s, 0.0, 24.0 * n_targets);
    hipLaunchKernelGGL(knn3_kernel, dim3(cdiv(max_targets_per_cloud, 256), n_clouds), dim3(256), 0, s, pos_x, ldx, ptr_x, pos_y, ldy,
                       ptr_y, n_clouds, k, idx, wgt);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_knn_apply(const float* feat, int32_t ldf, int32_t C, const int32_t* idx, const float* wgt, int32_t n_targets,
                               float* out, int32_t ldo, void* stream) {
    if (!feat || !idx || !wgt || !out || n_targets < 0 || C <= 0 || ldo < C || ldf < C) return MORIG_E_INVALID;
    if (n_targets == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_KNN_INTERP, s, 0.0, 4.0 * n_targets * (double)C * 4.0);
    int64_t blocks = ((int64_t)n_targets * C + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(knn_interp_kernel, dim3((int)blocks), dim3(256), 0, s, feat, ldf, C, idx, wgt, n_targets, out, ldo);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_knn_interpolate(const float* feat, int32_t ldf, int32_t C, const float* pos_x, int32_t ldx,
                                     const int32_t* ptr_x, const float* pos_y, int32_t ldy, const int32_t* ptr_y,
                                     int32_t n_clouds, int32_t n_targets, int32_t max_targets_per_cloud, int32_t k,
                                     int32_t* idx_ws, float* wgt_ws, float* out, int32_t ldo, void* stream) {
    if (!feat || !out || C <= 0 || ldo < C || ldf < C) return MORIG_E_INVALID;
    const int st = morig_knn_search(pos_x, ldx, ptr_x, pos_y, ldy, ptr_y, n_clouds, n_targets, max_targets_per_cloud, k, idx_ws, wgt_ws, stream);
    if (st != MORIG_OK) return st;
    return morig_knn_apply(feat, ldf, C, idx_ws, wgt_ws, n_targets, out, ldo, stream);
}

extern "C" int morig_gather_rows(const float* src, int32_t lds, const int32_t* idx, int32_t rows, int32_t cols,
                                 float* dst, int32_t ldd, void* stream) {
    if (!src || !idx || !dst || rows < 0 || cols <= 0 || ldd < cols) return MORIG_E_INVALID;
    if (rows == 0) return MORIG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = ((int64_t)rows * cols + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfScope ps(K_COPY, s, 0.0, 8.0 * rows * cols);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((int)blocks), dim3(256), 0, s, src, lds, idx, rows, cols, dst, ldd);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}
