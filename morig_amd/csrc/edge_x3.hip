// Persistent 32-wide EdgeConv on a 3-channel vertex input (morig_edgeconv_x3, split-fp16 path): the position branches
// (models/basic_modules.py:193-195, nn_pos([pos_i, pos_j - pos_i]), two 16-wide units paired) and motionNet's first unit (nn_x on the
// 3-channel keyframe flow, models/rignet.py:86).
//
// The generic tile engine runs these layers one 128-row tile per workgroup; a phase ablation of it on the headline graphs
// (tools/x3_phases.py) gave: tile skeleton (edge count -> edge ids -> gathers -> staging -> barriers, W2 re-read per tile) 54 % of
// the launch, segmented-max epilogue 37 %, MFMAs + gathers 9 %. The work per tile is tiny (6 MFMAs per wave); what it pays for is
// a chain of dependent memory latencies per tile. Here a workgroup is persistent and walks tiles t, t + grid, ...:
//   * W2 fragments, bias / BN constants and the first-layer rows are loaded ONCE per workgroup;
//   * the edge ids of tile t + 2 and the endpoint inputs of tile t + 1 (2 x 16 bytes per row) are in flight while tile t computes:
//     no dependent load is ever waited for inside a tile;
//   * the first Linear runs on the matrix pipe as well, transposed (D = W1ext Xext^T, K = 16: [W1a | W1b | b1] against [x_i | x_j | 1]):
//     a lane then owns a ROW and 16 of its 32 hidden channels, which it ReLUs, splits to (hi, lo) fp16 and writes into the row's
//     144-byte LDS line; a wave stages exactly the 32 rows its own second-layer MFMAs read, so the operand hand-over needs no
//     workgroup barrier; the results go back into the SAME lines as fp32 (32 floats + pad = 144 bytes);
//   * ONE barrier per tile (in front of the segmented max, which crosses wave boundaries); LDS lines, destination ids and segment
//     lists are double-buffered by tile parity so that the next tile's staging never waits for this tile's scan.
// Segmented max, tile-straddling segments (integer-atomic float max onto rows pre-set by init_boundary_rows) and the range flag are
// those of the tile engine's narrow path (tile_gemm.hip).
#include "common.h"

namespace morig {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_BM = 128, X3_LINE = 144;          // rows per tile; bytes per LDS line: [32 hi | 32 lo | pad] = [32 fp32 | pad]

// AB = true [r06]: the same kernel for morig_edgeconv at H = 32 -- the first layer's per-vertex terms come as [A | B] rows (32 floats each,
// one vertex GEMM: the first unit of the head, the point networks' vertex branch), so a lane gathers the 16 hidden channels it stages
// from A[dst] and B[src] (4 + 4 16-byte loads, both half-waves), adds, ReLUs and splits them; everything behind the staging is shared.
// The tile engine ran these launches one 128-row tile per workgroup: 96 + 196 us per headline step against 58 + 127 us for the 3-channel
// form on the same graphs.
template <bool AB>
__global__ __launch_bounds__(256, 4) void edge_x3_kernel(const EdgeX3Params p) {
    __shared__ __attribute__((aligned(16))) char lines[2][X3_BM * X3_LINE];
    __shared__ int sseg[2][X3_BM];
    __shared__ int sstart[2][X3_BM + 2];
    __shared__ int snseg[2];
    __shared__ int sflag[2][2];                                         // first / last segment continues in a neighbour tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int row = wave * 32 + l31;                                    // this lane's tile row (both half-waves: the same row)
    const int Etot = p.rowptr[p.n_nodes];
    const int tpr = (Etot + X3_BM - 1) / X3_BM;
    const int T = tpr * p.replicas;
    if ((int)blockIdx.x >= T) return;

    // ---- resident operands. First layer as an MFMA too, TRANSPOSED (D = W1ext Xext^T: a lane owns a ROW and 16 of its 32 hidden
    // channels -- exactly what it must write into the row's line): W1ext[c] = [W1a[c] (3) | W1b[c] (3) | b1[c] | 0 ...] against
    // Xext[row] = [x_i (3) | x_j (3) | 1 | 0 ...], K = 16 with only k block 0 (lanes 0..31) non-zero, split-fp16 like every product.
    f16x8 w1h, w1l, wh[2], wl[2];
    {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!AB && hi == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { v[j] = p.W1a[l31 * 4 + j]; v[3 + j] = p.W1b[l31 * 4 + j]; }
            v[6] = p.b1[l31];
        }
        f32x4 hb, lb;
#pragma unroll
        for (int q = 0; q < 4; ++q) { float h2, l2; split_pair_f16(v[2 * q], v[2 * q + 1], h2, l2); hb[q] = h2; lb[q] = l2; }
        w1h = __builtin_bit_cast(f16x8, hb); w1l = __builtin_bit_cast(f16x8, lb);
        const char* wr = reinterpret_cast<const char*>(p.W2s + (size_t)l31 * p.ldw);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            wh[st] = *reinterpret_cast<const f16x8*>(wr + 32 * st + 16 * hi);
            wl[st] = *reinterpret_cast<const f16x8*>(wr + 64 + 32 * st + 16 * hi);
        }
    }
    const float cb = p.bias[l31], cs = p.scale[l31], ct = p.shift[l31];

    // ---- tile state in flight: edge ids two tiles ahead, endpoint inputs one tile ahead (named registers, rotated: no indexing) ----
    auto tile_row0 = [&](int t, int& rep) __attribute__((always_inline)) { rep = t / tpr; return (t - rep * tpr) * X3_BM; };
    auto load_ids = [&](int t, int& d, int& sidx, int& edge) __attribute__((always_inline)) {
        if (t >= T) return;
        int rep; const int r0 = tile_row0(t, rep);
        const int r = min(r0 + row, Etot - 1);
        d = p.dstS[r]; sidx = p.srcS[r];
        if (tid < 2) edge = p.dstS[min(max(tid == 0 ? r0 - 1 : r0 + X3_BM, 0), Etot - 1)];     // thread 0: id in front of the tile, 1: behind
    };
    auto load_x = [&](int t, int d, int sidx, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        if (AB || t >= T || hi != 0) return;
        int rep; const int r0 = tile_row0(t, rep);
        const bool live = r0 + row < Etot;
        const size_t base = (size_t)rep * p.rep_in;
        a = *reinterpret_cast<const f32x4*>(p.X + (base + (live ? d : 0)) * p.ldx);
        b = *reinterpret_cast<const f32x4*>(p.X + (base + (live ? sidx : 0)) * p.ldx);
    };
    // AB: the 16 channels this lane stages, (8 g4 + 4 hi .. + 3) for g4 = 0 .. 3, of A[dst] and of B[src]
    auto load_ab = [&](int t, int d, int sidx, f32x4 (&a)[4], f32x4 (&b)[4]) __attribute__((always_inline)) {
        if (!AB || t >= T) return;
        int rep; const int r0 = tile_row0(t, rep);
        const bool live = r0 + row < Etot;
        const size_t base = (size_t)rep * p.rep_in;
        const float* ar = p.A + (base + (live ? d : 0)) * p.lda + 4 * hi;
        const float* br = p.B + (base + (live ? sidx : 0)) * p.ldb + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) { a[g4] = *reinterpret_cast<const f32x4*>(ar + 8 * g4); b[g4] = *reinterpret_cast<const f32x4*>(br + 8 * g4); }
    };
    const int t0 = blockIdx.x, tstep = gridDim.x;
    int d_cur = 0, s_cur = 0, e_cur = 0, d_nxt = 0, s_nxt = 0, e_nxt = 0;
    f32x4 xi_cur = {0.f, 0.f, 0.f, 0.f}, xj_cur = xi_cur, xi_nxt = xi_cur, xj_nxt = xi_cur;
    f32x4 ga_cur[4], gb_cur[4], ga_nxt[4], gb_nxt[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) ga_cur[g4] = gb_cur[g4] = ga_nxt[g4] = gb_nxt[g4] = xi_cur;
    load_ids(t0, d_cur, s_cur, e_cur);
    load_ids(t0 + tstep, d_nxt, s_nxt, e_nxt);
    load_x(t0, d_cur, s_cur, xi_cur, xj_cur);
    load_ab(t0, d_cur, s_cur, ga_cur, gb_cur);
    bool amax_bad = false;

    int par = 0;
    for (int t = t0; t < T; t += tstep, par ^= 1) {
        int rep; const int r0 = tile_row0(t, rep);
        const bool live = r0 + row < Etot;
        // the next tile's inputs (its ids arrived a tile ago) and the ids of the tile after it
        load_x(t + tstep, d_nxt, s_nxt, xi_nxt, xj_nxt);
        load_ab(t + tstep, d_nxt, s_nxt, ga_nxt, gb_nxt);
        int d_nn = 0, s_nn = 0, e_nn = 0;
        load_ids(t + 2 * tstep, d_nn, s_nn, e_nn);

        char* L = lines[par];
        if (hi == 0) sseg[par][row] = live ? d_cur : -1;
        if (tid == 0) sflag[par][0] = (r0 > 0 && d_cur == e_cur) ? 1 : 0;                          // thread 0 holds row 0 and the id in front
        // ---- first layer: Xext fragment (B operand) of this lane's row, 3 MFMAs (AB: the gathered terms added), ReLU, split, staged
        // into the row's line ----
        {
            f32x16 a1;
            if constexpr (AB) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int q = 0; q < 4; ++q) a1[4 * g4 + q] = ga_cur[g4][q] + gb_cur[g4][q];
            } else {
                f32x4 hb = {0.f, 0.f, 0.f, 0.f}, lb = hb;
                if (hi == 0) {
                    float h2, l2;
                    split_pair_f16(xi_cur[0], xi_cur[1], h2, l2); hb[0] = h2; lb[0] = l2;
                    split_pair_f16(xi_cur[2], xj_cur[0], h2, l2); hb[1] = h2; lb[1] = l2;
                    split_pair_f16(xj_cur[1], xj_cur[2], h2, l2); hb[2] = h2; lb[2] = l2;
                    split_pair_f16(1.0f, 0.0f, h2, l2);           hb[3] = h2; lb[3] = l2;
                    const float am = fmaxf(fmaxf(fmaxf(fabsf(xi_cur[0]), fabsf(xi_cur[1])), fmaxf(fabsf(xi_cur[2]), fabsf(xj_cur[0]))),
                                           fmaxf(fabsf(xj_cur[1]), fabsf(xj_cur[2])));
                    if (!(am < 65000.f)) amax_bad = true;
                }
                const f16x8 xh = __builtin_bit_cast(f16x8, hb), xl = __builtin_bit_cast(f16x8, lb);
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[r] = 0.f;
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1l, xh, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, xl, a1, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1h, xh, a1, 0, 0, 0);
            }
            // register r of a1 = hidden channel (r & 3) + 8 (r >> 2) + 4 hi of row `row`
            char* line = L + row * X3_LINE;
            float am = 0.f, chk = 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (AB) chk = fmaf(a1[4 * g4 + q], 0.f, chk);                // NaN / Inf of a gathered term: fmaxf would drop it (-> 0)
                    h[q] = fmaxf(a1[4 * g4 + q], 0.f); am = fmaxf(am, h[q]);
                }
                float h0, l0, h1, l1;
                split_pair_f16(h[0], h[1], h0, l0);
                split_pair_f16(h[2], h[3], h1, l1);
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 hv = {h0, h1}, lv = {l0, l1};
                const int c0 = 8 * g4 + 4 * hi;                                  // first of my 4 adjacent channels
                *reinterpret_cast<f32x2*>(line + 2 * c0) = hv;
                *reinterpret_cast<f32x2*>(line + 64 + 2 * c0) = lv;
            }
            if (!(chk == 0.f) || !(am < 65000.f)) amax_bad = true;
        }
        asm volatile("" ::: "memory");                     // lines written as float pairs, read back as halves (type punning: DESIGN 5 (13))
        // ---- second layer on this wave's own 32 rows: same-wave LDS traffic is in order, no barrier ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const char* ar = L + (wave * 32 + l31) * X3_LINE + 16 * hi;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(ar + 32 * st);
                const f16x8 al = *reinterpret_cast<const f16x8*>(ar + 64 + 32 * st);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[st], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[st], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[st], acc, 0, 0, 0);
            }
        }
        asm volatile("" ::: "memory");
        // ---- ReLU / BN affine, results back into the same lines as fp32 (lane = column) ----
        {
            float* Z = reinterpret_cast<float*>(L);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                Z[rl * (X3_LINE / 4) + l31] = fmaxf(acc[r] + cb, 0.f) * cs + ct;
            }
        }
        __syncthreads();                                   // every wave's result rows and the destination ids are visible
        // ---- segment starts of the tile (wave 0), then one slot of 8 threads x 4 columns per segment ----
        if (wave == 0) {
            const int rA = lane, rB = lane + 64;
            const bool f0 = rA == 0 || sseg[par][rA] != sseg[par][rA - 1];
            const bool f1 = sseg[par][rB] != sseg[par][rB - 1];
            const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (f0) sstart[par][__popcll(m0 & below)] = rA;
            if (f1) sstart[par][__popcll(m0) + __popcll(m1 & below)] = rB;
            if (lane == 0) { const int ns = __popcll(m0) + __popcll(m1); snseg[par] = ns; sstart[par][ns] = X3_BM; }
            if (lane == 1) sflag[par][1] = (r0 + X3_BM < Etot && sseg[par][X3_BM - 1] == e_cur) ? 1 : 0;    // thread 1 holds the id behind
        }
        __syncthreads();
        {
            const float* Z = reinterpret_cast<const float*>(L);
            const int c4 = (tid & 7) * 4, slot = tid >> 3;
            const int nseg = snseg[par];
            const bool first_cont = sflag[par][0] != 0, last_cont = sflag[par][1] != 0;
            float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + c4;
            for (int k = slot; k < nseg; k += 32) {
                const int rs = sstart[par][k], re = sstart[par][k + 1];
                const int sg = sseg[par][rs];
                if (sg < 0) continue;                      // rows past the last edge
                f32x4 m = *reinterpret_cast<const f32x4*>(Z + rs * (X3_LINE / 4) + c4);
                for (int r = rs + 1; r < re; r += 4) {
                    f32x4 z[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) z[u] = *reinterpret_cast<const f32x4*>(Z + min(r + u, re - 1) * (X3_LINE / 4) + c4);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], z[u][q]);
                }
                float* o = obase + (size_t)sg * p.ldy;
                const bool partial = (rs == 0 && first_cont) || (re == X3_BM && last_cont);
                if (partial) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) atomic_max_f32(o + q, m[q]);
                } else {
                    *reinterpret_cast<f32x4*>(o) = m;
                }
            }
        }
        // (no barrier here: the next tile works on the other parity's lines / lists; the tile after it passes the next tile's
        // barriers first, which every wave reaches only after this scan)
        d_cur = d_nxt; s_cur = s_nxt; e_cur = e_nxt; xi_cur = xi_nxt; xj_cur = xj_nxt;
        if constexpr (AB) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { ga_cur[g4] = ga_nxt[g4]; gb_cur[g4] = gb_nxt[g4]; }
        }
        d_nxt = d_nn; s_nxt = s_nn; e_nxt = e_nn;
    }
    if (amax_bad) *p.ovf = 1;
}

int launch_edge_x3(const EdgeX3Params& p, int n_tiles_cap, hipStream_t s) {
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    int avail = ncu - reserved_cus();
    if (avail < 8) avail = 8;
    int grid = avail * 4;                                  // 4 workgroups per CU (LDS: 2 x 18 KB of lines each)
    if (grid > n_tiles_cap) grid = n_tiles_cap;
    if (grid < 1) grid = 1;
    if (p.X) hipLaunchKernelGGL(edge_x3_kernel<false>, dim3(grid), dim3(256), 0, s, p);
    else     hipLaunchKernelGGL(edge_x3_kernel<true>, dim3(grid), dim3(256), 0, s, p);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig
