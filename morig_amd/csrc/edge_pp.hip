// Fused EdgeConv, PERSISTENT wave-specialised kernel for the wide layers (H = 128, 256) on the split-fp16 path.
//
// Same arithmetic and tile shape as edge_pc.hip (128 edge rows x H columns per tile, waves 0-3 consume with MFMA,
// waves 4-7 produce the operand tile: gather A[dst] + B[src], add, ReLU, split into fp16 hi/lo), but ONE workgroup
// per CU walks a list of tiles, so that
//   * the per-tile start-up chain (index loads -> row pointers -> first gathers, ~5 us with nothing to overlap it
//     when a 110 KB-LDS workgroup owns the CU) runs during the previous tile's MFMAs: the producers always hold the
//     next two K-chunks in registers, across tile boundaries, and load the next tile's indices half a tile ahead;
//   * the segmented-max scan of tile t has its own LDS region (Z), so chunk 0 of tile t+1 is already staged when
//     the scan ends and the consumers restart at once.
// Measured on MI355X (tools/ubench/overlap.hip, gaps.hip; MI355X_MICROARCH.md "LDS-DMA piece issue cost"): every VMEM
// wave-instruction issued on a SIMD whose matrix pipe is busy costs ~60-180 cycles, whichever wave issues it: a sibling
// wave's loads starve behind back-to-back MFMAs, and loads/LDS-DMA issued from inside the MFMA stream stall it instead
// (consumer-issued W2 LDS-DMA measured 13 % SLOWER than this version). What counts is the NUMBER of VMEM
// wave-instructions per MFMA: the producers use 16-byte loads only, fetch W2 one chunk ahead and the gathered rows two.
//
// Barrier protocol (all 8 waves, same count in both roles), per tile:  B_0 .. B_7, E1
//   consumers:  for c: { B_c; consume chunk c }   write Z;  E1;  scan
//   producers:  B_0;  for c = 1..7: { stage chunk c; B_c }   stage chunk 0 of the NEXT tile;  E1;  scan
// Chunk c lives in ring stage c & 1; it is overwritten only after the barrier that follows its consumption.
#include "common.h"
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace morig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));

template <int C> using IC = std::integral_constant<int, C>;

// Workgroup barrier that only drains this wave's LDS traffic (lgkmcnt): global loads in flight and the scan's global
// stores must not be waited for here (__syncthreads() adds vmcnt(0) once stores are pending).
__device__ __forceinline__ void pp_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Build with -DMORIG_PP_TRACE to record s_memtime stamps of workgroup 8, tile 3 (wave 0 = a consumer, wave 4 = a
// producer) into p.trace[role][32]; launch_edge_pp prints the deltas. Diagnostic only.
#ifdef MORIG_PP_TRACE
#define PP_TS(k) do { if (j == 3 && blockIdx.x == 8 && lane == 0 && (wave & 3) == 0) p.trace[(wave >> 2) * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_TS(k) do { } while (0)
#endif

template <int H, bool QUAD>
__global__ __launch_bounds__(512, 2) void edge_pp_kernel(const EdgePcParams p) {
    constexpr int BM = 128, KC = 32, LDB = 144;         // LDB: bytes per LDS row = [32 hi | 32 lo | 16 pad]
    constexpr int NT = H / 64;                          // consumer wave tile: 64 rows x H/2 cols
    constexpr int MT = 2;
    constexpr int NCHUNK = H / KC;
    static_assert(NCHUNK % 2 == 0 && NCHUNK >= 4, "ring/register-set parity");
    constexpr bool WRES = (H == 128);                   // H = 128: all of W2 (4 chunks, 72 KB padded) stays resident in LDS
    constexpr int STAGE = (BM + (WRES ? 0 : H)) * LDB;  // bytes per ring stage: operand rows, then (streamed) W2 rows
    constexpr int WRESB = WRES ? NCHUNK * H * LDB : 0;
    constexpr int ZQ = H + 4;                           // quad mode: 32 quad-rows x H columns (+4: 16-byte rows, 2-way store conflicts only)
    constexpr int ZC = 64, ZLD = ZC + 1;                // general mode: 128 rows x 64 columns per pass
    constexpr int ZB = (32 * ZQ > BM * ZLD ? 32 * ZQ : BM * ZLD) * 4;
    constexpr int ATAB = QUAD ? 4 * 1024 : 0;           // quad mode: per producer wave, 8 quads x 128 B of A rows for one chunk
    __shared__ __attribute__((aligned(16))) char smem[WRESB + 2 * STAGE + ZB + 2 * BM * 4 + 32 + 3 * H * 4 + ATAB];
    char* wres = smem;                                  // [chunk][row][LDB]
    char* aring = smem + WRESB;
    float* Z = reinterpret_cast<float*>(aring + 2 * STAGE);
    int* sseg_all = reinterpret_cast<int*>(aring + 2 * STAGE + ZB);         // [2][BM] destination id per tile row
    int* sflag = sseg_all + 2 * BM;                                         // [2][2] first/last segment continues
    float* sbias = reinterpret_cast<float*>(sflag + 8);                     // [3][H] bias, BN scale, BN shift
    char* atab = reinterpret_cast<char*>(sbias + 3 * H);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = (wave & 3) >> 1, wn = wave & 1;

    // ---- this workgroup's tile list: XCD x (= blockIdx & 7 under round-robin dispatch) owns a contiguous range ----
    const int Etot = p.rowptr[p.n_nodes];
    const int tpr = (Etot + BM - 1) / BM;
    const int T = tpr * p.replicas;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int t_lo = (int)((long long)T * xcd / 8), t_hi = (int)((long long)T * (xcd + 1) / 8);
    const int n_my = (t_hi - t_lo - bi + nbx - 1) / nbx;                     // tiles t_lo + bi + j * nbx < t_hi
    if (n_my <= 0) return;                                                   // block-uniform
    auto tile_of = [&](int j) __attribute__((always_inline)) { return t_lo + bi + (j < n_my ? j : n_my - 1) * nbx; };
    if (tid < H) { sbias[tid] = p.bias[tid]; sbias[H + tid] = p.scale[tid]; sbias[2 * H + tid] = p.shift[tid]; }   // visible after B_0

    // ---- segmented max of one finished tile (Z already holds pass 0); all 8 waves scan ----
    f32x16 acc[MT][NT];
    // y = relu(acc + b) * sc + sh is monotone in acc (rising for sc >= 0, falling for sc < 0), so the max over a quad's
    // four rows is f(max acc) or f(min acc): two 3-input min/max per quad instead of the affine on every element
    auto write_z_quad = [&]() __attribute__((always_inline)) {
        // The min/max below read the accumulators from INLINE ASSEMBLY, where the compiler does not insert the wait states the
        // matrix pipe's write -> VALU read needs (lesson 11: pointconv_fused.hip read stale accumulators that
        // way). This statement takes every accumulator as an in/out operand, so it is ordered behind the MFMAs and in front of
        // the reads, and its own s_nop covers the 11 wait states of an 8-pass MFMA.
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("s_nop 15" : "+v"(acc[mt][nt]));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = wn * NT * 32 + nt * 32 + l31;
            const float b = sbias[col], sc = sbias[H + col], sh = sbias[2 * H + col];
            const bool rising = sc >= 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a0 = acc[mt][nt][4 * q], a1 = acc[mt][nt][4 * q + 1], a2 = acc[mt][nt][4 * q + 2], a3 = acc[mt][nt][4 * q + 3];
                    float hi4, lo4, t3;                    // raw v_max3/v_min3: fmaxf() would first canonicalise every input
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(a0), "v"(a1), "v"(a2));
                    asm("v_max_f32 %0, %1, %2" : "=v"(hi4) : "v"(t3), "v"(a3));
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(a0), "v"(a1), "v"(a2));
                    asm("v_min_f32 %0, %1, %2" : "=v"(lo4) : "v"(t3), "v"(a3));
                    const float x = rising ? hi4 : lo4;
                    Z[(wm * 16 + mt * 8 + 2 * q + hi) * ZQ + col] = fmaxf(x + b, 0.f) * sc + sh;
                }
        }
    };
    auto write_z_pass = [&](auto cb_const) __attribute__((always_inline)) {
        constexpr int cb = decltype(cb_const)::value;
        const int col = wn * NT * 32 + cb * 32 + l31;
        const float b = sbias[col], sc = sbias[H + col], sh = sbias[2 * H + col];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                Z[rl * ZLD + wn * 32 + l31] = fmaxf(acc[mt][cb][r] + b, 0.f) * sc + sh;
            }
    };
    // quad mode scan: a scanning wave owns NOWN quad-rows and every segment that STARTS there (following it into later
    // rows); a lane holds VEC = H/64 adjacent columns, so a segment's result leaves as one 16-byte (8-byte) store per lane.
    // Segment bookkeeping is two ballots, not a per-row state machine (which compiled to ~60 scalar-and-branch instructions
    // per quad row, ~2 000-2 800 cycles per tile): lane q < 32 looks at quad row q, START has a bit where a row opens a
    // segment, VALID where it holds a real vertex; the wave then walks the set bits of (START & VALID & its rows) --
    // ~1.4 segments -- and reduces each segment's rows with up to four LDS reads in flight.
    auto scan_quad = [&](int rep, const int* sseg, bool first_cont, bool last_cont, auto nownc, int wslot) __attribute__((always_inline)) {
        constexpr int VEC = H / 64;
        constexpr int NOWN = decltype(nownc)::value;   // quad rows owned per scanning wave
        typedef float fvec __attribute__((ext_vector_type(VEC)));
        const int q0 = __builtin_amdgcn_readfirstlane(wslot * NOWN);
        const float* zl = Z + VEC * lane;
        float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + VEC * lane;
        const int ql = lane & 31;
        const int sq = sseg[4 * ql];                                        // id of quad row ql (its 4 rows share it)
        const int sp = sseg[ql > 0 ? 4 * ql - 1 : 0];                       // id of the row just above
        const unsigned START = (unsigned)__ballot(lane < 32 && (ql == 0 || sq != sp));
        const unsigned VALID = (unsigned)__ballot(lane < 32 && sq >= 0);
        unsigned mine = START & VALID & (((1u << NOWN) - 1u) << q0);
        while (mine) {                                                       // wave-uniform: SALU bit walking
            const int b = __builtin_ctz(mine);
            mine &= mine - 1u;
            const unsigned later = b < 31 ? (START & ~((2u << b) - 1u)) : 0u;
            const int e = later ? __builtin_ctz(later) : 32;                 // the segment covers quad rows [b, e)
            const int sg = __builtin_amdgcn_readlane(sq, b);
            fvec m = *reinterpret_cast<const fvec*>(zl + b * ZQ);
            for (int q = b + 1; q < e; q += 4) {
                const int l = e - 1;
                const fvec z0 = *reinterpret_cast<const fvec*>(zl + q * ZQ);
                const fvec z1 = *reinterpret_cast<const fvec*>(zl + min(q + 1, l) * ZQ);
                const fvec z2 = *reinterpret_cast<const fvec*>(zl + min(q + 2, l) * ZQ);
                const fvec z3 = *reinterpret_cast<const fvec*>(zl + min(q + 3, l) * ZQ);
#pragma unroll
                for (int v = 0; v < VEC; ++v) m[v] = fmaxf(fmaxf(m[v], z0[v]), fmaxf(fmaxf(z1[v], z2[v]), z3[v]));
            }
            float* o = obase + (size_t)sg * p.ldy;
            const bool partial = (b == 0 && first_cont) || (e == 32 && last_cont);
            if (partial) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) atomic_max_f32(o + v, m[v]);
            } else {
                *reinterpret_cast<fvec*>(o) = m;
            }
        }
    };
    // general mode scan of one 64-column pass: wave g owns rows [16 g, 16 g + 16) and every segment that STARTS there; same
    // ballot bookkeeping as the quad scan, over 128 rows (lane l looks at rows l and l + 64)
    struct SegMasks { unsigned long long start_lo, start_hi, own; int s_own; };     // the same for every pass of a tile
    auto seg_masks = [&](const int* sseg) __attribute__((always_inline)) {
        const int r0 = __builtin_amdgcn_readfirstlane((tid >> 6) * 16);
        const int s_lo = sseg[lane], s_hi = sseg[lane + 64];
        const int p_lo = sseg[lane > 0 ? lane - 1 : 0], p_hi = sseg[lane + 63];
        SegMasks k;
        k.start_lo = __ballot(lane == 0 || s_lo != p_lo); k.start_hi = __ballot(s_hi != p_hi);
        const bool upper = r0 >= 64;
        const unsigned long long valid = upper ? __ballot(s_hi >= 0) : __ballot(s_lo >= 0);
        k.own = (upper ? k.start_hi : k.start_lo) & valid & (0xFFFFull << (r0 & 63));
        k.s_own = upper ? s_hi : s_lo;
        return k;
    };
    auto scan_pass = [&](int cb, int rep, const SegMasks& k, bool first_cont, bool last_cont) __attribute__((always_inline)) {
        const int zc = tid & 63;
        const bool upper = (tid >> 6) >= 4;
        const int col = (zc >> 5) * NT * 32 + cb * 32 + (zc & 31);
        const float* zcolp = Z + zc;
        float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + col;
        const unsigned long long START_lo = k.start_lo, START_hi = k.start_hi;
        unsigned long long mine = k.own;
        while (mine) {                                                       // wave-uniform
            const int bl = __builtin_ctzll(mine);
            mine &= mine - 1ull;
            const unsigned long long above = bl < 63 ? ~((2ull << bl) - 1ull) : 0ull;
            int b, e, sg;
            if (upper) {
                const unsigned long long later = START_hi & above;
                b = 64 + bl; e = later ? 64 + __builtin_ctzll(later) : BM;
            } else {
                const unsigned long long later = START_lo & above;
                b = bl; e = later ? __builtin_ctzll(later) : (START_hi ? 64 + __builtin_ctzll(START_hi) : BM);
            }
            sg = __builtin_amdgcn_readlane(k.s_own, bl);
            float m = zcolp[b * ZLD];
            for (int r = b + 1; r < e; r += 4) {
                const int l = e - 1;
                const float z0 = zcolp[r * ZLD], z1 = zcolp[min(r + 1, l) * ZLD], z2 = zcolp[min(r + 2, l) * ZLD], z3 = zcolp[min(r + 3, l) * ZLD];
                m = fmaxf(fmaxf(m, z0), fmaxf(fmaxf(z1, z2), z3));
            }
            float* o = obase + (size_t)sg * p.ldy;
            if ((b == 0 && first_cont) || (e == BM && last_cont)) atomic_max_f32(o, m); else *o = m;
        }
    };
    // everything after E1 for tile j (both roles; the general mode re-stages Z per 64-column pass)
    auto finish_tile = [&](int j, auto role) __attribute__((always_inline)) {
        constexpr bool is_producer = decltype(role)::value != 0;
        const int t = tile_of(j);
        const int rep = t / tpr;
        const int* sseg = sseg_all + (j & 1) * BM;
        const bool fc = sflag[(j & 1) * 2] != 0, lc = sflag[(j & 1) * 2 + 1] != 0;
        if (p.dbg & 1) return;
        if constexpr (QUAD && H == 128) {                   // the consumer waves scan 6 quad-rows each; the producers, which are the
            if constexpr (!is_producer) scan_quad(rep, sseg, fc, lc, IC<6>{}, wave);     // critical path, stage the next tile's
            else scan_quad(rep, sseg, fc, lc, IC<2>{}, 12 + (wave - 4));                 // chunk 1 first and then take 2 each
            return;
        }
        if constexpr (QUAD) {                               // H = 256: the consumers scan alone (8 each), the producers only stage
            if constexpr (!is_producer) scan_quad(rep, sseg, fc, lc, IC<8>{}, wave);
            return;
        }
        const SegMasks k = seg_masks(sseg);
        scan_pass(0, rep, k, fc, lc);
        auto more = [&](auto cbc) __attribute__((always_inline)) {
            pp_barrier();                            // previous pass scanned
            if constexpr (!is_producer) write_z_pass(cbc);
            pp_barrier();
            scan_pass(decltype(cbc)::value, rep, k, fc, lc);
        };
        if constexpr (!QUAD && NT > 1) more(IC<1>{});
        if constexpr (!QUAD && NT > 2) more(IC<2>{});
        if constexpr (!QUAD && NT > 3) more(IC<3>{});
    };

    if (producer) {
        // ---------------- producers: 256 threads, thread (row = pt/8 + 32 i, 16-byte piece = pt%8) ----------------
        const int pt = tid - 256;
        const int lrow = pt >> 3, lkq = pt & 7;
        unsigned oa[4], ob[4];                               // byte offsets of this thread's 4 gathered rows (< 4 GB per replica)
        // quad mode (4-aligned segments): the four rows of a quad share their destination, so A[dst] is fetched once
        // per QUAD: this lane loads piece (lane & 7) of quad qe of its own wave's 8 quads into a 1 KB per-wave LDS
        // table, from which the wave's threads take the A piece of each of their 4 rows (1 gather + 4 ds_reads
        // instead of 4 gathers per chunk)
        const int pw4 = pt >> 6;
        const int qe = 8 * ((lane >> 4) & 3) + 2 * pw4 + ((lane >> 3) & 1);
        char* atab_w = atab + pw4 * 1024;
        const char* atab_r = atab_w + ((pt >> 5) & 1) * 128 + lkq * 16;      // + 256 i
        unsigned oq = 0; int nq = 0;
        f32x4 ta[2];
        const char* abase = reinterpret_cast<const char*>(p.A);             // + replica offset: wave-uniform, lives in SGPRs
        const char* bbase = reinterpret_cast<const char*>(p.B);
        int nd[4], ns[4], nprev, nlast, nafter;             // next tile's indices, in flight
        int nrow0 = 0, nrep = 0;
        auto load_indices = [&](int j) __attribute__((always_inline)) {
            const int t = tile_of(j);
            nrep = t / tpr; nrow0 = (t - nrep * tpr) * BM;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = min(nrow0 + lrow + 32 * i, Etot - 1);
                if constexpr (!QUAD) nd[i] = p.dstS[row];
                ns[i] = p.srcS[row];
            }
            if constexpr (QUAD) nq = p.dstS[min(nrow0 + 4 * qe, Etot - 1)];
            nprev = p.dstS[max(nrow0 - 1, 0)];
            nlast = p.dstS[min(nrow0 + BM - 1, Etot - 1)];
            nafter = p.dstS[min(nrow0 + BM, Etot - 1)];
        };
        auto switch_tile = [&](int j) __attribute__((always_inline)) {   // make tile j (indices loaded) the one fetched from
            abase = reinterpret_cast<const char*>(p.A + (size_t)nrep * p.rep_in * p.lda);
            bbase = reinterpret_cast<const char*>(p.B + (size_t)nrep * p.rep_in * p.ldb);
            int* sseg = sseg_all + (j & 1) * BM;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool valid = nrow0 + lrow + 32 * i < Etot;
                if constexpr (!QUAD) oa[i] = ((unsigned)nd[i] * (unsigned)p.lda + 4u * lkq) * 4u;   // rows past the end re-read the
                ob[i] = ((unsigned)ns[i] * (unsigned)p.ldb + 4u * lkq) * 4u;          // last edge: finite, ignored by the scan (id -1)
                if (p.dbg & 4) oa[i] = ob[i] = 16u * lkq;                              // ablation: every gather hits row 0
                if constexpr (!QUAD) { if (lkq == 0) sseg[lrow + 32 * i] = valid ? nd[i] : -1; }
            }
            if constexpr (QUAD) {
                oq = ((unsigned)nq * (unsigned)p.lda + 4u * lkq) * 4u;
                if (lkq == 0) {                             // Etot is a multiple of 4 here: a quad is valid or not as a whole
                    const int d = (nrow0 + 4 * qe < Etot) ? nq : -1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sseg[4 * qe + k] = d;
                }
            }
            if (pt == 0) {
                const int first = QUAD ? nq : nd[0];                               // pt == 0 holds row nrow0 / quad 0
                sflag[(j & 1) * 2] = (nrow0 > 0 && nprev == first) ? 1 : 0;
                sflag[(j & 1) * 2 + 1] = (nrow0 + BM < Etot && nlast == nafter) ? 1 : 0;
            }
        };
        const unsigned ow = ((unsigned)lrow * (unsigned)p.ldw + 4u * lkq) * 4u;
        const char* wbase = reinterpret_cast<const char*>(p.W);
        constexpr int PW = H / 32;                          // W2 passes of 32 rows
        f32x4 ra[2][4], rb[2][4], rw[PW];                   // gathers: two chunks in flight; W2 (L2-resident): one
        auto fetch_w = [&](int c) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                unsigned o = ow;
                asm volatile("" : "+v"(o));                 // keep the (scalar base + 32-bit lane offset) form: no hoisted 64-bit VGPR pairs
                rw[i] = *reinterpret_cast<const f32x4*>(wbase + (size_t)i * 32 * p.ldw * 4 + c * KC * 4 + o);
            }
        };
        auto stage_w = [&](int c) __attribute__((always_inline)) {
            char* sB = WRES ? wres + c * (H * LDB) : aring + (c & 1) * STAGE + BM * LDB;
#pragma unroll
            for (int i = 0; i < PW; ++i) *reinterpret_cast<f32x4*>(sB + (lrow + 32 * i) * LDB + 16 * lkq) = rw[i];
        };
        float amax = 0.f;
        auto fetch_g = [&](int c, auto setc) __attribute__((always_inline)) {
            constexpr int S = decltype(setc)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (!QUAD) ra[S][i] = *reinterpret_cast<const f32x4*>(abase + c * KC * 4 + oa[i]);
                rb[S][i] = *reinterpret_cast<const f32x4*>(bbase + c * KC * 4 + ob[i]);
            }
            if constexpr (QUAD) ta[S] = *reinterpret_cast<const f32x4*>(abase + c * KC * 4 + oq);
        };
        auto stage_a = [&](int c, auto setc) __attribute__((always_inline)) {
            constexpr int S = decltype(setc)::value;
            char* sA = aring + (c & 1) * STAGE;
            f32x4 qa[4];
            if constexpr (QUAD) {                           // same-wave LDS hand-over: the LDS executes a wave's operations in order
                *reinterpret_cast<f32x4*>(atab_w + lane * 16) = ta[S];
#pragma unroll
                for (int i = 0; i < 4; ++i) qa[i] = *reinterpret_cast<const f32x4*>(atab_r + 256 * i);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf((QUAD ? qa[i][q] : ra[S][i][q]) + rb[S][i][q], 0.f);
                typedef float b32x2 __attribute__((ext_vector_type(2)));
                b32x2 h, l;                                   // [hi(v0) hi(v1) | hi(v2) hi(v3)], the same for lo (common.h)
                float h0, h1, l0, l1;
                split_pair_f16(v[0], v[1], h0, l0);
                split_pair_f16(v[2], v[3], h1, l1);
                h[0] = h0; h[1] = h1; l[0] = l0; l[1] = l1;
                amax = fmaxf(amax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));               // v >= 0 after the ReLU
                char* rowp = sA + (lrow + 32 * i) * LDB + 8 * lkq;
                *reinterpret_cast<b32x2*>(rowp) = h;
                *reinterpret_cast<b32x2*>(rowp + 64) = l;
            }
        };
        // one producer interval: stage chunk c from register set c & 1, then refill that set with chunk c + 2
        auto interval = [&](auto cc, int j, auto refill) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;          // chunk staged (0 = next tile's chunk 0)
            constexpr bool fill = decltype(refill)::value != 0;
            constexpr int cn = (c + 2) % NCHUNK;            // chunk fetched
            using S = IC<(c & 1)>;
            if constexpr (!WRES) {
                stage_w(c);
                fetch_w((c + 1) % NCHUNK);                  // W2 loads first: they issue while the consumers wait for their ds_reads
            }
            stage_a(c, S{});
            if constexpr (c == NCHUNK - 2) switch_tile(j + 1);     // chunks 0, 1, ... fetched from here on are the next tile's
            if constexpr (fill) fetch_g(cn, S{});
            // indices for the next switch_tile: loaded well ahead, but (H = 256) not held across the scan
            if constexpr (NCHUNK == 8 && c == 1) load_indices(j + 1);
            if constexpr (NCHUNK == 4 && c == NCHUNK - 1) load_indices(j + 2);
        };

        load_indices(0);
        switch_tile(0);
        if constexpr (WRES) {
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) { fetch_w(c); stage_w(c); }                   // visible after B_0
        } else {
            fetch_w(0);
        }
        fetch_g(0, IC<0>{});
        fetch_g(1, IC<1>{});
        if constexpr (NCHUNK == 4) load_indices(1);
        interval(IC<0>{}, -1000, IC<1>{});                  // stages chunk 0 of tile 0, fetches chunk 2 (no switch: c != NCHUNK-2)
#pragma unroll 1
        for (int j = 0; j < n_my; ++j) {
            PP_TS(0); pp_barrier();                      // B_0
            // quad mode: chunk 1 of every tile but the first was staged while the consumers scanned the previous tile
            if (!QUAD || j == 0) { PP_TS(1); interval(IC<1>{}, j, IC<1>{}); PP_TS(2); }
            pp_barrier();
            PP_TS(3); interval(IC<2>{}, j, IC<1>{}); PP_TS(4); pp_barrier();
            PP_TS(5); interval(IC<3>{}, j, IC<1>{}); PP_TS(6); pp_barrier();
            if constexpr (NCHUNK == 8) {
                PP_TS(7); interval(IC<4>{}, j, IC<1>{}); PP_TS(8); pp_barrier();
                PP_TS(9); interval(IC<5>{}, j, IC<1>{}); PP_TS(10); pp_barrier();
                PP_TS(11); interval(IC<6>{}, j, IC<1>{}); PP_TS(12); pp_barrier();
                PP_TS(13); interval(IC<7>{}, j, IC<1>{}); PP_TS(14); pp_barrier();
            }
            if constexpr (QUAD) {
                PP_TS(15); interval(IC<0>{}, j, IC<1>{});   // next tile's chunk 0 (the last tile re-stages itself, unused)
                PP_TS(16); pp_barrier();                 // E1: chunk 7 is consumed, ring stage 1 is free
                PP_TS(17); interval(IC<1>{}, j + 1, IC<1>{});          // next tile's chunk 1, under the consumers' scan
                PP_TS(18); if constexpr (H == 128) finish_tile(j, IC<1>{});
            } else {
                PP_TS(15); interval(IC<0>{}, j, IC<0>{});
                PP_TS(16); pp_barrier();                 // E1
                PP_TS(17); finish_tile(j, IC<1>{});
                PP_TS(18); fetch_g(2, IC<0>{});             // register set 0 refilled AFTER the scan (keeps it out of the scan's way)
            }
        }
        if (!(amax < 65000.f)) *p.ovf = 1;
    } else {
        // ---------------- consumers: 4 waves as 2 (rows) x 2 (cols), fragments + MFMA only ----------------
        // one K-chunk: fragments of both 16-k steps, 3 MFMAs per (mt, nt) and step. `first`: the accumulators start from
        // the inline constant 0 (no 128 v_mov per tile)
        auto consume = [&](int c, auto firstc) __attribute__((always_inline)) {
            constexpr bool first = decltype(firstc)::value != 0;
            const char* sA = aring + (c & 1) * STAGE;
            const char* sB = WRES ? wres + c * (H * LDB) : sA + BM * LDB;
            const char* a0 = sA + (wm * 64 + l31) * LDB + 16 * hi;
            const char* b0 = sB + (wn * NT * 32 + l31) * LDB + 16 * hi;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    ah[mt] = *reinterpret_cast<const f16x8*>(a0 + mt * 32 * LDB + 32 * st);
                    al[mt] = *reinterpret_cast<const f16x8*>(a0 + mt * 32 * LDB + 32 * st + 64);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt] = *reinterpret_cast<const f16x8*>(b0 + nt * 32 * LDB + 32 * st);
                    bl[nt] = *reinterpret_cast<const f16x8*>(b0 + nt * 32 * LDB + 32 * st + 64);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (first && st == 0) {
                            f32x16 z;
#pragma unroll
                            for (int r = 0; r < 16; ++r) z[r] = 0.f;
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], z, 0, 0, 0);
                        } else {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        }
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    }
            }
        };
#pragma unroll 1
        for (int j = 0; j < n_my; ++j) {
            PP_TS(0); pp_barrier();                         // B_0: chunk 0 is in stage 0
            PP_TS(1);
            if (!(p.dbg & 2)) consume(0, IC<1>{});
#pragma unroll 1
            for (int c = 1; c < NCHUNK; ++c) {
                PP_TS(2 * c); pp_barrier();                 // B_c: chunk c is in stage c&1
                PP_TS(2 * c + 1);
                if (!(p.dbg & 2)) consume(c, IC<0>{});
            }
            PP_TS(16);
            if (!(p.dbg & 1)) { if constexpr (QUAD) write_z_quad(); else write_z_pass(IC<0>{}); }
            PP_TS(17); pp_barrier();                     // E1: Z visible; next tile's chunk 0 staged
            PP_TS(18); finish_tile(j, IC<0>{});
            PP_TS(19);
        }
    }
}

// CUs the caller wants left alone by the persistent kernels (set while long single-CU kernels such as FPS run on another
// stream: a persistent workgroup that cannot be placed until they finish would hold back its whole static tile list)
static std::atomic<int> g_reserved_cus{0};
void set_reserved_cus(int n) { g_reserved_cus.store(n < 0 ? 0 : n); }
int reserved_cus() { return g_reserved_cus.load(); }

int launch_edge_pp(const EdgePcParams& p0, int nblocks, hipStream_t s) {
    EdgePcParams p = p0;
    static const int dbg = [] { const char* e = getenv("MORIG_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n > 8 ? (n / 8) * 8 : 8;
    }();
    p.dbg = dbg;
#ifdef MORIG_PP_TRACE
    static unsigned long long* trace_buf = [] { void* b = nullptr; return hipMalloc(&b, 64 * 8) == hipSuccess ? (unsigned long long*)b : nullptr; }();
    p.trace = trace_buf;
#endif
    int avail = ncu - ((g_reserved_cus.load() + 7) / 8) * 8;
    if (avail < 8) avail = 8;
    const int grid = nblocks < avail ? ((nblocks + 7) / 8) * 8 : avail;  // one persistent workgroup per CU, multiple of 8 (XCDs)
#define PP_LAUNCH(HH, QQ) hipLaunchKernelGGL((edge_pp_kernel<HH, QQ>), dim3(grid), dim3(512), 0, s, p)
    if (p.H == 256 && p.quad) PP_LAUNCH(256, true);
    else if (p.H == 256) PP_LAUNCH(256, false);
    else if (p.H == 128 && p.quad) PP_LAUNCH(128, true);
    else if (p.H == 128) PP_LAUNCH(128, false);
    else return MORIG_E_UNSUPPORTED;
    MORIG_LAUNCH_CHECK();
#ifdef MORIG_PP_TRACE
    {
        unsigned long long h[64];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, p.trace, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int r = 0; r < 2; ++r) {
                fprintf(stderr, "PP_TRACE H=%d quad=%d %s:", p.H, p.quad, r ? "producer" : "consumer");
                for (int q = 1; q < 20; ++q) fprintf(stderr, " %lld", (long long)(h[r * 32 + q] - h[r * 32 + q - 1]));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    return MORIG_OK;
}

}  // namespace morig
