// Fused EdgeConv for H = 128 on the split-fp16 path, ROW-LOCAL persistent kernel: no barrier in the main loop.
//
// edge_pp.hip / edge_ws.hip share every operand tile Z = relu(A[dst] + B[src]) between the waves of a workgroup (one converts, all
// consume), which costs a Z ring in LDS (ds_write at 85 B/clk) and an s_barrier per K-chunk: 8 waves in lock step, everybody waits for
// the slowest, and the two waves of a SIMD stall together. At H = 128 the ablations put 40-48 % of the kernel on exactly that
// (profiles/r02_edge_ws_ablations_b.txt: no barrier 0.877 -> 0.452 ms, no fragment reads -> 0.477 ms).
// Here the roles are cut the other way. At H = 128 the WHOLE second-layer weight W2 (hi and lo halves, 64 KB) fits in LDS, stored once
// per workgroup in MFMA-fragment order (conflict-free linear ds_read_b128). A wave then owns 64 edge rows x ALL 128 output columns:
//   D  it gathers its own rows -- B[src] (64 rows) and A[dst] (16 quads, 4-aligned CSR) -- with LDS-DMA into a PRIVATE ring of four
//      half-slots (8 k each: 2 + 1 wave-instructions), six MFMA units ahead of their use;
//   V  reads them back in the A-operand pattern of v_mfma_f32_32x32x16_f16 (a lane = one row, 4 k of a half-slot), adds, ReLU, splits
//      into fp16 (hi, lo) IN REGISTERS: the operand tile never exists in LDS, nobody else needs it;
//   M  24 MFMAs per 16-k step against the W2 fragments (a ring of four register sets, fetched three units ahead).
// Nothing is shared but the read-only W2 image, so the eight waves drift freely: a wave that waits for a gather or runs its epilogue
// leaves the matrix pipe to its SIMD partner. Epilogue in registers as well: quad max / min (monotone affine), one v_permlane32_swap
// per register pair so that a lane holds the 16 consecutive quads of a column, segmented max by wave-uniform control flow (the quads'
// destination ids are scalars), results leave as stores (tile-straddling segments: integer-atomic float max onto rows pre-set by
// init_boundary_rows, as in edge_ws.hip).
// k order inside a 16-k step: lane half `hi`, element j < 4 -> k0 + 4 hi + j, j >= 4 -> k0 + 8 + 4 hi + (j - 4): each 8-k half-slot is
// consumed by BOTH half-waves at once and can be refilled as soon as its two row tiles are converted. The W2 image is permuted the same.
// Only for 4-aligned CSRs. Reference op: models/basic_modules.py:185-202 (EdgeConvMotion.message/update), second Linear of nn_x / nn_pos.
#include "common.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>

namespace morig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int C> using RC = std::integral_constant<int, C>;

// Y16: whole segments leave as split-fp16 halves (per 32-column chunk 32 hi, then 32 lo: the layout the unit's MLP GEMM DMAs into LDS);
// tile-straddling segments stay fp32 atomics and are rewritten by split_boundary_rows (tile_gemm.hip)
template <bool Y16>
__global__ __launch_bounds__(512, 2) void edge_rl128_kernel(const EdgePcParams p) {
    constexpr int H = 128, BM = 64, MT = 2, NT = 4, NS = H / 16;
    constexpr int WB = NS * NT * 2 * 1024;               // W2 image: [step][column tile][hi | lo][lane] x 16 B = 64 KB
    constexpr int HSLOT = 2048 + 512;                    // one half-slot: 64 B rows x 32 B, then 16 A quads x 32 B
    constexpr int RAWW = 4 * HSLOT;                      // per wave
    constexpr int NDMA = 3, NIDX = 4;                    // wave-instructions per half-slot / per index fetch
    __shared__ __attribute__((aligned(1024))) char smem[WB + 8 * RAWW];
    char* wlds = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform for the compiler: scalar bases, s_cbranch)
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- W2 image (once per workgroup) ----
    for (int i = tid; i < WB / 16; i += 512) {
        const int ln = i & 63, f = i >> 6;
        const int hl = f & 1, nt = (f >> 1) & 3, s = f >> 3;
        const int n = 32 * nt + (ln & 31), h2 = ln >> 5;
        const char* src = reinterpret_cast<const char*>(p.W + (size_t)n * p.ldw) + (s >> 1) * 128 + hl * 64 + ((s & 1) * 16 + 4 * h2) * 2;
        const f32x2 a = *reinterpret_cast<const f32x2*>(src), b = *reinterpret_cast<const f32x2*>(src + 16);
        // rows of W2 whose output column has a NEGATIVE BatchNorm scale are stored negated (sign bits of the four halves of a word pair:
        // exact), so that y = relu(acc + b) * sc + sh is RISING in the accumulator of every column: the epilogue needs max only
        const unsigned flip = p.scale[n] < 0.f ? 0x80008000u : 0u;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4*>(wlds + i * 16) = u32x4{__float_as_uint(a[0]) ^ flip, __float_as_uint(a[1]) ^ flip,
                                                        __float_as_uint(b[0]) ^ flip, __float_as_uint(b[1]) ^ flip};
    }
    __syncthreads();                                     // the only workgroup barrier of the kernel
    // the four column constants of this lane's two output columns (epilogue): col = 32 (pr + 2 hi) + l31
    float eb[2], eg[2], es[2], et[2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int col = 32 * (pr + 2 * hi) + l31;
        es[pr] = p.scale[col]; et[pr] = p.shift[col]; eb[pr] = p.bias[col];
        eg[pr] = es[pr] < 0.f ? -1.f : 1.f;
    }

    // ---- this WAVE's tile list: the 64-row tiles (numbered over the replicas) are cut into RUNS of R = 2^lg consecutive tiles; XCD x owns
    // a contiguous range of runs, the waves of a workgroup take adjacent runs. Inside a run the wave carries a segment that is still
    // open at the end of a tile into the next one (cm0 / cm1 below): only rows that straddle a RUN boundary are shared through atomics
    // (R = 1: every boundary, the [r04] form -- 19 % of the geo graph's rows went through init + atomics then) ----
    const int Etot = p.rowptr[p.n_nodes];
    const int tpr = (Etot + BM - 1) / BM;
    const int T = tpr * p.replicas;
    const int lg = 31 - __builtin_clz(p.run > 0 ? p.run : 1), R = 1 << lg;
    const int NR = (T + R - 1) >> lg;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int r_lo = (int)((long long)NR * xcd / 8), r_hi = (int)((long long)NR * (xcd + 1) / 8);
    const int wi = bi * 8 + wave, nw = nbx * 8;
    const int nruns = (r_hi - r_lo - wi + nw - 1) / nw;
    if (nruns <= 0) return;                              // wave-uniform
    const int n_my = ((nruns - 1) << lg) + min(R, T - ((r_lo + wi + (nruns - 1) * nw) << lg));
    auto tile_of = [&](int j) __attribute__((always_inline)) {
        const int jj = j < n_my ? j : n_my - 1;
        return ((r_lo + wi + (jj >> lg) * nw) << lg) + (jj & (R - 1));
    };
    float cm0 = 0.f, cm1 = 0.f;                          // the open segment carried into the next tile of the run (this lane's 2 columns)
    bool cpart = false;                                  // ... and whether it came in over the run's first boundary (shared row)

    // ---- gather state ----
    // B instruction i covers rows 32 i + (lane >> 1), 16-byte piece lane & 1 of the row's 32 bytes; the A instruction (lanes 0..31)
    // quad lane >> 1. The LDS image is swizzled at the SOURCE: slot (row, piece) holds piece ^ ((row >> 3) & 1)
    const unsigned pz = 16u * (unsigned)((lane & 1) ^ ((lane >> 4) & 1));
    int ns0 = 0, ns1 = 0, nq = 0, nfl = 0, nrow0 = 0, nrep = 0;      // the NEXT tile's indices (in flight / loaded)
    int cq = 0, cfl = 0, crow0 = 0, crep = 0;                        // the tile being computed (epilogue)
    unsigned ob0 = 0, ob1 = 0, oq = 0;                               // the tile being gathered: byte offsets of this lane's rows
    const char* abase = reinterpret_cast<const char*>(p.A);
    const char* bbase = reinterpret_cast<const char*>(p.B);
    // The index fetches are inline assembly on purpose: the compiler's vmcnt bookkeeping does not see the LDS-DMA instructions, so a
    // wait it inserted for a plain load would drain the gather ring. The values are claimed (tie()) behind a counted wait instead.
    auto gload = [&](const int* ptr) __attribute__((always_inline)) {
        int v;
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
        return v;
    };
    auto load_indices = [&](int j) __attribute__((always_inline)) {
        const int t = tile_of(j);
        nrep = t / tpr; nrow0 = (t - nrep * tpr) * BM;
        const int r = nrow0 + (lane >> 1);
        ns0 = gload(p.srcS + min(r, Etot - 1));
        ns1 = gload(p.srcS + min(r + 32, Etot - 1));
        nq = gload(p.dstS + min(nrow0 + 4 * (l31 >> 1), Etot - 1));
        const int fr = lane == 0 ? nrow0 - 1 : nrow0 + BM;           // lane 0: the row above the tile, lane 1: the row below
        nfl = gload(p.dstS + min(max(fr, 0), Etot - 1));
    };
    auto tie = [&]() __attribute__((always_inline)) { asm volatile("" : "+v"(ns0), "+v"(ns1), "+v"(nq), "+v"(nfl)); };
    auto switch_gather = [&]() __attribute__((always_inline)) {    // the loaded tile becomes the one fetched from
        abase = reinterpret_cast<const char*>(p.A + (size_t)nrep * p.rep_in * p.lda);
        bbase = reinterpret_cast<const char*>(p.B + (size_t)nrep * p.rep_in * p.ldb);
        ob0 = (unsigned)ns0 * (unsigned)p.ldb * 4u + pz;             // rows past the end re-read the last edge: finite, ignored (id -1)
        ob1 = (unsigned)ns1 * (unsigned)p.ldb * 4u + pz;
        oq = (unsigned)nq * (unsigned)p.lda * 4u + pz;
    };
    auto switch_compute = [&]() __attribute__((always_inline)) { cq = nq; cfl = nfl; crow0 = nrow0; crep = nrep; };
    char* raww = smem + WB + wave * RAWW;
    const unsigned raww_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)raww);
    // half-slot (s, h) of the gather tile -> ring position (2 s + h) & 3: scalar base + lane offset, LDS address through M0
    auto dma = [&](auto sc, auto hc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value, h = decltype(hc)::value;
        const unsigned d0 = raww_lds + ((2 * s + h) & 3) * HSLOT;
        const char* const sb = bbase + s * 64 + h * 32;
        const char* const sa = abase + s * 64 + h * 32;
        const unsigned vo0 = ob0, vo1 = ob1, vo2 = oq;
        unsigned d2 = d0 + 2048u;
        asm volatile("" : "+s"(d2));                                  // (an SGPR before the divergent branch)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:0" :: "s"(d0), "v"(vo0), "s"(sb) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:0" :: "s"(d0 + 1024u), "v"(vo1), "s"(sb) : "memory");
        if (lane < 32) {                                             // one wave-instruction whatever the exec mask: vmcnt counts 3
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:0" :: "s"(d2), "v"(vo2), "s"(sa) : "memory");
        }
    };

    // ---- conversion: (row tile mt, half h) of a step: this lane's row 32 mt + l31, 4 k -> words 2 h, 2 h + 1 of the A-operand pair ----
    typedef float b32x4 __attribute__((ext_vector_type(4)));
    b32x4 zh[2][MT], zl[2][MT];                           // [step parity][row tile]: 8 halves hi / 8 halves lo
    float amax = 0.f;
    const int rb0 = l31 * 32 + 16 * (hi ^ ((l31 >> 3) & 1));              // B piece `hi` of row l31 (row tile mt: + 1024 mt)
    const int ra0 = 2048 + (l31 >> 2) * 32;                                 // A quad 8 mt + (l31 >> 2), piece hi ^ mt
    auto raw_load = [&](int slot, auto mtc, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        constexpr int mt = decltype(mtc)::value;
        const char* src = raww + slot * HSLOT;
        b = *reinterpret_cast<const f32x4*>(src + mt * 1024 + rb0);
        a = *reinterpret_cast<const f32x4*>(src + ra0 + mt * 256 + 16 * (hi ^ mt));
    };
    auto convert = [&](auto parc, auto mtc, auto hc, const f32x4& a, const f32x4& b) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value, mt = decltype(mtc)::value, h = decltype(hc)::value;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(a[i] + b[i], 0.f);
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            float hb, lb;
            split_pair_f16(v[i], v[i + 1], hb, lb);
            zh[par][mt][2 * h + (i >> 1)] = hb; zl[par][mt][2 * h + (i >> 1)] = lb;
#ifndef RL_NO_AMAX                                             // (measurement build: no range guard)
            { float am = amax; const float u0 = v[i], u1 = v[i + 1]; asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(am) : "v"(u0), "v"(u1)); amax = am; }   // (in place: left to the scheduler, the
                                                                                                 // four values were kept alive and spilled)
#endif
        }
    };

    // ---- MFMA side ----
    f32x16 acc[MT][NT];
    f16x8 wh[2], wl[2];                                   // W2 fragments: unit (s, nt) uses entry nt & 1, fetched one unit ahead
    const char* wfrag = wlds + lane * 16;
    auto load_w = [&](auto sc, auto ntc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value & (NS - 1), nt = decltype(ntc)::value;
        wh[nt & 1] = *reinterpret_cast<const f16x8*>(wfrag + ((s * 4 + nt) * 2 + 0) * 1024);
        wl[nt & 1] = *reinterpret_cast<const f16x8*>(wfrag + ((s * 4 + nt) * 2 + 1) * 1024);
    };
    auto mma = [&](auto parc, auto ntc, auto firstc) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value, nt = decltype(ntc)::value, we = nt & 1;
        constexpr bool first = decltype(firstc)::value != 0;
        f32x16 c0, c1;
        if constexpr (first) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        } else { c0 = acc[0][nt]; c1 = acc[1][nt]; }
        const f16x8 ah0 = __builtin_bit_cast(f16x8, zh[par][0]), al0 = __builtin_bit_cast(f16x8, zl[par][0]);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, zh[par][1]), al1 = __builtin_bit_cast(f16x8, zl[par][1]);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, wh[we], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, wh[we], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, wl[we], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, wl[we], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, wh[we], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, wh[we], c1, 0, 0, 0);
        acc[0][nt] = c0; acc[1][nt] = c1;
    };

    // ---- epilogue of the computed tile: everything in registers ----
    auto epilogue = [&](int j) __attribute__((always_inline)) {
        const bool run_first = (j & (R - 1)) == 0, run_last = (j & (R - 1)) == R - 1 || j == n_my - 1;   // wave-uniform
        // the min / max below read the accumulators from inline assembly: ordered behind the MFMAs and given their wait states by hand
        // (edge_pp.hip write_z_quad; DESIGN section 5, lesson 11)
        {
            f32x16 &c00 = acc[0][0], &c01 = acc[0][1], &c02 = acc[0][2], &c03 = acc[0][3];
            f32x16 &c10 = acc[1][0], &c11 = acc[1][1], &c12 = acc[1][2], &c13 = acc[1][3];
            asm volatile("s_nop 15" : "+v"(c00), "+v"(c01), "+v"(c02), "+v"(c03), "+v"(c10), "+v"(c11), "+v"(c12), "+v"(c13));
        }
        if (p.dbg & 1) return;
        // quad 2 q + hi of row tile mt, column 32 nt + l31: max of the quad's four accumulators (every column is rising, see the W2 image);
        // half-wave exchange: afterwards lanes 0..31 hold column 32 pr + l31, lanes 32..63 column 32 (pr + 2) + l31, and seq[pr][i] is
        // quad i = 8 mt + 2 q + {0, 1} of the tile, i = 0..15 in row order
        float seq[2][16];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float fa, fb, t3;
                    {
                        const float a0 = acc[mt][pr][4 * q], a1 = acc[mt][pr][4 * q + 1], a2 = acc[mt][pr][4 * q + 2], a3 = acc[mt][pr][4 * q + 3];
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(a0), "v"(a1), "v"(a2));
                        asm("v_max_f32 %0, %1, %2" : "=v"(fa) : "v"(t3), "v"(a3));
                    }
                    {
                        const float a0 = acc[mt][pr + 2][4 * q], a1 = acc[mt][pr + 2][4 * q + 1], a2 = acc[mt][pr + 2][4 * q + 2], a3 = acc[mt][pr + 2][4 * q + 3];
                        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(a0), "v"(a1), "v"(a2));
                        asm("v_max_f32 %0, %1, %2" : "=v"(fb) : "v"(t3), "v"(a3));
                    }
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fa), __float_as_uint(fb), false, false);
                    const unsigned ua = sw[0], ub = sw[1];
                    seq[pr][mt * 8 + 2 * q] = __uint_as_float(ua);
                    seq[pr][mt * 8 + 2 * q + 1] = __uint_as_float(ub);
                }
        // segmented max over the 16 quads: the destination ids are wave-uniform (lane 2 q of cq holds quad q's)
        const int prev = __builtin_amdgcn_readlane(cfl, 0), after = __builtin_amdgcn_readlane(cfl, 1);
        const int nvalid = min(16, (Etot - crow0) >> 2);                     // Etot is a multiple of 4: a quad is valid as a whole
        const int first = __builtin_amdgcn_readlane(cq, 0);
        const bool first_cont = crow0 > 0 && prev == first;
        float* obase = p.Y + (size_t)crep * p.rep_out * p.ldy + 32 * (2 * hi) + l31;
        // the affine once per SEGMENT: y = relu(sg x + b) * sc + sh is rising in the stored accumulator x = sg acc, so max commutes with it
        auto flush = [&](int id, float x0, float x1, bool partial) __attribute__((always_inline)) {
            const float m0 = fmaxf(x0 * eg[0] + eb[0], 0.f) * es[0] + et[0];
            const float m1 = fmaxf(x1 * eg[1] + eb[1], 0.f) * es[1] + et[1];
            float* o = obase + (size_t)id * p.ldy;
            if (partial) { atomic_max_f32(o, m0); atomic_max_f32(o + 32, m1); }
            else if constexpr (Y16) {
                // my two columns sit in chunks 2 hi and 2 hi + 1 at position l31: [hi half | +64 B lo half] in each 128-byte chunk.
                // Lane pairs (l31, l31 ^ 1) trade halves (one DPP move each way) so that every lane stores two DWORDS -- the even lane the
                // pair's hi halves, the odd lane its lo halves -- instead of four 2-byte stores (measured: those cost the kernel 4 %)
                float hp, lp;
                split_pair_f16(m0, m1, hp, lp);
                const unsigned hu = __float_as_uint(hp), lu = __float_as_uint(lp);
                const unsigned nh = (unsigned)__builtin_amdgcn_mov_dpp((int)hu, 0xB1, 0xf, 0xf, true);      // quad_perm [1, 0, 3, 2]
                const unsigned nl = (unsigned)__builtin_amdgcn_mov_dpp((int)lu, 0xB1, 0xf, 0xf, true);
                const bool odd = (l31 & 1) != 0;
                const unsigned x = odd ? nl : hu, y = odd ? lu : nh;                  // first / second column of the pair
                const unsigned wa = __builtin_amdgcn_perm(y, x, 0x05040100u);         // chunk 2 hi:     [x.lo16 | y.lo16 << 16]
                const unsigned wb = __builtin_amdgcn_perm(y, x, 0x07060302u);         // chunk 2 hi + 1: [x.hi16 | y.hi16 << 16]
                unsigned* ow = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(o - l31) + 2 * (l31 & ~1) + (odd ? 64 : 0));
                ow[0] = wa; ow[32] = wb;
                float am = amax; asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(am) : "v"(m0), "v"(m1)); amax = am;
            }
            else { o[0] = m0; o[32] = m1; }
        };
        float m0 = seq[0][0], m1 = seq[1][0];
        int id = first;
        bool part = first_cont;
        if (first_cont && !run_first) { m0 = fmaxf(m0, cm0); m1 = fmaxf(m1, cm1); part = cpart; }     // continued inside the run: mine
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            const int di = __builtin_amdgcn_readlane(cq, 2 * i);
            if (i < nvalid) {                                                // wave-uniform
                if (di != id) {
                    flush(id, m0, m1, part);
                    id = di; part = false; m0 = seq[0][i]; m1 = seq[1][i];
                } else { m0 = fmaxf(m0, seq[0][i]); m1 = fmaxf(m1, seq[1][i]); }
            }
        }
        const bool last_cont = crow0 + BM < Etot && id == after;
        if (last_cont && !run_last) { cm0 = m0; cm1 = m1; cpart = part; }                            // goes on in my next tile
        else flush(id, m0, m1, part || last_cont);
    };

    // ---- one unit = the 6 MFMAs of (step s, column tile nt) and its share of the side work: the conversions of step s + 1 ----
    //   unit 0: (mt 0, h 0)   unit 1: (mt 1, h 0) -> half-slot (s + 1, 0) is free: D(s + 3, 0)
    //   unit 2: (mt 0, h 1)   unit 3: (mt 1, h 1) -> half-slot (s + 1, 1) is free: D(s + 3, 1)
    // The raw reads of a conversion are issued ONE UNIT AHEAD (two register pairs, unit parity), the W2 fragments of a unit likewise: a
    // unit never waits for an LDS round trip it has just started (measured: with the reads in the same unit every unit exposed one).
    // vmcnt: the reads issued in unit 1 need D(s + 1, 1) (unit 3 of step s - 2), those issued in unit 3 need D(s + 2, 0) (unit 1 of step
    // s - 1): two half-slot fetches are younger at either point, plus the index fetch of step 0, unit 0 while it is younger
#if defined(RL_NO_DMA) || defined(RL_NO_WAIT)
#define RL_VMWAIT(n) do { } while (0)
#else
#define RL_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
#endif
    f32x4 ra[2], rb[2];
    auto unit = [&](auto sc, auto ntc, int j) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value, nt = decltype(ntc)::value;
        constexpr int par = s & 1, npar = par ^ 1;
        constexpr int mt = nt & 1, h = nt >> 1;                              // this unit converts (mt, h) of step s + 1 from pair nt & 1
        constexpr int n1 = (nt + 1) & 3, sn = s + 1 + (nt + 1) / 4;          // the next unit converts (n1 & 1, n1 >> 1) of step sn
        // (measurement builds, tools/gpu_rl_ablate.sh, results wrong by construction: RL_NO_RAW / RL_NO_CONV / RL_NO_DMA / RL_NO_W / RL_NO_WAIT)
#ifdef RL_NO_W
        asm volatile("" : "+v"(wh[(nt + 1) & 1]), "+v"(wl[(nt + 1) & 1]));
#else
        load_w(RC<s + (nt + 1) / 4>{}, RC<n1>{});
#endif
        if constexpr (nt == 0) {
            if constexpr (s == 0) load_indices(j + 1);
            if constexpr (s == 5) { tie(); switch_gather(); }                // (the index fetch completed behind the wait of step 1, unit 3)
        }
        if constexpr (nt == 1) { constexpr int cnt = 2 * NDMA + (s <= 1 ? NIDX : 0); RL_VMWAIT(cnt); }
        if constexpr (nt == 3) { constexpr int cnt = 2 * NDMA + (s == 0 ? NIDX : 0); RL_VMWAIT(cnt); }
#ifdef RL_NO_RAW
        asm volatile("" : "+v"(ra[(nt + 1) & 1]), "+v"(rb[(nt + 1) & 1]));
#else
        raw_load((2 * sn + (n1 >> 1)) & 3, RC<(n1 & 1)>{}, ra[(nt + 1) & 1], rb[(nt + 1) & 1]);
#endif
#ifdef RL_NO_CONV
        asm volatile("" :: "v"(ra[nt & 1]), "v"(rb[nt & 1]));
#else
        convert(RC<npar>{}, RC<mt>{}, RC<h>{}, ra[nt & 1], rb[nt & 1]);
#endif
        mma(RC<par>{}, ntc, RC<(s == 0 ? 1 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
#ifndef RL_NO_DMA
        // D() last in the unit: the MFMAs are queued (an LDS-DMA costs the issuing wave ~60-180 cycles of issue)
        if constexpr (nt == 1) { dma(RC<(s + 3) & (NS - 1)>{}, RC<0>{}); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (nt == 3) { dma(RC<(s + 3) & (NS - 1)>{}, RC<1>{}); __builtin_amdgcn_sched_barrier(0); }
#endif
    };
    auto step = [&](auto sc, int j) __attribute__((always_inline)) {
        unit(sc, RC<0>{}, j); unit(sc, RC<1>{}, j); unit(sc, RC<2>{}, j); unit(sc, RC<3>{}, j);
    };

    // ---- prologue: tile 0's steps 0..2 in flight, step 0 converted, the first raw pair of step 1 read ----
    load_indices(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tie();
    switch_gather();
    switch_compute();
    dma(RC<0>{}, RC<0>{}); dma(RC<0>{}, RC<1>{}); dma(RC<1>{}, RC<0>{}); dma(RC<1>{}, RC<1>{});
    load_w(RC<0>{}, RC<0>{});
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NDMA) : "memory");
    raw_load(0, RC<0>{}, ra[0], rb[0]);   convert(RC<0>{}, RC<0>{}, RC<0>{}, ra[0], rb[0]);
    raw_load(0, RC<1>{}, ra[0], rb[0]);   convert(RC<0>{}, RC<1>{}, RC<0>{}, ra[0], rb[0]);
    raw_load(1, RC<0>{}, ra[0], rb[0]);   convert(RC<0>{}, RC<0>{}, RC<1>{}, ra[0], rb[0]);
    raw_load(1, RC<1>{}, ra[0], rb[0]);   convert(RC<0>{}, RC<1>{}, RC<1>{}, ra[0], rb[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    dma(RC<2>{}, RC<0>{}); dma(RC<2>{}, RC<1>{});
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NDMA) : "memory");        // D(1, 0) landed
    raw_load(2, RC<0>{}, ra[0], rb[0]);

#pragma unroll 1
    for (int j = 0; j < n_my; ++j) {
        step(RC<0>{}, j); step(RC<1>{}, j); step(RC<2>{}, j); step(RC<3>{}, j);
        step(RC<4>{}, j); step(RC<5>{}, j); step(RC<6>{}, j); step(RC<7>{}, j);
        epilogue(j);
        switch_compute();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
    if (!(amax < 65000.f)) *p.ovf = 1;
}

static int cu_count_rl() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev].load();
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n);
    }
    return n;
}

// nblocks = 64-row tiles x replicas; one persistent workgroup of 8 independent waves per CU
int launch_edge_rl(const EdgePcParams& p0, int ntiles, hipStream_t s) {
    EdgePcParams p = p0;
    static const int dbg = [] { const char* e = getenv("MORIG_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    if (!p.quad || p.H != 128) return MORIG_E_UNSUPPORTED;
    int ncu = cu_count_rl();
    ncu = ncu > 8 ? (ncu / 8) * 8 : 8;
    int avail = ncu - ((reserved_cus() + 7) / 8) * 8;
    if (avail < 8) avail = 8;
    const int nwg = (ntiles + 7) / 8;
    const int grid = nwg < avail ? ((nwg + 7) / 8) * 8 : avail;
    if (p.y16) hipLaunchKernelGGL(edge_rl128_kernel<true>, dim3(grid), dim3(512), 0, s, p);
    else       hipLaunchKernelGGL(edge_rl128_kernel<false>, dim3(grid), dim3(512), 0, s, p);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig
