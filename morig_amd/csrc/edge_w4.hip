// Fused EdgeConv H = 256 on 4-aligned CSRs, W2-stationary, FOUR waves per workgroup -- one per SIMD, 512 registers each.
//
// edge_ws.hip keeps W2 in registers too, but with eight waves at 256 registers a wave can hold only a 32-column slice of it, so
// every row of the operand tile Z is read from LDS by eight waves: 128 of the 184 KB that cross LDS per 64-deep chunk are those
// fragment reads, LDS is busy 1 437 of the 1 536 cycles the chunk's MFMAs take, and the kernel sits at a counter-measured MFMA
// utilisation of 0.62. Here a wave owns 64 output columns -- its W2 slice is 256 registers, which only a
// one-wave-per-SIMD kernel can afford (gfx950: 512 VGPR + AGPR per lane and SIMD) -- so a Z fragment read from LDS feeds 6 MFMAs
// instead of 3 and the fragment traffic halves (64 KB per chunk; 120 KB in total = 940 cycles). What it costs: nobody shares the
// SIMD, so every wait of the wave is a bubble in its matrix pipe; the side work (fragment reads one group ahead, conversion one
// chunk ahead, LDS-DMA last) is therefore spread over the four 12-MFMA groups of a chunk, at most 2-3 instructions per MFMA.
// Tiles, rings, index prefetch, quad epilogue and the segmented-max scan are edge_ws.hip's (64-row tiles, 64-deep chunks, raw ring
// of 3 private stages per wave, Z ring of 2); per accumulator the additions run in the same order, so results are bit-identical.
// Reference op: models/basic_modules.py:185-202 (EdgeConvMotion.message/update), second Linear of nn_x.
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

template <int C> using W4C = std::integral_constant<int, C>;

#ifdef W4_TRACE     // measurement build: s_memtime stamps of one wave of one mid-launch workgroup, one tile (5 per chunk interval)
#define W4_TS(k) do { if (j == 5 && blockIdx.x == 8 && lane == 0) p.trace[wave * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define W4_TS(k) do { } while (0)
#endif

// The issue order of one block (LLVM sched_group_barrier pipeline; masks: 0x8 MFMA, 0x2 VALU, 0x100 DS read, 0x200 DS write): the six
// LDS reads whose data the NEXT block consumes first, then the block's twelve MFMAs with the conversion's VALU work spread between
// them -- at most three other instructions per MFMA: a lone wave hides up to five behind one 32-cycle MFMA (MI355X_MICROARCH.md) --
// and the two LDS writes of the converted piece behind the eighth.
#ifndef W4_NO_PIPE
#define W4_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define W4_PIPE() do { W4_SG(0x100, 6); \
    W4_SG(0x8, 1); W4_SG(0x2, 3); W4_SG(0x8, 1); W4_SG(0x2, 3); W4_SG(0x8, 1); W4_SG(0x2, 3); W4_SG(0x8, 1); W4_SG(0x2, 3); \
    W4_SG(0x8, 1); W4_SG(0x2, 3); W4_SG(0x8, 1); W4_SG(0x2, 3); W4_SG(0x8, 1); W4_SG(0x2, 3); W4_SG(0x8, 1); W4_SG(0x2, 3); \
    W4_SG(0x200, 2); W4_SG(0x8, 1); W4_SG(0x2, 2); W4_SG(0x8, 1); W4_SG(0x2, 2); W4_SG(0x8, 1); W4_SG(0x8, 1); } while (0)
#else
#define W4_PIPE() do { } while (0)
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void edge_w4_kernel(const EdgePcParams p) {
    constexpr int H = 256, BM = 64, KC = 64;
    constexpr int LDB = 4 * KC + 16;                     // bytes per Z row = [KC hi | KC lo | 16 pad]
    constexpr int NW = 4, CT = 2;                        // waves; 32-column tiles per wave
    constexpr int MT = BM / 32;                          // 32-row tiles per wave (every wave reads all rows)
    constexpr int NC = H / KC, SPC = KC / 16, NS = SPC * NC;
    constexpr int RPW = BM / NW;                         // 16 tile rows gathered and converted by one wave
    constexpr int NQ = BM / 4;                           // quad rows per tile
    constexpr int RAWW = 4096 + 1024;                    // raw bytes per wave and stage: 2 row groups x 2 halves of B, then 4 A quads
    constexpr int RAWS = NW * RAWW;
    constexpr int ZSTAGE = BM * LDB;
    constexpr int ZQ = H + 4;
    constexpr int VEC = H / 64;
    constexpr int SWITCHC = NC - 3;                      // chunk whose D() is the first of the NEXT tile
    static_assert(NC == 4 && SPC == 4, "schedule");

    __shared__ __attribute__((aligned(128))) char smem[3 * RAWS + 2 * ZSTAGE + NQ * ZQ * 4 + 3 * 32 * 4 + 64 + 3 * H * 4 + NW * 1024];
    char* raw = smem;
    char* zring = smem + 3 * RAWS;
    float* Z = reinterpret_cast<float*>(zring + 2 * ZSTAGE);
    int* sq_all = reinterpret_cast<int*>(zring + 2 * ZSTAGE + NQ * ZQ * 4);   // [3][32] destination id per quad row
    int* sflag = sq_all + 3 * 32;                                              // [3][2] first / last segment continues
    float* sbias = reinterpret_cast<float*>(sflag + 16);                       // [3][H] bias, BN scale, BN shift
    int* sidx = reinterpret_cast<int*>(sbias + 3 * H);                         // [NW][4][64] a wave's index table of the next gather tile

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- tile list: XCD x (= blockIdx & 7 under round-robin dispatch) owns a contiguous range ----
    const int Etot = p.rowptr[p.n_nodes];
    const int tpr = (Etot + BM - 1) / BM;
    const int T = tpr * p.replicas;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int t_lo = (int)((long long)T * xcd / 8), t_hi = (int)((long long)T * (xcd + 1) / 8);
    const int n_my = (t_hi - t_lo - bi + nbx - 1) / nbx;
    if (n_my <= 0) return;                                                     // block-uniform
    auto tile_of = [&](int j) __attribute__((always_inline)) { return t_lo + bi + (j < n_my ? j : n_my - 1) * nbx; };
    { sbias[tid] = p.bias[tid]; sbias[H + tid] = p.scale[tid]; sbias[2 * H + tid] = p.shift[tid]; }

    const bool all_rising = __ballot(p.scale[64 * wave + lane] >= 0.f) == ~0ull;   // the wave's 64 columns

    // ---- resident W2 slice: column tile ct of this wave = columns 64 wave + 32 ct + l31; fragment layout as edge_ws.hip (Z slot
    // s2 = 2 * step + hi of a chunk holds the chunk's k = 4 s2 + {0..3} and KC/2 + 4 s2 + {0..3}) ----
    f16x8 wh[NS][CT], wl[NS][CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const char* wrow = reinterpret_cast<const char*>(p.W + (size_t)(64 * wave + 32 * ct + l31) * p.ldw);
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            const int s2 = 2 * (S % SPC) + hi;
            const int k0 = (S / SPC) * KC + 4 * s2, k1 = k0 + KC / 2;
            const char* c0 = wrow + (k0 >> 5) * 128 + 2 * (k0 & 31);
            const char* c1 = wrow + (k1 >> 5) * 128 + 2 * (k1 & 31);
            const f16x4 h0 = *reinterpret_cast<const f16x4*>(c0), h1 = *reinterpret_cast<const f16x4*>(c1);
            const f16x4 l0 = *reinterpret_cast<const f16x4*>(c0 + 64), l1 = *reinterpret_cast<const f16x4*>(c1 + 64);
            wh[S][ct] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            wl[S][ct] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    }

    // W2 lives in the AGPR half of the register file (MFMA operands may be AGPRs; nothing else ever reads it), which leaves the
    // accumulators in VGPRs, where the quad epilogue's VALU instructions can read them: left to itself the allocator put the
    // accumulators into AGPRs and paid 128 v_accvgpr moves per tile around write_z
#pragma unroll
    for (int S = 0; S < NS; ++S)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+a"(wh[S][ct]), "+a"(wl[S][ct]));

    // ---- gather state. B: instruction (g, h) = 128-byte half h of the 256-byte row chunk of the 8 rows of row group g (lane >> 3),
    // 16-byte piece lane & 7; A: 4 quads x 16 pieces (all 64 lanes) ----
    const int drow = lane >> 3, dpiece = lane & 7;
    const int aq = lane >> 4, apiece = lane & 15;
    unsigned ob0 = 0, ob1 = 0, oq = 0;                   // byte offsets of row groups 0 / 1 and of the quad (< 4 GB per replica)
    int nrow0 = 0, nrep = 0;
    const char* abase = reinterpret_cast<const char*>(p.A);
    const char* bbase = reinterpret_cast<const char*>(p.B);
    // The indices of a tile travel by LDS-DMA too (four dword instructions per wave into a private 1 KB table), so that EVERY VMEM
    // load of the loop is one whose vmcnt this file counts itself: as ordinary loads their first use (switch_tile) got a compiler-
    // placed s_waitcnt that knows nothing of the LDS-DMA pieces in flight and waited for all of them (~850 cycles per tile, traced).
    // They are issued in a tile's last interval and read two intervals later, behind the counted wait of the interval in between.
    const unsigned sidx_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(sidx + wave * 256));
    auto load_indices = [&](int j) __attribute__((always_inline)) {
        const int t = tile_of(j);
        nrep = t / tpr; nrow0 = (t - nrep * tpr) * BM;
        const int r = nrow0 + RPW * wave + drow;
        const unsigned o0 = 4u * (unsigned)min(r, Etot - 1), o1 = 4u * (unsigned)min(r + 8, Etot - 1);
        const unsigned o2 = 4u * (unsigned)min(nrow0 + 4 * ((RPW / 4) * wave + aq), Etot - 1);
        const int fr = lane == 0 ? nrow0 - 1 : (lane == 1 ? nrow0 + BM - 1 : nrow0 + BM);
        const unsigned o3 = 4u * (unsigned)min(max(fr, 0), Etot - 1);
        const int* const ss = p.srcS; const int* const sd = p.dstS;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(sidx_lds), "v"(o0), "s"(ss) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(sidx_lds + 256u), "v"(o1), "s"(ss) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(sidx_lds + 512u), "v"(o2), "s"(sd) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(sidx_lds + 768u), "v"(o3), "s"(sd) : "memory");
    };
    auto switch_tile = [&](int slot) __attribute__((always_inline)) {           // the loaded tile becomes the one fetched from
        const int* my = sidx + wave * 256 + lane;
        const int ns0 = my[0], ns1 = my[64], nq = my[128], nfl = my[192];
        abase = reinterpret_cast<const char*>(p.A + (size_t)nrep * p.rep_in * p.lda);
        bbase = reinterpret_cast<const char*>(p.B + (size_t)nrep * p.rep_in * p.ldb);
        ob0 = ((unsigned)ns0 * (unsigned)p.ldb + 4u * dpiece) * 4u;           // rows past the end re-read the last edge: finite, ignored
        ob1 = ((unsigned)ns1 * (unsigned)p.ldb + 4u * dpiece) * 4u;           // by the scan (id -1)
        oq = ((unsigned)nq * (unsigned)p.lda + 4u * apiece) * 4u;
        int* sq = sq_all + slot * 32;
        if (apiece == 0) {
            const int q = (RPW / 4) * wave + aq;                                // Etot is a multiple of 4: a quad is valid as a whole
            sq[q] = (nrow0 + 4 * q < Etot) ? nq : -1;
        }
        if (wave == 0) {
            const int prev = __builtin_amdgcn_readlane(nfl, 0), last = __builtin_amdgcn_readlane(nfl, 1),
                      after = __builtin_amdgcn_readlane(nfl, 2), first = __builtin_amdgcn_readlane(nq, 0);
            if (lane == 0) {
                sflag[slot * 2] = (nrow0 > 0 && prev == first) ? 1 : 0;
                sflag[slot * 2 + 1] = (nrow0 + BM < Etot && last == after) ? 1 : 0;
            }
        }
    };
    char* raww = raw + wave * RAWW;
    const unsigned raww_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(raw + wave * RAWW));
    // piece i (0..3: B rows of row group i >> 1, half i & 1; 4: the A quads) of chunk c of the current gather tile -> raw stage rs.
    // ONE instruction per call: a lone wave pays every LDS-DMA issue (~60-100 cycles) as a bubble of its own matrix pipe unless
    // MFMAs are queued in front of it, so the five pieces of a chunk are placed in the MIDDLE of five different MFMA groups.
    auto dma1 = [&](auto cc, auto ic, int rs) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value, i = decltype(ic)::value;
        const unsigned d0 = raww_lds + rs * RAWS + 1024u * i;
        const unsigned vo = i < 2 ? ob0 : (i < 4 ? ob1 : oq);   // (asm operands inside a generic lambda need locals)
        const char* const sb = (i < 4 ? bbase : abase) + c * (KC * 4) + (i < 4 ? 128 * (i & 1) : 0);
        // (the instruction's immediate offset would move BOTH the global and the LDS address: the 128-byte half goes into the base)
#ifndef W4_NO_DMA
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(d0), "v"(vo), "s"(sb) : "memory");
#else
        asm volatile("" :: "s"(d0), "v"(vo), "s"(sb));
#endif
    };
    auto dma = [&](auto cc, int rs) __attribute__((always_inline)) {
        dma1(cc, W4C<0>{}, rs); dma1(cc, W4C<1>{}, rs); dma1(cc, W4C<2>{}, rs); dma1(cc, W4C<3>{}, rs); dma1(cc, W4C<4>{}, rs);
    };
    // conversion of one raw piece pair: thread = (row vrow of row group g, 16-byte raw pieces vq [half 0] and vq + 8 [half 1]) -> Z slot vq
    float amax = 0.f;
    const int vrow = lane >> 3, vq = lane & 7;
    auto raw_load = [&](int rs, auto gc, auto halfc, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, hf = decltype(halfc)::value;
#ifdef W4_NO_CONV
        return;
#endif
        const char* src = raww + rs * RAWS;
        b = *reinterpret_cast<const f32x4*>(src + g * 2048 + hf * 1024 + vrow * 128 + 16 * vq);
        a = *reinterpret_cast<const f32x4*>(src + 4096 + (2 * g + (vrow >> 2)) * 256 + hf * 128 + 16 * vq);
    };
    auto conv_store = [&](int zs, auto gc, auto halfc, const f32x4& a, const f32x4& b) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, hf = decltype(halfc)::value;
#ifdef W4_NO_CONV
        return;
#endif
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(a[i] + b[i], 0.f);
        // hi = fp16(v) truncated, lo = fp16(v - hi) rounded to nearest (v_fma_mix): as edge_ws.hip, bit for bit
        typedef float b32x2 __attribute__((ext_vector_type(2)));
        b32x2 hw, lw;
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const f16x2 h = __builtin_amdgcn_cvt_pkrtz(v[i], v[i + 1]);
            const float hb = __builtin_bit_cast(float, h);
            float lb;
            asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(v[i]), "v"(hb));
            asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(v[i + 1]), "v"(hb));
            hw[i >> 1] = hb; lw[i >> 1] = lb;
            amax = fmaxf(amax, fmaxf(v[i], v[i + 1]));                         // v >= 0 after the ReLU
        }
        char* rowp = zring + zs * ZSTAGE + (RPW * wave + 8 * g + vrow) * LDB + 16 * vq + 8 * hf;
        *reinterpret_cast<b32x2*>(rowp) = hw;
        *reinterpret_cast<b32x2*>(rowp + 2 * KC) = lw;
    };

    // ---- MFMA side ----
    f32x16 acc[MT][CT];
    struct Frag { f16x8 ah[MT], al[MT]; };
    const char* zfrag = zring + l31 * LDB + 16 * hi;
    auto load_frag = [&](Frag& f, int zs, int g) __attribute__((always_inline)) {   // group g of a chunk = its 16-k step g
        const char* b = zfrag + zs * ZSTAGE + 32 * g;
#ifdef W4_NO_FRAG
        asm volatile("" : "+v"(f.ah[0]), "+v"(f.al[0]), "+v"(f.ah[1]), "+v"(f.al[1]));
        return;
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f.ah[mt] = *reinterpret_cast<const f16x8*>(b + mt * 32 * LDB);
            f.al[mt] = *reinterpret_cast<const f16x8*>(b + mt * 32 * LDB + 2 * KC);
        }
    };
    // one 16-k step on the wave's four accumulators; each split term sweeps all four before the next touches them (a lone wave must
    // not wait on its own previous MFMA); per accumulator the order lo*hi, hi*lo, hi*hi is edge_ws.hip's
    auto mma = [&](const Frag& f, auto Sc, auto firstc, auto&& mid) __attribute__((always_inline)) {
        constexpr int S = decltype(Sc)::value;
        constexpr bool first = decltype(firstc)::value != 0;
        f32x16 c[MT][CT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if constexpr (first) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[mt][ct][r] = 0.f;
                } else c[mt][ct] = acc[mt][ct];
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) c[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[mt], wh[S][ct], c[mt][ct], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) c[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mt], wl[S][ct], c[mt][ct], 0, 0, 0);
#ifdef W4_MID_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
        mid();                                               // (an LDS-DMA piece: eight MFMAs queued in front of it, four behind)
#ifdef W4_MID_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) c[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mt], wh[S][ct], c[mt][ct], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = c[mt][ct];
    };
    auto nothing = []() __attribute__((always_inline)) {};
    // y = relu(acc + b) * sc + sh is monotone in acc: the max over a quad's four rows is f(max acc) or f(min acc)
    // ---- epilogue of a finished tile, in two parts.
    // write_z(slot): quad reduction (y = relu(acc + b) * sc + sh is monotone in acc: the max over a quad's four rows is f(max acc)
    // or f(min acc)), then ONE v_permlane32_swap per pair of column tiles brings all sixteen quad rows of one output column into one
    // lane (lanes 0-31: column tile 0, lanes 32-63: column tile 1 -> the lane's column is 64 wave + lane), where the segmented max
    // runs as a sixteen-step chain on wave-uniform segment starts: P[R] = max over the rows of R's segment up to R. P goes to the LDS
    // scan region, so that scan() reads ONE row per segment (its last) instead of walking the segment's rows with an LDS round trip
    // per step -- traced: 1 800 of a tile's 13 000 cycles, fully exposed on a one-wave-per-SIMD kernel.
    // The quads of a 32 x 32 tile are worked on side by side (one after the other the compiler reused one temporary for all sixteen
    // quads and every instruction waited for its predecessor: ~1 200 cycles per tile, traced).
    // what the epilogue of a tile needs besides its accumulators -- the quad rows' destination ids, segment starts, the segments
    // this wave stores (round robin), continuation flags, the replica -- is read and derived in the tile's LAST interval (prep_epi,
    // in the tail behind the interval's MFMAs), so that neither write_z nor scan starts with an LDS round trip and a division
    unsigned eSTART = 0u, eMINE = 0u;
    int esv = -1, erep = 0;
    bool efirst = false, elast = false;
    auto prep_epi = [&](int t, int slot) __attribute__((always_inline)) {
        const int* sq = sq_all + slot * 32;
        const int ql = lane & (NQ - 1);
        esv = sq[ql];
        const int sp = sq[ql > 0 ? ql - 1 : 0];
        efirst = sflag[slot * 2] != 0; elast = sflag[slot * 2 + 1] != 0;
        erep = t / tpr;
        eSTART = (unsigned)__ballot(lane < NQ && (ql == 0 || esv != sp));        // wave-uniform: bit R = a segment starts at quad row R
        const unsigned valid = (unsigned)__ballot(lane < NQ && esv >= 0);
        // the tile's segments are dealt round-robin to the four waves: segment k (in start order) belongs to wave k & 3
        unsigned todo = eSTART & valid, mine = 0u;
        int k = 0;
        while (todo) {
            const unsigned low = todo & (0u - todo);
            if ((k & (NW - 1)) == wave) mine |= low;
            todo ^= low;
            ++k;
        }
        eMINE = mine;
    };
    auto write_z = [&](int slot) __attribute__((always_inline)) {
        // the min / max below read the accumulators from inline assembly: ordered behind the MFMAs, wait states by hand
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) asm volatile("s_nop 15" : "+v"(acc[mt][ct]));
        (void)slot;
        const unsigned START = eSTART;
        float val[NQ];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float x[CT][4];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col = 64 * wave + 32 * ct + l31;
                const float b = sbias[col], sc = sbias[H + col], sh = sbias[2 * H + col];
                float hi4[4], lo4[4], t3[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t3[q]) : "v"(acc[mt][ct][4 * q]), "v"(acc[mt][ct][4 * q + 1]), "v"(acc[mt][ct][4 * q + 2]));
#pragma unroll
                for (int q = 0; q < 4; ++q) asm("v_max_f32 %0, %1, %2" : "=v"(hi4[q]) : "v"(t3[q]), "v"(acc[mt][ct][4 * q + 3]));
                if (all_rising) {                          // wave-uniform: every BatchNorm scale of this wave's columns is >= 0 -> max only
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[ct][q] = hi4[q];
                } else {
                    const bool rising = sc >= 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t3[q]) : "v"(acc[mt][ct][4 * q]), "v"(acc[mt][ct][4 * q + 1]), "v"(acc[mt][ct][4 * q + 2]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm("v_min_f32 %0, %1, %2" : "=v"(lo4[q]) : "v"(t3[q]), "v"(acc[mt][ct][4 * q + 3]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[ct][q] = rising ? hi4[q] : lo4[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) x[ct][q] = fmaxf(x[ct][q] + b, 0.f) * sc + sh;
            }
            // v_permlane32_swap(vdst, src1): lanes 32-63 of vdst <-> lanes 0-31 of src1. With vdst = the column-tile-0 value and src1 =
            // the column-tile-1 value of quad rows 8 mt + 2 q + hi: afterwards [0] = the EVEN row, [1] = the ODD row of column tile
            // (lane >> 5) in every lane
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0][q]), __float_as_uint(x[1][q]), false, false);
                const unsigned e0 = sw[0], e1 = sw[1];
                val[8 * mt + 2 * q] = __uint_as_float(e0);
                val[8 * mt + 2 * q + 1] = __uint_as_float(e1);
            }
        }
        // running segmented max over the sixteen quad rows of this lane's column (segment starts are wave-uniform: scalar selects)
        float* zc = Z + 64 * wave + lane;
        float m = val[0];
        zc[0] = m;
#pragma unroll
        for (int R = 1; R < NQ; ++R) {
            const bool st = (START >> R) & 1u;
            m = st ? val[R] : fmaxf(m, val[R]);
            zc[R * ZQ] = m;
        }
    };
    // the segments of a finished tile -> global memory: a segment's result is row (its last quad row) of the scan region; the tile's
    // segments are dealt round-robin to the four waves (segment k in start order belongs to wave k & 3); a lane holds VEC adjacent columns
    auto scan = [&](int t, int slot) __attribute__((always_inline)) {
        typedef float fvec __attribute__((ext_vector_type(VEC)));
        (void)t; (void)slot;
        if (p.dbg & 1) return;
        const float* zl = Z + VEC * lane;
        float* obase = p.Y + (size_t)erep * p.rep_out * p.ldy + VEC * lane;
        const unsigned START = eSTART;
#ifdef W4_TRACE
        if (t == tile_of(4) && blockIdx.x == 8 && lane == 0) p.trace[wave * 32 + 22] = __builtin_readcyclecounter();
#endif
        unsigned mine = eMINE;
        while (mine) {                                                       // wave-uniform: SALU bit walking
            const int b = __builtin_ctz(mine);
            mine &= mine - 1u;
            const unsigned later = b < 31 ? (START & ~((2u << b) - 1u)) : 0u;
            const int e = later ? __builtin_ctz(later) : NQ;                 // the segment covers quad rows [b, e)
            const int sg = __builtin_amdgcn_readlane(esv, b);
            const fvec m = *reinterpret_cast<const fvec*>(zl + (e - 1) * ZQ);
            float* o = obase + (size_t)sg * p.ldy;
            const bool partial = (b == 0 && efirst) || (e == NQ && elast);
            if (partial && !(p.dbg & 32)) {               // (dbg 32: timing experiment -- plain stores, wrong results on shared rows)
#pragma unroll
                for (int v = 0; v < VEC; ++v) atomic_max_f32(o + v, m[v]);
            } else {
                *reinterpret_cast<fvec*>(o) = m;
            }
        }
    };

    // ---- prologue: tile 0's chunks 0..2 in flight, chunk 0 converted ----
    load_indices(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    switch_tile(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the table is read before it is refilled)
    load_indices(1);                                      // the gather switches tiles at chunk 1: indices are loaded a tile ahead
    dma(W4C<0>{}, 0); dma(W4C<1>{}, 1); dma(W4C<2>{}, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        f32x4 a, b;
        raw_load(0, W4C<0>{}, W4C<0>{}, a, b); conv_store(0, W4C<0>{}, W4C<0>{}, a, b);
        raw_load(0, W4C<0>{}, W4C<1>{}, a, b); conv_store(0, W4C<0>{}, W4C<1>{}, a, b);
        raw_load(0, W4C<1>{}, W4C<0>{}, a, b); conv_store(0, W4C<1>{}, W4C<0>{}, a, b);
        raw_load(0, W4C<1>{}, W4C<1>{}, a, b); conv_store(0, W4C<1>{}, W4C<1>{}, a, b);
    }
    int rs = 0;                                           // raw stage of the chunk whose MFMAs run next (g % 3)
    Frag F0, F1;
    f32x4 qa, qb;                                         // raw pieces of the NEXT chunk's first conversion unit (read one interval ahead)
    raw_load(1, W4C<0>{}, W4C<0>{}, qa, qb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // B_0: Z chunk 0, sbias, sq[0] visible

    // One chunk interval [B_c, B_c+1) of tile j: four blocks of ONE 12-MFMA group each (sched_barrier between them); every block
    // first issues the LDS reads whose data the NEXT block consumes (fragments of the next group, raw pieces of the next conversion
    // unit), then the VALU work on data read one block earlier (one of the four conversion units of chunk g+1), interleaved with
    // its MFMAs; the five LDS-DMA instructions of chunk g+3 come last, behind the interval's last queued MFMA.
    auto chunk = [&](auto cc, int j) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        constexpr int zs = c & 1;
        constexpr int cd = (c + 3) % NC;                   // the chunk this interval fetches (of the NEXT tile from c == SWITCHC on)
        const int rs_v = rs == 2 ? 0 : rs + 1;             // raw stage of chunk g+1 (converted in this interval)
        const int rs_w = rs_v == 2 ? 0 : rs_v + 1;         // ... of chunk g+2 (its first unit is read at the end of this interval)
        const int rs_d = rs;                               // raw(g) was converted during the previous interval: its stage is free
        f32x4 ra0, rb0, ra1, rb1;
        if constexpr (c == SWITCHC) switch_tile((j + 1) % 3);
        W4_TS(5 * c);
        // ---- block 0: pending group (step 3 of the previous chunk) + conversion unit 0 (its raw pieces were read in the previous
        // interval's last block)
        load_frag(F0, zs, 0);
        raw_load(rs_v, W4C<0>{}, W4C<1>{}, ra1, rb1);
        conv_store(zs ^ 1, W4C<0>{}, W4C<0>{}, qa, qb);
        if (j > 0 || c > 0) {
            constexpr int cp = (c + NC - 1) % NC;
            mma(F1, W4C<SPC * cp + SPC - 1>{}, W4C<0>{}, [&]() __attribute__((always_inline)) { dma1(W4C<cd>{}, W4C<0>{}, rs_d); });
            W4_PIPE();
            if constexpr (c == 0) { __builtin_amdgcn_sched_barrier(0); if (!(p.dbg & 1)) write_z((j + 2) % 3); }
        } else dma1(W4C<cd>{}, W4C<0>{}, rs_d);
        __builtin_amdgcn_sched_barrier(0);
        W4_TS(5 * c + 1);
        // ---- block 1: group 0 + unit 1
        load_frag(F1, zs, 1);
        raw_load(rs_v, W4C<1>{}, W4C<0>{}, ra0, rb0);
        conv_store(zs ^ 1, W4C<0>{}, W4C<1>{}, ra1, rb1);
        mma(F0, W4C<SPC * c>{}, W4C<(c == 0 ? 1 : 0)>{}, [&]() __attribute__((always_inline)) { dma1(W4C<cd>{}, W4C<1>{}, rs_d); });
        W4_PIPE();
        __builtin_amdgcn_sched_barrier(0);
        W4_TS(5 * c + 2);
        // ---- block 2: group 1 + unit 2
        load_frag(F0, zs, 2);
        raw_load(rs_v, W4C<1>{}, W4C<1>{}, ra1, rb1);
        conv_store(zs ^ 1, W4C<1>{}, W4C<0>{}, ra0, rb0);
        mma(F1, W4C<SPC * c + 1>{}, W4C<0>{}, [&]() __attribute__((always_inline)) { dma1(W4C<cd>{}, W4C<2>{}, rs_d); });
        W4_PIPE();
        __builtin_amdgcn_sched_barrier(0);
        W4_TS(5 * c + 3);
        // ---- block 3: group 2 + unit 3; the first raw pieces of chunk g+2 (D(g+2) was issued an interval ago: all but this interval's
        // three pieces must have retired)
        load_frag(F1, zs, 3);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        raw_load(rs_w, W4C<0>{}, W4C<0>{}, qa, qb);
        conv_store(zs ^ 1, W4C<1>{}, W4C<1>{}, ra1, rb1);
        mma(F0, W4C<SPC * c + 2>{}, W4C<0>{}, [&]() __attribute__((always_inline)) { dma1(W4C<cd>{}, W4C<3>{}, rs_d); });
        W4_PIPE();
        __builtin_amdgcn_sched_barrier(0);
        W4_TS(5 * c + 4);
        dma1(W4C<cd>{}, W4C<4>{}, rs_d);                   // the A quads last
        if constexpr (c == 1) W4_TS(23);
        if constexpr (c == 3) { load_indices(j + 2); prep_epi(tile_of(j), j % 3); }
        rs = rs_v;
        if constexpr (c == 1) { if (j > 0) scan(tile_of(j - 1), (j - 1) % 3); }
        if constexpr (c == 1) W4_TS(21);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef W4_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
    };
#pragma unroll 1
    for (int j = 0; j < n_my; ++j) {
        chunk(W4C<0>{}, j);
        chunk(W4C<1>{}, j);
        chunk(W4C<2>{}, j);
        chunk(W4C<3>{}, j);
        W4_TS(20);
    }
    // ---- drain: last group of the last tile, its epilogue ----
    mma(F1, W4C<NS - 1>{}, W4C<0>{}, nothing);
    if (!(p.dbg & 1)) write_z((n_my - 1) % 3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             // no LDS-DMA may outlive the workgroup
    __builtin_amdgcn_s_barrier();
    scan(tile_of(n_my - 1), (n_my - 1) % 3);
    if (!(amax < 65000.f)) *p.ovf = 1;
}

static int w4_cu_count() {
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

int launch_edge_w4(const EdgePcParams& p0, int nblocks, hipStream_t s) {
    EdgePcParams p = p0;
    static const int dbg = [] { const char* e = getenv("MORIG_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    if (!p.quad || p.H != 256) return MORIG_E_UNSUPPORTED;
    int ncu = w4_cu_count();
    ncu = ncu > 8 ? (ncu / 8) * 8 : 8;
    int avail = ncu - ((reserved_cus() + 7) / 8) * 8;
    if (avail < 8) avail = 8;
    const int grid = nblocks < avail ? ((nblocks + 7) / 8) * 8 : avail;      // one persistent workgroup per CU, multiple of 8 (XCDs)
#ifdef W4_TRACE
    static unsigned long long* trace_buf = [] { void* b = nullptr; return hipMalloc(&b, 128 * 8) == hipSuccess ? (unsigned long long*)b : nullptr; }();
    p.trace = trace_buf;
#endif
    hipLaunchKernelGGL(edge_w4_kernel, dim3(grid), dim3(256), 0, s, p);
    MORIG_LAUNCH_CHECK();
#ifdef W4_TRACE
    {
        unsigned long long h[128];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, p.trace, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int w = 0; w < 4; ++w) {
                const unsigned long long* g = h + 32 * w;
                fprintf(stderr, "W4_TRACE wave %d:", w);
                for (int q = 1; q <= 20; ++q) fprintf(stderr, " %lld%s", (long long)(g[q] - g[q - 1]), q % 5 == 0 ? " |" : "");
                fprintf(stderr, "  tile %lld  chunk-1 tail from stamp 9: dma-issued@%lld scan-preamble-done@%lld scan-end@%lld\n", (long long)(g[20] - g[0]),
                        (long long)(g[23] - g[9]), (long long)(g[22] - g[9]), (long long)(g[21] - g[9]));
            }
        }
    }
#endif
    return MORIG_OK;
}

}  // namespace morig
