// Fused EdgeConv for the wide layers (H = 128, 256) on the split-fp16 path, W2-STATIONARY persistent kernel.
//
// edge_pp.hip streams the W2 chunk of every 128-row tile through the producer waves: 32 of the 52 KB that cross the
// L2 -> CU path per K-chunk and 8 of the 13 VMEM wave-instructions per producer and chunk are W2, and VMEM issue beside
// a busy matrix pipe is what bounds that kernel (chunk period 2 850 cycles for 1 536 of MFMA).
// Here the roles are turned round: W2 never moves. All 8 waves are MFMA waves; wave (wm, wn) owns output columns
// [32 wn, 32 wn + 32) for ALL K and keeps that slice of W2 -- hi and lo halves, H/16 k-steps x 8 registers = 128
// VGPRs at H = 256 -- in registers for the whole launch (W2 is the B operand of v_mfma_f32_32x32x16_f16, the operand
// tile Z = relu(A[dst] + B[src]) the A operand). What streams is only the gathered rows:
//   D(g)  each wave fetches the rows it will convert itself -- 16 edge rows of B[src] and its 4 quads of A[dst]
//         (4-aligned CSR) per 32-k chunk -- with global_load_lds (LDS-DMA, no VGPR round trip) into a PRIVATE 3-stage
//         raw ring: 3 VMEM wave-instructions per wave and chunk (24 per CU against 52);
//   V(g)  converts raw fp32 -> add, ReLU, fp16 hi/lo split -> the shared Z ring (2 stages), one chunk ahead;
//   M(g)  every wave reads all 128 rows of the Z chunk (ds_read_b128 fragments) against its resident W2 slice.
// One s_barrier per chunk (Z hand-over); the raw ring needs none (same-wave producer and consumer, counted vmcnt).
// The last MFMA group of chunk g is issued AFTER barrier g+1 together with the first fragment loads of chunk g+1, so
// the matrix pipe has work while those loads are in flight. The accumulators of a finished tile are quad-reduced
// straight into an LDS scan region (monotone epilogue: max or min of the 4 rows of a quad, affine applied once) and the
// segmented max + global stores of tile j run inside chunk 1 of tile j+1, beside its MFMAs.
//
// Only for 4-aligned CSRs (MORIG_CSR_PAD4); everything else stays on edge_pp.hip / tile_gemm.hip.
// Reference op: models/basic_modules.py:185-202 (EdgeConvMotion.message/update), second Linear of nn_x / nn_pos.
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
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

template <int C> using WC = std::integral_constant<int, C>;

#ifdef MORIG_WS_TRACE
#define WS_TS(k) do { if (j == 3 && blockIdx.x == 8 && lane == 0) p.trace[(k) * 8 + wave] = __builtin_readcyclecounter(); } while (0)
#else
#define WS_TS(k) do { } while (0)
#endif

// Tile geometry. The register file is the constraint: W2 (H/16 steps x 8 VGPRs) + accumulators + two fragment sets must
// leave room for the conversion, so H = 256 takes 64-row tiles with 64-deep chunks (accumulators 32 VGPRs, W2 128) and
// H = 128 takes 128-row tiles with 32-deep chunks (waves 2 x 4: accumulators 32, W2 64). Either way a chunk is 16 KB of
// gathered B rows + 4 KB of A quads per workgroup.
template <int H> struct WsGeom { static constexpr int BM = H == 256 ? 64 : 128, KC = H == 256 ? 64 : 32; };

// Y16: whole segments leave as split-fp16 halves (per 32-column chunk 32 hi, then 32 lo: the layout the unit's MLP GEMM DMAs into LDS);
// tile-straddling segments stay fp32 atomics and are rewritten by split_boundary_rows (tile_gemm.hip)
// MIX [r06]: the CSR's segments are NOT 4-aligned, only at least 4 rows long (MORIG_CSR_MIN4): a quad of 4 consecutive tile rows then holds
// the rows of at most TWO segments -- `s` leading rows of the segment of its first row, 4 - s rows of the next one -- and at most one
// segment starts in it. What changes: a wave gathers the A row of the FIRST and of the LAST row of each of its quads (4 instead of 2 A
// rows per chunk: the same one wave-instruction, now unmasked) and a converting lane adds the one its row belongs to; the quad max
// becomes two values (T = max over the first segment's rows, Hd = over the second's: two planes of the scan region) wherever one of a
// register pair's two quads is mixed (a wave-uniform test; pure quads keep the 2-instruction form); the segmented max starts a mid-quad
// segment from the Hd plane and ends the segment in front of it with that quad's T. No padded rows: 9-14 % fewer MFMAs on the rig graphs.
template <int H, bool Y16 = false, bool MIX = false>
__global__ __launch_bounds__(512, 2) void edge_ws_kernel(const EdgePcParams p) {
    static_assert(!MIX || (H == 256 && Y16), "the mixed-quad form exists for the H = 256 split-rows kernel");
    constexpr int BM = WsGeom<H>::BM, KC = WsGeom<H>::KC;
    constexpr int LDB = 4 * KC + 16;                     // bytes per Z row = [KC hi | KC lo | 16 pad]
    constexpr int NWN = H / 32;                          // waves along the output columns
    constexpr int NWM = 8 / NWN;                         // waves along the tile rows
    constexpr int MT = BM / 32 / NWM;                    // 32-row MFMA tiles per wave
    static_assert(MT == 2, "one MFMA group = 2 row tiles x 3 MFMAs");
    constexpr int NC = H / KC;                           // K-chunks per tile
    constexpr int SPC = KC / 16;                         // 16-k MFMA steps (= groups) per chunk
    constexpr int NS = SPC * NC;                         // steps per tile
    constexpr int NG = SPC;
    constexpr int NDMA = 3;                              // LDS-DMA instructions per wave and chunk
    constexpr int RPW = BM / 8;                          // tile rows gathered and converted by one wave
    constexpr int NQ = BM / 4;                           // quad rows per tile
    constexpr int RAWW = 2048 + (MIX ? 1024 : 512);      // raw bytes per wave and stage: B rows, then A quads (MIX: first and last row's)
    constexpr int RAWS = 8 * RAWW;
    constexpr int ZSTAGE = BM * LDB;
    constexpr int ZQ = H + 4;                            // scan region: NQ quad rows x H columns (+4: 16-byte rows)
    constexpr int VEC = H / 64;
    constexpr int SWITCHC = NC >= 3 ? NC - 3 : (NC + NC - 3) % NC;   // chunk whose D() is the first of the NEXT tile
    static_assert(NC == 4, "index prefetch schedule");

    constexpr int NPL = MIX ? 2 : 1;                     // planes of the scan region (MIX: T, then Hd)
    __shared__ __attribute__((aligned(128))) char smem[3 * RAWS + 2 * ZSTAGE + NPL * NQ * ZQ * 4 + 3 * 32 * 4 + 64 + 3 * H * 4 + 2 * H * 4 + 16 + 32];
    char* raw = smem;
    char* zring = smem + 3 * RAWS;
    float* Z = reinterpret_cast<float*>(zring + 2 * ZSTAGE);
    int* sq_all = reinterpret_cast<int*>(zring + 2 * ZSTAGE + NPL * NQ * ZQ * 4);   // [3][32] destination id per quad row (MIX: [16] first row's, [16] last row's)
    int* sflag = sq_all + 3 * 32;                                              // [3][2] first / last segment continues
    float* sbias = reinterpret_cast<float*>(sflag + 16);                       // [3][H] bias, BN scale, BN shift
    float* carry = sbias + 3 * H;                                              // [2][H] open segment handed to the next tile of the run (tile parity)
    int* cshared = reinterpret_cast<int*>(carry + 2 * H);                      // [2] ... and whether it entered over the run's first boundary
    unsigned* ssp = reinterpret_cast<unsigned*>(cshared + 4);                  // MIX [3][2]: byte w = wave w's quads: (s - 1) of quad 2 w | (s - 1) of quad 2 w + 1 << 2

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wn = wave % NWN, wm = wave / NWN;

    // ---- tile list: XCD x (= blockIdx & 7 under round-robin dispatch) owns a contiguous range ----
    const int Etot = p.rowptr[p.n_nodes];
    const int tpr = (Etot + BM - 1) / BM;
    const int T = tpr * p.replicas;
    // [r05] the tiles (numbered over the replicas) are cut into RUNS of R = 2^lg consecutive tiles and a workgroup works through whole
    // runs: a segment that is still open at the end of a tile is handed to the next tile through `carry` instead of going through
    // atomics -- only rows that straddle a RUN boundary are shared between workgroups (R = 1: every boundary, the [r04] form)
    const int lg = 31 - __builtin_clz(p.run > 0 ? p.run : 1), R = 1 << lg;
    const int NR = (T + R - 1) >> lg;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int r_lo = (int)((long long)NR * xcd / 8), r_hi = (int)((long long)NR * (xcd + 1) / 8);
#ifdef WS_OLD_MAP                                       // (measurement build, valid with MORIG_EDGE_RUN=1 only: the [r04] list)
    const int t_lo = (int)((long long)T * xcd / 8), t_hi = (int)((long long)T * (xcd + 1) / 8);
    const int n_my = (t_hi - t_lo - bi + nbx - 1) / nbx;
    if (n_my <= 0) return;
    auto tile_of = [&](int j) __attribute__((always_inline)) { return t_lo + bi + (j < n_my ? j : n_my - 1) * nbx; };
    (void)r_lo; (void)r_hi;
#else
    const int nruns = (r_hi - r_lo - bi + nbx - 1) / nbx;
    if (nruns <= 0) return;                                                    // block-uniform
    const int n_my = ((nruns - 1) << lg) + min(R, T - ((r_lo + bi + (nruns - 1) * nbx) << lg));
    auto tile_of = [&](int j) __attribute__((always_inline)) {
        const int jj = j < n_my ? j : n_my - 1;
        return ((r_lo + bi + (jj >> lg) * nbx) << lg) + (jj & (R - 1));
    };
#endif
    if (tid < H) { sbias[tid] = p.bias[tid]; sbias[H + tid] = p.scale[tid]; sbias[2 * H + tid] = p.shift[tid]; }

    // ---- resident W2 slice. Z slot s2 = 2 * step + hi of a chunk holds the chunk's k = 4 s2 + {0..3} and KC/2 + 4 s2 + {0..3}
    // (what a converting thread produces from its two 16-byte raw pieces), so the W2 fragment is built the same way.
    // W2 memory image: 32-k chunks of [32 hi | 32 lo] halves (packing.split_f16) ----
    f16x8 wh[NS], wl[NS];
    {
        const char* wrow = reinterpret_cast<const char*>(p.W + (size_t)(32 * wn + l31) * p.ldw);
#pragma unroll
        for (int S = 0; S < NS; ++S) {
            const int s2 = 2 * (S % SPC) + hi;
            const int k0 = (S / SPC) * KC + 4 * s2, k1 = k0 + KC / 2;
            const char* c0 = wrow + (k0 >> 5) * 128 + 2 * (k0 & 31);
            const char* c1 = wrow + (k1 >> 5) * 128 + 2 * (k1 & 31);
            const f16x4 h0 = *reinterpret_cast<const f16x4*>(c0), h1 = *reinterpret_cast<const f16x4*>(c1);
            const f16x4 l0 = *reinterpret_cast<const f16x4*>(c0 + 64), l1 = *reinterpret_cast<const f16x4*>(c1 + 64);
            wh[S] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            wl[S] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
        // [r05] a column whose BatchNorm scale is negative keeps its W2 row NEGATED (exact): y = relu(acc + b) * sc + sh is then rising
        // in the stored accumulator x = sg * acc of EVERY column, so the epilogue takes max only (no min, no select) and applies the
        // affine once per segment, after the segmented max (edge_rl.hip does the same)
        if (p.scale[32 * wn + l31] < 0.f) {
#pragma unroll
            for (int S = 0; S < NS; ++S) { wh[S] = -wh[S]; wl[S] = -wl[S]; }
        }
    }

    // ---- gather state: this lane's slice of D().
    // KC = 64: B instruction i = 128-byte half i of the 256-byte row chunk of 8 rows (lane>>3), A: 2 quads x 16 pieces;
    // KC = 32: B instruction i = rows 8 i + (lane>>3) x 8 pieces, A: 4 quads x 8 pieces. ----
    const int drow = lane >> 3, dpiece = lane & 7;
    const int aq = KC == 64 ? (lane >> 4) & 1 : drow & 3, apiece = KC == 64 ? lane & 15 : dpiece;    // lanes 0..31
    unsigned ob0 = 0, ob1 = 0, oq = 0;                   // byte offsets (< 4 GB per replica)
    int ns0 = 0, ns1 = 0, nq = 0, nfl = 0;               // next tile's indices, in flight (MIX: nq = the destination of THIS lane's row)
    int snext = 0x44;                                    // MIX (scalar): s of the gather tile's two quads of this wave, s0 | s1 << 4
    int nrow0 = 0, nrep = 0;
    const char* abase = reinterpret_cast<const char*>(p.A);
    const char* bbase = reinterpret_cast<const char*>(p.B);
    auto load_indices = [&](int j) __attribute__((always_inline)) {
        const int t = tile_of(j);
        nrep = t / tpr; nrow0 = (t - nrep * tpr) * BM;
        const int r = nrow0 + RPW * wave + drow;
        ns0 = p.srcS[min(r, Etot - 1)];
        if constexpr (KC == 32) ns1 = p.srcS[min(r + 8, Etot - 1)];
        if constexpr (MIX) nq = p.dstS[min(r, Etot - 1)];                      // per ROW (rows past the end repeat the last edge)
        else nq = p.dstS[min(nrow0 + 4 * ((RPW / 4) * wave + aq), Etot - 1)];
        // wave 0, lanes 0..2: the ids just outside / at the end of the tile (segment continuation flags)
        const int fr = lane == 0 ? nrow0 - 1 : (lane == 1 ? nrow0 + BM - 1 : nrow0 + BM);
        nfl = p.dstS[min(max(fr, 0), Etot - 1)];
    };
    auto switch_tile = [&](int slot) __attribute__((always_inline)) {           // the loaded tile becomes the one fetched from
        abase = reinterpret_cast<const char*>(p.A + (size_t)nrep * p.rep_in * p.lda);
        bbase = reinterpret_cast<const char*>(p.B + (size_t)nrep * p.rep_in * p.ldb);
        ob0 = ((unsigned)ns0 * (unsigned)p.ldb + 4u * dpiece) * 4u;           // rows past the end re-read the last edge:
        if constexpr (KC == 32) ob1 = ((unsigned)ns1 * (unsigned)p.ldb + 4u * dpiece) * 4u;   // finite, ignored by the scan (id -1)
        else ob1 = ob0 + 128u;
        int* sq = sq_all + slot * 32;
        if constexpr (MIX) {
            // rows 0..3 of quad 0 sit in lanes 0, 8, 16, 24 (8 lanes per row), those of quad 1 in lanes 32 .. 56
            const int dF0 = __builtin_amdgcn_readlane(nq, 0), dL0 = __builtin_amdgcn_readlane(nq, 24),
                      dF1 = __builtin_amdgcn_readlane(nq, 32), dL1 = __builtin_amdgcn_readlane(nq, 56);
            const unsigned long long eq = __ballot(nq == (lane < 32 ? dF0 : dF1));
            const int s0 = __builtin_popcount((unsigned)eq & 0x01010101u), s1 = __builtin_popcount((unsigned)(eq >> 32) & 0x01010101u);
            snext = s0 | (s1 << 4);
            // A rows of this wave: [quad 0 first | quad 0 last | quad 1 first | quad 1 last] x 16 pieces = the 64 lanes
            const int arow = lane >> 4;
            const int aid = arow == 0 ? dF0 : (arow == 1 ? dL0 : (arow == 2 ? dF1 : dL1));
            oq = ((unsigned)aid * (unsigned)p.lda + 4u * (unsigned)(lane & 15)) * 4u;
            if (lane == 0) {
                const int qa = 2 * wave;
                const bool v0 = nrow0 + 4 * qa < Etot, v1 = nrow0 + 4 * (qa + 1) < Etot;      // a quad counts while its FIRST row exists
                sq[qa] = v0 ? dF0 : -1; sq[16 + qa] = v0 ? dL0 : -1;
                sq[qa + 1] = v1 ? dF1 : -1; sq[16 + qa + 1] = v1 ? dL1 : -1;
                reinterpret_cast<unsigned char*>(ssp + slot * 2)[wave] = (unsigned char)((s0 - 1) | ((s1 - 1) << 2));
            }
        } else {
            oq = ((unsigned)nq * (unsigned)p.lda + 4u * apiece) * 4u;
            if (lane < 32 && apiece == 0) {
                const int q = (RPW / 4) * wave + aq;                            // Etot is a multiple of 4: a quad is valid as a whole
                sq[q] = (nrow0 + 4 * q < Etot) ? nq : -1;
            }
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
    // D(): scalar base (SGPR pair) + 32-bit lane offset + immediate chunk offset; the LDS destination goes through M0. The
    // builtin only selects the 64-bit-VGPR address form (2 v_lshl_add_u64 + moves per instruction), hence the asm.
    const unsigned raww_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(raw + wave * RAWW));
    auto dma = [&](auto cc, int rs) __attribute__((always_inline)) {          // chunk c of the current gather tile -> raw stage rs
        constexpr int c = decltype(cc)::value;
#ifdef WS_BUILTIN_DMA
        char* dst = raww + rs * RAWS;
        unsigned o0 = ob0, o1 = ob1, o2 = oq;
        asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2));
        __builtin_amdgcn_global_load_lds((glb_void_t*)(bbase + c * (KC * 4) + o0), (lds_void_t*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void_t*)(bbase + c * (KC * 4) + o1), (lds_void_t*)(dst + 1024), 16, 0, 0);
        if (MIX || lane < 32) __builtin_amdgcn_global_load_lds((glb_void_t*)(abase + c * (KC * 4) + o2), (lds_void_t*)(dst + 2048), 16, 0, 0);
#else
        const unsigned d0 = raww_lds + rs * RAWS;
        const unsigned vo0 = ob0, vo1 = ob1, vo2 = oq;
        // NB the instruction's immediate offset would move BOTH the global and the LDS address: the chunk offset goes into the base
        const char* const sb = bbase + c * (KC * 4); const char* const sa = abase + c * (KC * 4);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                     :: "s"(d0), "v"(vo0), "s"(sb), "n"(0) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                     :: "s"(d0 + 1024u), "v"(vo1), "s"(sb), "n"(0) : "memory");
        // one wave-instruction whatever the exec mask: vmcnt counts 3 per chunk (MIX: all 64 lanes -- 4 A rows)
        if (MIX || lane < 32) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                         :: "s"(d0 + 2048u), "v"(vo2), "s"(sa), "n"(0) : "memory");
        }
#endif
    };
    // conversion of one raw chunk: thread = (row vrow of this wave's RPW, 16-byte raw pieces vq and vq + KC/8) -> Z slot vq.
    // Two halves (one raw piece each) keep the transient registers down.
    float amax = 0.f;
    constexpr int TPR = KC / 8;                           // converting threads per row
    const int vrow = lane / TPR, vq = lane % TPR;
    const int vb = vrow * 128 + 16 * vq;
    int va = 2048 + (vrow >> 2) * (KC * 4) * (MIX ? 2 : 1) + 16 * vq;        // MIX: per tile -- the first or the last row's A of my quad
    // MIX: the A row of MY row for the tile whose indices were switched in last: row position (vrow & 3) >= s of its quad -> the LAST row's
    auto va_of = [&](int sn) __attribute__((always_inline)) {
        const int sk = (vrow >> 2) ? (sn >> 4) : (sn & 15);
        return 2048 + ((vrow >> 2) * 2 + ((vrow & 3) >= sk ? 1 : 0)) * (KC * 4) + 16 * vq;
    };
    auto raw_load = [&](int rs, auto halfc, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        constexpr int hf = decltype(halfc)::value;
        const char* src = raww + rs * RAWS;
        b = *reinterpret_cast<const f32x4*>(src + vb + hf * (KC == 64 ? 1024 : 64));
        a = *reinterpret_cast<const f32x4*>(src + va + hf * (KC * 2));
    };
    auto conv_store = [&](int zs, auto halfc, const f32x4& a, const f32x4& b) __attribute__((always_inline)) {
        constexpr int hf = decltype(halfc)::value;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(a[i] + b[i], 0.f);
        // hi = fp16(v) truncated (one v_cvt_pkrtz per pair); lo = fp16(v - hi) rounded to nearest, one v_fma_mix per element
        // (f32 v * 1.0 - f16 hi, result rounded into one half of the destination): hi + lo == v to ~2^-22. v - hi is exact in
        // fp32, so this is bit-identical to the cvt / sub / cvt form the compiler emits for the plain C expression (5 VALU per
        // pair instead of 2 -- VALU issue beside the MFMA stream is what the conversion costs)
        // (operands travel as float bit patterns: the host pass also type-checks the constraints of this asm)
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
        char* rowp = zring + zs * ZSTAGE + (RPW * wave + vrow) * LDB + 16 * vq + 8 * hf;
        *reinterpret_cast<b32x2*>(rowp) = hw;
        *reinterpret_cast<b32x2*>(rowp + 2 * KC) = lw;
    };
    auto convert_half = [&](int rs, int zs, auto halfc) __attribute__((always_inline)) {
        f32x4 a, b;
        raw_load(rs, halfc, a, b);
        conv_store(zs, halfc, a, b);
    };

    // ---- MFMA side ----
    f32x16 acc[MT];
    struct Frag { f16x8 ah[2], al[2]; };
    const char* zfrag = zring + ((wm * MT) * 32 + l31) * LDB + 16 * hi;
    auto load_frag = [&](Frag& f, int zs, int g) __attribute__((always_inline)) {   // group g of a chunk = its 16-k step g
#ifdef WS_NO_FRAG
        asm volatile("" : "+v"(f.ah[0]), "+v"(f.al[0]), "+v"(f.ah[1]), "+v"(f.al[1]));
        return;
#endif
        const char* b = zfrag + zs * ZSTAGE + 32 * g;
        f.ah[0] = *reinterpret_cast<const f16x8*>(b);
        f.al[0] = *reinterpret_cast<const f16x8*>(b + 2 * KC);
        f.ah[1] = *reinterpret_cast<const f16x8*>(b + 32 * LDB);
        f.al[1] = *reinterpret_cast<const f16x8*>(b + 32 * LDB + 2 * KC);
    };
    auto mma = [&](const Frag& f, auto Sc, auto firstc) __attribute__((always_inline)) {
        constexpr int S = decltype(Sc)::value;            // k16 step of the tile (selects the W2 registers)
        constexpr bool first = decltype(firstc)::value != 0;
        f32x16 c0, c1;
        if constexpr (first) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        } else { c0 = acc[0]; c1 = acc[1]; }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[0], wh[S], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[1], wh[S], c1, 0, 0, 0);
#ifndef MORIG_2MFMA_EDGE      // measurement build: W2 rounded to fp16
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[0], wl[S], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[1], wl[S], c1, 0, 0, 0);
#endif
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[0], wh[S], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[1], wh[S], c1, 0, 0, 0);
        acc[0] = c0; acc[1] = c1;
    };
    // y = relu(sg x + b) * sc + sh is rising in the stored accumulator x (see the W2 slice above): the scan region receives the max of
    // a quad's four accumulators, the affine waits for the end of the segmented max
    auto write_z = [&](int slot) __attribute__((always_inline)) {
        // the max below reads the accumulators from inline assembly: ordered behind the MFMAs and given their wait states by
        // hand (see edge_pp.hip write_z_quad; DESIGN section 5, lesson 11)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("s_nop 15" : "+v"(acc[mt]));
        const int col = 32 * wn + l31;
        unsigned spw[2] = {0u, 0u};
        if constexpr (MIX) { spw[0] = ssp[slot * 2]; spw[1] = ssp[slot * 2 + 1]; }      // (broadcast reads: the 16 quads' split bytes)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a0 = acc[mt][4 * q], a1 = acc[mt][4 * q + 1], a2 = acc[mt][4 * q + 2], a3 = acc[mt][4 * q + 3];
                float* zt = Z + ((wm * MT + mt) * 8 + 2 * q + hi) * ZQ + col;
                bool pure = true;
                if constexpr (MIX) pure = ((__builtin_amdgcn_readfirstlane(spw[mt]) >> (8 * q)) & 0xFu) == 0xFu;    // both quads of the register pair (hi = 0, 1)
                if (pure || (p.dbg & 128)) {                  // (dbg 128: timing experiment, wrong results on mixed quads)
                    float hi4, t3;
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(a0), "v"(a1), "v"(a2));
                    asm("v_max_f32 %0, %1, %2" : "=v"(hi4) : "v"(t3), "v"(a3));
                    *zt = hi4;
                } else {
                    // rows 0 .. sm1 of the quad belong to the segment of its first row (T), the others to the next one (Hd)
                    const int sm1 = (int)((spw[mt] >> (8 * q + 2 * hi)) & 3u);
                    const float NI = -INFINITY;
                    const bool c1 = sm1 >= 1, c2 = sm1 >= 2, c3 = sm1 == 3;
                    const float t1 = c1 ? a1 : NI, t2 = c2 ? a2 : NI, t3 = c3 ? a3 : NI;
                    const float h1 = c1 ? NI : a1, h2 = c2 ? NI : a2, h3 = c3 ? NI : a3;
                    zt[0] = fmaxf(fmaxf(a0, t1), fmaxf(t2, t3));
                    zt[NQ * ZQ] = fmaxf(h1, fmaxf(h2, h3));
                }
            }
    };
    // ([r04] measured and not adopted here, profiles/r04r_*: segments dealt round-robin to the waves instead of by quad-row ownership,
    // and the four quads of a tile worked on side by side in write_z -- 2.18 vs 2.14 ms on the geo graph, 1.17 vs 1.12 ms on tpl: with a
    // partner wave on the SIMD the epilogue's latency chains are already covered, the extra bit walking is not)
    // segmented max over the NQ quad rows of a finished tile: a wave owns NQ/8 quad rows and every segment that STARTS
    // there; a lane holds VEC adjacent columns; two ballots list the segment starts (same scheme as edge_pp.hip)
    auto scan = [&](int js, int slot) __attribute__((always_inline)) {
        typedef float fvec __attribute__((ext_vector_type(VEC)));
        if (p.dbg & 1) return;
        const int t = tile_of(js);
        const bool run_first = (js & (R - 1)) == 0, run_last = (js & (R - 1)) == R - 1 || js == n_my - 1;
        const int rep = t / tpr;
        const int* sq = sq_all + slot * 32;
        // (scalars: the branches on them below are s_cbranch, not exec-mask regions -- as VGPR values the run bookkeeping cost 3 %)
        const bool first_cont = __builtin_amdgcn_readfirstlane(sflag[slot * 2]) != 0,
                   last_cont = __builtin_amdgcn_readfirstlane(sflag[slot * 2 + 1]) != 0;
        const int q0 = __builtin_amdgcn_readfirstlane(wave * (NQ / 8));
        const float* zl = Z + VEC * lane;
        float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + VEC * lane;
        const int ql = lane & (NQ - 1);
        const int sv = sq[ql];
        const int sp = sq[ql > 0 ? ql - 1 : 0];
        const unsigned START = (unsigned)__ballot(lane < NQ && (ql == 0 || sv != sp));
        const unsigned VALID = (unsigned)__ballot(lane < NQ && sv >= 0);
        unsigned mine = START & VALID & (((1u << (NQ / 8)) - 1u) << q0);
        // One segment [b, e) of quad rows. SPECIAL = the tile's first segment if it began above the tile, and its last one if it goes on
        // below: the only ones the run bookkeeping (carry in / carry out / shared row) concerns. They are taken OUT of the bit walk
        // below and handled after it by their owner wave, so the hot loop is the plain form: reduce, affine, store -- with the
        // bookkeeping's branches inside the loop the kernel lost 3 % (profiles/r05p_*), although none of them is taken there.
        auto segment = [&](int b, auto special_c) __attribute__((always_inline)) {
            constexpr bool SPECIAL = decltype(special_c)::value;
            const unsigned later = b < 31 ? (START & ~((2u << b) - 1u)) : 0u;
            const int e = later ? __builtin_ctz(later) : NQ;                 // the segment covers quad rows [b, e)
            const int sg = __builtin_amdgcn_readlane(sv, b);
            fvec m = *reinterpret_cast<const fvec*>(zl + b * ZQ);
            for (int q = b + 1; q < e; q += 2) {
                const fvec z0 = *reinterpret_cast<const fvec*>(zl + q * ZQ);
                const fvec z1 = *reinterpret_cast<const fvec*>(zl + min(q + 1, e - 1) * ZQ);
#pragma unroll
                for (int v = 0; v < VEC; ++v) m[v] = fmaxf(m[v], fmaxf(z0[v], z1[v]));
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* sbl = sbias + VEC * lane;        // ONE address register + immediate offsets for the panels AND the carry rows behind
            asm volatile("" : "+v"(sbl));                 // them (hoisted per-panel addresses were what spilled into the main loop)
            // (the fp32-rows form of the kernel -- a fallback since the units take split rows -- keeps the [r04] shape: ONE copy of this
            // code inside the bit walk, every tile boundary shared, launched with run = 1; three copies + the carry code spilled 11
            // registers into its main loop)
            bool shared = SPECIAL && ((b == 0 && first_cont) || (e == NQ && last_cont));
            if constexpr (SPECIAL && Y16) {
                // a first segment that began in my previous tile takes what that tile left in `carry`; a last segment that goes on into my
                // next tile is left there (max domain, before the affine) and not stored; rows over a RUN boundary are shared (atomics)
                shared = b == 0 && first_cont && run_first;
                if (b == 0 && first_cont && !run_first) {
                    const fvec cv = *reinterpret_cast<const fvec*>(sbl + 3 * H + ((js & 1) ^ 1) * H);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) m[v] = fmaxf(m[v], cv[v]);
                    shared = __builtin_amdgcn_readfirstlane(cshared[(js & 1) ^ 1]) != 0;
                }
                if (e == NQ && last_cont && !run_last) {
                    *reinterpret_cast<fvec*>(const_cast<float*>(sbl) + 3 * H + (js & 1) * H) = m;
                    if (lane == 0) cshared[js & 1] = shared ? 1 : 0;
                    return;
                }
                shared = shared || (e == NQ && last_cont);
            }
            {   // this lane's VEC columns: bias, scale, shift from the LDS panel, fetched one after the other behind the reduction (the
                // kernel sits at its register limit: three more live vectors spilled a lane constant into the main loop)
                const fvec cb = *reinterpret_cast<const fvec*>(sbl);
                const fvec cs = *reinterpret_cast<const fvec*>(sbl + H);
#pragma unroll
                for (int v = 0; v < VEC; ++v) m[v] = fmaxf((cs[v] < 0.f ? -m[v] : m[v]) + cb[v], 0.f);
                __builtin_amdgcn_sched_barrier(0);
                const fvec ct = *reinterpret_cast<const fvec*>(sbl + 2 * H);
#pragma unroll
                for (int v = 0; v < VEC; ++v) m[v] = m[v] * cs[v] + ct[v];
            }
            float* o = obase + (size_t)sg * p.ldy;
            if (SPECIAL && shared && !(p.dbg & 32)) {     // (dbg 32: timing experiment -- plain stores, wrong results on shared rows)
#pragma unroll
                for (int v = 0; v < VEC; ++v) atomic_max_f32(o + v, m[v]);
            } else if constexpr (Y16) {
                // my VEC adjacent columns never straddle a chunk: hi halves at (column % 32) * 2 inside the 128-byte chunk, lo halves 64 B on
                char* oc = reinterpret_cast<char*>(o - VEC * lane) + ((VEC * lane) >> 5) * 128 + ((VEC * lane) & 31) * 2;
                typedef float hvec __attribute__((ext_vector_type(VEC / 2)));
                hvec hv, lv;
#pragma unroll
                for (int v = 0; v < VEC; v += 2) {
                    float hb, lb;
                    split_pair_f16(m[v], m[v + 1], hb, lb);
                    if constexpr (VEC == 2) { hv = hb; lv = lb; } else { hv[v >> 1] = hb; lv[v >> 1] = lb; }
                    amax = fmaxf(amax, fmaxf(fabsf(m[v]), fabsf(m[v + 1])));
                }
                // (measured and not kept, profiles/r05r_*: lane pairs trading halves for ONE 16-byte store per lane, as edge_rl.hip does --
                // no change; at 252 of 256 registers this kernel's time moves by +-4 % with ANY edit of the epilogue: dropping the range
                // guard above made it 4 % slower)
                *reinterpret_cast<hvec*>(oc) = hv;
                *reinterpret_cast<hvec*>(oc + 64) = lv;
            } else {
                *reinterpret_cast<fvec*>(o) = m;
            }
        };
        if constexpr (!Y16) {
            while (mine) {                                                   // wave-uniform: SALU bit walking
                const int b = __builtin_ctz(mine);
                mine &= mine - 1u;
                segment(b, std::true_type{});
            }
        } else {
            const unsigned sv_all = START & VALID;
            const int lastb = sv_all ? 31 - __builtin_clz(sv_all) : 0;        // where the tile's last segment starts
            const bool own_first = first_cont && (mine & 1u) != 0;
            const bool own_last = last_cont && sv_all != 0 && ((mine >> lastb) & 1u) != 0 && !(own_first && lastb == 0);
            if (own_first) mine &= ~1u;
            if (own_last) mine &= ~(1u << lastb);
            while (mine) {                                                   // wave-uniform: SALU bit walking
                const int b = __builtin_ctz(mine);
                mine &= mine - 1u;
                segment(b, std::false_type{});
            }
            if (own_first) segment(0, std::true_type{});
            if (own_last) segment(lastb, std::true_type{});
        }
    };

    // MIX: the segmented max over quad rows that may hold two segments each (see the kernel's header). Per quad row q: id of its first /
    // last row (sq[q], sq[16 + q]), s - 1 of it (ssp). A segment STARTS in q when q is mixed (mid-quad start, from the Hd plane) or when
    // q's first row differs from q - 1's last row (aligned start, T plane); it ends with the T of the quad the next one starts in when
    // that start is mid-quad. The tile's head piece -- quad 0 mixed: T[0] alone is the tail of a segment that began above the tile --
    // is handled by quad 0's owner behind the walk, like the tile's first / last segment (carry in / carry out / shared row).
    auto scan_mix = [&](int js, int slot) __attribute__((always_inline)) {
        typedef float fvec __attribute__((ext_vector_type(VEC)));
        if (p.dbg & (1 | 64)) return;                                         // (dbg 64: timing experiment, no segmented max)
        const int t = tile_of(js);
        const bool run_first = (js & (R - 1)) == 0, run_last = (js & (R - 1)) == R - 1 || js == n_my - 1;
        const int rep = t / tpr;
        const int* sq = sq_all + slot * 32;
        const bool first_cont = __builtin_amdgcn_readfirstlane(sflag[slot * 2]) != 0,
                   last_cont = __builtin_amdgcn_readfirstlane(sflag[slot * 2 + 1]) != 0;
        const int q0 = __builtin_amdgcn_readfirstlane(wave * (NQ / 8));
        const float* zl = Z + VEC * lane;
        asm volatile("" : "+v"(zl));                                          // (rebuilt per scan: hoisted out of the tile loop it was spilled)
        float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + VEC * lane;
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));                                      // (the lane constants below are rebuilt per scan, not kept across the tile loop)
        const int ql = lane_v & (NQ - 1);
        const int sv = sq[ql], sl = sq[16 + ql];
        const int slp = sq[16 + (ql > 0 ? ql - 1 : 0)];
        const unsigned spw = ssp[slot * 2 + (ql >> 3)];
        const bool valid = sv >= 0;
        const bool midq = valid && ((spw >> (8 * ((ql >> 1) & 3) + 2 * (ql & 1))) & 3u) != 3u;
        const unsigned START = (unsigned)__ballot(lane_v < NQ && (ql == 0 || midq || sv != slp));
        const unsigned VALID = (unsigned)__ballot(lane_v < NQ && valid);
        const unsigned MID = (unsigned)__ballot(lane_v < NQ && midq);
        // the segments are DEALT to the waves in start order (k-th start -> wave k % 8), not owned by quad row: unaligned segments put up
        // to three starts into one wave's two quad rows (7-row segments: waves 0 and 7 took 2 + the head piece), and the chunk barrier
        // waits for the slowest wave
        const unsigned sv_all = START & VALID;
        unsigned mine = 0u;
        {
            unsigned bits = sv_all;
            int r = (8 - (q0 >> 1)) & 7;                                      // (q0 >> 1 = my wave) my turn whenever r % 8 == 0
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1u;
                if ((r & 7) == 0) mine |= 1u << b;
                ++r;
            }
        }
        // bias / scale / shift, then the store of one finished segment (max domain in, see the W2 slice): plain split rows, or atomics
        // (ONE LDS address register per use site, rebuilt behind an asm barrier, + immediate offsets for the panels and the carry rows: hoisted
        // per-row addresses spilled, and a spill's reload brings an s_waitcnt vmcnt(0) with it that drains the gather ring -- measured:
        // +10 % on the tpl graph with three spilled address registers)
        auto finish = [&](fvec m, int sg, bool shared, const float* sbl) __attribute__((always_inline)) {
            {
                const fvec cb = *reinterpret_cast<const fvec*>(sbl);
                const fvec cs = *reinterpret_cast<const fvec*>(sbl + H);
#pragma unroll
                for (int v = 0; v < VEC; ++v) m[v] = fmaxf((cs[v] < 0.f ? -m[v] : m[v]) + cb[v], 0.f);
                __builtin_amdgcn_sched_barrier(0);
                const fvec ct = *reinterpret_cast<const fvec*>(sbl + 2 * H);
#pragma unroll
                for (int v = 0; v < VEC; ++v) m[v] = m[v] * cs[v] + ct[v];
            }
            float* o = obase + (size_t)sg * p.ldy;
            if (shared) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) atomic_max_f32(o + v, m[v]);
            } else {
                char* oc = reinterpret_cast<char*>(o - VEC * lane) + ((VEC * lane) >> 5) * 128 + ((VEC * lane) & 31) * 2;
                typedef float hvec __attribute__((ext_vector_type(VEC / 2)));
                hvec hv, lv;
#pragma unroll
                for (int v = 0; v < VEC; v += 2) {
                    float hb, lb;
                    split_pair_f16(m[v], m[v + 1], hb, lb);
                    if constexpr (VEC == 2) { hv = hb; lv = lb; } else { hv[v >> 1] = hb; lv[v >> 1] = lb; }
                    amax = fmaxf(amax, fmaxf(fabsf(m[v]), fabsf(m[v + 1])));
                }
                *reinterpret_cast<hvec*>(oc) = hv;
                *reinterpret_cast<hvec*>(oc + 64) = lv;
            }
        };
        auto segment = [&](int b, auto special_c) __attribute__((always_inline)) {
            constexpr bool SPECIAL = decltype(special_c)::value;
            const unsigned later = START & ~((2u << b) - 1u);
            const int e = later ? __builtin_ctz(later) : NQ;                 // the next start (or the tile's end)
            const bool midb = ((MID >> b) & 1u) != 0, mide = e < NQ && ((MID >> e) & 1u) != 0;
            const int sg = midb ? __builtin_amdgcn_readlane(sl, b) : __builtin_amdgcn_readlane(sv, b);
            fvec m = *reinterpret_cast<const fvec*>(zl + (midb ? NQ * ZQ : 0) + b * ZQ);
            const int qe = mide ? e + 1 : e;                                  // T rows (b, qe) belong to it as well
            for (int q = b + 1; q < qe; q += 2) {
                const fvec z0 = *reinterpret_cast<const fvec*>(zl + q * ZQ);
                const fvec z1 = *reinterpret_cast<const fvec*>(zl + min(q + 1, qe - 1) * ZQ);
#pragma unroll
                for (int v = 0; v < VEC; ++v) m[v] = fmaxf(m[v], fmaxf(z0[v], z1[v]));
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* sbl = sbias + VEC * lane;
            asm volatile("" : "+v"(sbl));
            bool shared = false;
            if constexpr (SPECIAL) {
                const bool fc = b == 0 && first_cont && !midb;                // began above the tile (a mixed quad 0: that is the head piece)
                shared = fc && run_first;
                if (fc && !run_first) {
                    const fvec cv = *reinterpret_cast<const fvec*>(sbl + 3 * H + ((js & 1) ^ 1) * H);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) m[v] = fmaxf(m[v], cv[v]);
                    shared = __builtin_amdgcn_readfirstlane(cshared[(js & 1) ^ 1]) != 0;
                }
                if (e == NQ && last_cont && !run_last) {                      // goes on in my next tile: left in `carry`, not stored
                    *reinterpret_cast<fvec*>(const_cast<float*>(sbl) + 3 * H + (js & 1) * H) = m;
                    if (lane == 0) cshared[js & 1] = shared ? 1 : 0;
                    return;
                }
                shared = shared || (e == NQ && last_cont);
            }
            finish(m, sg, shared, sbl);
        };
        const int lastb = sv_all ? 31 - __builtin_clz(sv_all) : 0;            // where the tile's last segment starts
        const bool mid0 = (MID & 1u) != 0;
        const bool own_first = first_cont && !mid0 && (mine & 1u) != 0;
        const bool own_last = last_cont && sv_all != 0 && ((mine >> lastb) & 1u) != 0 && !(own_first && lastb == 0);
        // quad 0 is mixed: its T is the tail of the segment above the tile -- one more piece, dealt like one more start
        const bool own_head = mid0 && ((__builtin_popcount(sv_all) - (q0 >> 1)) & 7) == 0;
        if (own_first) mine &= ~1u;
        if (own_last) mine &= ~(1u << lastb);
        while (mine) {                                                       // wave-uniform: SALU bit walking
            const int b = __builtin_ctz(mine);
            mine &= mine - 1u;
            segment(b, std::false_type{});
        }
        if (own_first) segment(0, std::true_type{});
        if (own_last) segment(lastb, std::true_type{});
        if (own_head) {
            fvec m = *reinterpret_cast<const fvec*>(zl);
            const float* sbl = sbias + VEC * lane;
            asm volatile("" : "+v"(sbl));
            bool shared = first_cont && run_first;
            if (first_cont && !run_first) {
                const fvec cv = *reinterpret_cast<const fvec*>(sbl + 3 * H + ((js & 1) ^ 1) * H);
#pragma unroll
                for (int v = 0; v < VEC; ++v) m[v] = fmaxf(m[v], cv[v]);
                shared = __builtin_amdgcn_readfirstlane(cshared[(js & 1) ^ 1]) != 0;
            }
            finish(m, __builtin_amdgcn_readlane(sv, 0), shared, sbl);
        }
    };

    // ---- prologue: tile 0's chunks 0..2 in flight, chunk 0 converted ----
    load_indices(0);
    switch_tile(0);
    if constexpr (MIX) va = va_of(snext);
    load_indices(1);                                      // the gather switches tiles at chunk 1: indices are loaded a tile ahead
    dma(WC<0>{}, 0); dma(WC<1>{}, 1); dma(WC<2>{}, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    convert_half(0, 0, WC<0>{}); convert_half(0, 0, WC<1>{});
    int rs = 0;                                           // raw stage of the chunk whose MFMAs run next (g % 3)
    Frag F0, F1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // B_0: Z chunk 0, sbias, sq[0] visible

    // One chunk interval [B_c, B_c+1) of tile j (g = NC j + c): the MFMAs -- pending last group of chunk c-1, (c == 0:
    // accumulators -> scan region), groups 0 .. NG-2 of chunk c -- and the SIDE work: V(g+1), D(g+3), (c == 1: scan of tile
    // j-1). Measured (tools/gpu_edge_ablate.sh ablations): left to the compiler's order the side work simply ADDS to the MFMA time,
    // because an in-order wave that waits for an LDS round trip or a DMA issue slot issues no MFMA either, and its SIMD
    // partner runs the same code in phase. So the interval is cut into blocks of ONE MFMA group each (sched_barrier: nothing
    // crosses), and every block first issues the LDS reads whose data the NEXT block consumes, then does the VALU work on
    // data read one block earlier, then its 6 MFMAs: no wait ever follows its own issue directly.
    auto chunk = [&](auto cc, int j) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        constexpr int zs = c & 1;
        const int rs_v = rs == 2 ? 0 : rs + 1;
        f32x4 ra0, rb0, ra1, rb1;
        // MIX: from here on the conversions read the NEXT tile's chunks (its indices were switched in at chunk SWITCHC)
        if constexpr (MIX && c == 3) va = va_of(snext);
        // ---- block 0: pending group
        load_frag(F0, zs, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");           // D(g+1) landed; D(g+2) may be in flight
#ifndef WS_NO_CONVERT
        raw_load(rs_v, WC<0>{}, ra0, rb0);
#endif
        if (j > 0 || c > 0) {
            constexpr int cp = (c + NC - 1) % NC;
            mma(F1, WC<SPC * cp + SPC - 1>{}, WC<0>{});
#ifndef WS_NO_WRITEZ
            if constexpr (c == 0) { WS_TS(1); if (!(p.dbg & 1)) write_z((j + 2) % 3); WS_TS(2); }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- block 1: group 0
        load_frag(F1, zs, 1);
#ifndef WS_NO_CONVERT
        conv_store(zs ^ 1, WC<0>{}, ra0, rb0);
        raw_load(rs_v, WC<1>{}, ra1, rb1);           // after the first half's arithmetic: its registers are free again
#endif
        mma(F0, WC<SPC * c>{}, WC<(c == 0 ? 1 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- block 2: group 1 (NG == 4) / the rest of the side work
        if constexpr (NG == 4) load_frag(F0, zs, 2);
#ifndef WS_NO_CONVERT
        conv_store(zs ^ 1, WC<1>{}, ra1, rb1);
#endif
        if constexpr (NG == 4) {
            mma(F1, WC<SPC * c + 1>{}, WC<0>{});
            __builtin_amdgcn_sched_barrier(0);
            // ---- block 3: group 2
            load_frag(F1, zs, 3);
        }
        if constexpr (NG == 4) mma(F0, WC<SPC * c + 2>{}, WC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        // D() last: each LDS-DMA costs the issuing wave ~100 cycles of issue stall (measured: 3 per chunk = 14 % of the
        // kernel when they sat in front of an MFMA group); here the wave's MFMAs are all queued and the barrier is next
        if constexpr (c == SWITCHC) switch_tile((j + 1) % 3);
#ifndef WS_NO_DMA
        dma(WC<(c + 3) % NC>{}, rs);                       // raw(g) was converted during the previous interval: its stage is free
#endif
        if constexpr (c == 3) load_indices(j + 2);
        rs = rs_v;
        if constexpr (c == 1) { if (j > 0) { WS_TS(4); if constexpr (MIX) scan_mix(j - 1, (j - 1) % 3); else scan(j - 1, (j - 1) % 3); WS_TS(5); } }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef WS_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
    };
#pragma unroll 1
    for (int j = 0; j < n_my; ++j) {
        WS_TS(0);
        chunk(WC<0>{}, j);
        WS_TS(3);
        chunk(WC<1>{}, j);
        WS_TS(6);
        chunk(WC<2>{}, j);
        chunk(WC<3>{}, j);
        WS_TS(7);
    }
    // ---- drain: last group of the last tile, its epilogue ----
    mma(F1, WC<NS - 1>{}, WC<0>{});
    if (!(p.dbg & 1)) write_z((n_my - 1) % 3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             // no LDS-DMA may outlive the workgroup
    __builtin_amdgcn_s_barrier();
    if constexpr (MIX) scan_mix(n_my - 1, (n_my - 1) % 3); else scan(n_my - 1, (n_my - 1) % 3);
    if (!(amax < 65000.f)) *p.ovf = 1;
}

static int cu_count_of_current_device() {
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

int reserved_cus();

int launch_edge_ws(const EdgePcParams& p0, int nblocks, hipStream_t s) {
    EdgePcParams p = p0;
    static const int dbg = [] { const char* e = getenv("MORIG_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    if (!p.quad && !(p.min4 && p.H == 256 && p.y16)) return MORIG_E_UNSUPPORTED;
    // ([r04] a four-wave / 512-register form of this kernel -- a wave owns 64 output columns: W2 in 256 AGPRs, half the LDS fragment
    // traffic; edge_w4.hip in commit b3bd7ab, bit-identical to this one -- reached this kernel's main-loop time and lost 5 % overall: a
    // lone wave per SIMD has nobody to hide the quad epilogue, the scan and the per-tile bookkeeping behind (3 500 of a tile's 14 000
    // cycles, cycle-stamped), and a second accumulator set to overlap them does not fit beside W2. Traces and A/Bs: profiles/r04e..r04t,
    // DESIGN section 5 [r04])
    int ncu = cu_count_of_current_device();
    ncu = ncu > 8 ? (ncu / 8) * 8 : 8;
#ifdef MORIG_WS_TRACE
    static unsigned long long* trace_buf = [] { void* b = nullptr; return hipMalloc(&b, 64 * 8) == hipSuccess ? (unsigned long long*)b : nullptr; }();
    p.trace = trace_buf;
#endif
    int avail = ncu - ((reserved_cus() + 7) / 8) * 8;
    if (avail < 8) avail = 8;
    const int grid = nblocks < avail ? ((nblocks + 7) / 8) * 8 : avail;      // one persistent workgroup per CU, multiple of 8 (XCDs)
    if (p.H == 256 && p.y16 && !p.quad) hipLaunchKernelGGL((edge_ws_kernel<256, true, true>), dim3(grid), dim3(512), 0, s, p);
    else if (p.H == 256 && p.y16) hipLaunchKernelGGL((edge_ws_kernel<256, true>), dim3(grid), dim3(512), 0, s, p);
    else if (p.H == 256) hipLaunchKernelGGL((edge_ws_kernel<256>), dim3(grid), dim3(512), 0, s, p);
    else if (p.H == 128 && p.y16) hipLaunchKernelGGL((edge_ws_kernel<128, true>), dim3(grid), dim3(512), 0, s, p);
    else if (p.H == 128) hipLaunchKernelGGL((edge_ws_kernel<128>), dim3(grid), dim3(512), 0, s, p);
    else return MORIG_E_UNSUPPORTED;
    MORIG_LAUNCH_CHECK();
#ifdef MORIG_WS_TRACE
    {
        unsigned long long h[64];
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, p.trace, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "WS_TRACE H=%d quad=%d (per wave: write_z start, write_z, rest of chunk 0, chunk 1 to scan, scan, rest of chunk 1, chunks 2-3):\n", p.H, p.quad);
            for (int w = 0; w < 8; ++w) {
                fprintf(stderr, "  w%d", w);
                for (int q = 1; q < 8; ++q) fprintf(stderr, " %6lld", (long long)(h[q * 8 + w] - h[(q - 1) * 8 + w]));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    return MORIG_OK;
}

}  // namespace morig
