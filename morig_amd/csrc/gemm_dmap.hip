// PERSISTENT variant of the LDS-DMA split-fp16 GEMM (gemm_dma.hip), 256 x 256 tiles, 8 waves of 64 x 128.
//
// gemm_dma.hip launches one workgroup per tile: every tile pays a cold start (~2.5 us until its first K-chunk has landed in
// LDS and the first fragments are read: 3-7 % of a launch) with nothing to hide it, because the 128 KB ring and 512 threads at
// 256 registers leave room for ONE workgroup per CU. Here one workgroup per CU walks a static list of tiles (contiguous per
// XCD, as the EdgeConv kernels do) and treats the K-chunks of consecutive tiles as ONE stream through the 2-stage ring:
//   * chunk 0 of tile j+1 is fetched in the DMA slot of tile j's second-to-last chunk -- the stage it goes to was freed by
//     that chunk's hand-over barrier -- so it lands under tile j's last chunk and store epilogue;
//   * the store epilogue transposes through the stage of tile j's LAST chunk only (16-row slabs, 8 KB per wave = one stage),
//     never through the stage that is receiving the next tile's chunk 0;
//   * chunk 1 of tile j+1 is fetched when the epilogue is over (the usual prologue slot).
// vmcnt counts stores as well, so the wait for chunk 0' sits BEFORE the epilogue's stores (it was issued a chunk earlier).
// Everything else -- swizzled ring image, counted waits, DMA instructions spread between MFMA groups, epilogues -- is
// gemm_dma.hip's. Reference op: the vertex MLP layers, models/basic_modules.py:31-36.
#include "common.h"
#include "epilogue_store.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>

namespace morig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// The parameter block stays in the kernarg segment: the persistent loop reads its fields through a pointer the compiler cannot see
// through (re-made opaque at the top of every tile and of every epilogue), so they are s_load-ed from the scalar cache where
// they are used instead of being hoisted out of the tile loop -- hoisted, they cost 59 spilled SGPRs and, through the spill lanes,
// 26 spilled VGPRs in a kernel that needs 250 for accumulators and fragments.
struct GemmDmaHot { int M, K, ldx, ldw, tiles_n; const float* X; const float* W; const int* seg; };
typedef __attribute__((address_space(4))) const GemmDmaParams* kernarg_params_t;
// POOL = true : pooled launches (column max per mesh): accumulators in the usual layout (a lane owns a column: the max over the
//               rows of a wave tile is 32 register maxima + one half-wave exchange).
// POOL = false: store launches: the MFMAs run with the operands swapped (A = W fragment, B = X fragment), the result tile is D^T
//               (a lane owns an output ROW and groups of 4 adjacent columns) and leaves through the register epilogue
//               (epilogue_store.h: store_tile_regs): no LDS scratch, no barrier in front of the epilogue; bias (+ the row bias of
//               a one-mesh tile) = initial accumulator value, taken from a double-buffered LDS panel that is loaded for tile
//               j + 1 BEFORE tile j's stores are issued (vmcnt retires in order).
template <bool POOL>
__global__ __launch_bounds__(512) void gemm16_dmap_kernel(const GemmDmaParams) {
#if defined(__HIP_DEVICE_COMPILE__)              // (the host pass only needs the symbol: the body reads the kernarg segment directly)
    kernarg_params_t q = (kernarg_params_t)__builtin_amdgcn_kernarg_segment_ptr();   // the one by-value argument
    asm volatile("" : "+s"(q));
    const GemmDmaHot p = {q->M, q->K, q->ldx, q->ldw, q->tiles_n, q->X, q->W, q->seg};        // what the main loop reads
    constexpr int BM = 256, BN = 256, NT = 4, MT = 2;
    constexpr int WNW = BN / (32 * NT);          // 2 waves along N
    constexpr int NW = (BM / 64) * WNW;          // 8 waves
    constexpr int STAGE = (BM + BN) * 128;       // 64 KB: X tile + W tile of one 32-column chunk
    constexpr int XJ = BM / 8 / NW, WJ = BN / 8 / NW;     // 4 + 4 one-KiB DMA instructions per wave and chunk
    constexpr int PANEL = 3 * BN;                // floats: [bias (+ row bias) | scale | shift] of a tile's columns
    __shared__ __attribute__((aligned(128))) char smem[2 * STAGE + (POOL ? 0 : 2 * PANEL * 4)];
    float* pan = reinterpret_cast<float*>(smem + 2 * STAGE);          // POOL = false: two panels (tile parity)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WNW, wn = wave % WNW;

    // ---- this workgroup's tile list: XCD x (= blockIdx & 7 under round-robin dispatch) owns a contiguous range of linear tile
    // ids (row tile major, so the N-tiles of one row tile run at the same time on one XCD and share its L2) ----
    const int tiles_m = (p.M + BM - 1) / BM;
    const int T = tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int t_lo = (int)((long long)T * xcd / 8), t_hi = (int)((long long)T * (xcd + 1) / 8);
    const int n_my = (t_hi - t_lo - bi + nbx - 1) / nbx;
    if (n_my <= 0) return;                       // block-uniform

    const int nchunk = (p.K + 31) / 32;          // >= 2 (the launcher sends shallower products to gemm_dma.hip)
    const int rsub = lane >> 3, pslot = lane & 7;
    unsigned ow[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int r = (wave * WJ + j) * 8 + rsub;
        ow[j] = ((unsigned)r * (unsigned)p.ldw + 4u * (pslot ^ ((r >> 1) & 7))) * 4u;
    }
    // X source offsets of a tile whose first row is row0 (rows past M are clamped: they are never stored)
    auto x_offset = [&](int j, int row0) __attribute__((always_inline)) {
        const int r = (wave * XJ + j) * 8 + rsub;
        int xr = r; if (row0 + xr >= p.M) xr = p.M - 1 - row0;
        return ((unsigned)xr * (unsigned)p.ldx + 4u * (pslot ^ ((r >> 1) & 7))) * 4u;   // rule 21: swizzle the SOURCE
    };
    // Measurement builds (tools/gpu_gemm_ablate.sh; results wrong by construction): DMAP_ABL_HOT = every LDS-DMA re-reads chunk 0 of
    // tile 0 (cache-hot: what the instructions cost without the memory behind them), DMAP_ABL_NODMA = none issued in the steady
    // state, DMAP_ABL_NOWAIT = nobody waits for them to land, DMAP_ABL_NOFRAG = no fragment reads in the steady state
    auto dma_x = [&](const char* xb, unsigned o, int c, int stage, int j) __attribute__((always_inline)) {
#ifdef DMAP_ABL_HOT
        xb = reinterpret_cast<const char*>(p.X); c = 0;
#endif
        asm volatile("" : "+v"(o));              // keep (scalar base + lane offset) addressing
        __builtin_amdgcn_global_load_lds((glb_void_t*)(xb + c * 128 + o), (lds_void_t*)(smem + stage * STAGE + (wave * XJ + j) * 1024), 16, 0, 0);
    };
    auto dma_w = [&](const char* wb, int c, int stage, int j) __attribute__((always_inline)) {
#ifdef DMAP_ABL_HOT
        wb = reinterpret_cast<const char*>(p.W); c = 0;
#endif
        unsigned o = ow[j];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds((glb_void_t*)(wb + c * 128 + o), (lds_void_t*)(smem + stage * STAGE + BM * 128 + (wave * WJ + j) * 1024), 16, 0, 0);
    };

    const int x7 = (l31 >> 1) & 7;
    const int aoff = (wm * 64 + l31) * 128, boff = BM * 128 + (wn * NT * 32 + l31) * 128;
    struct Frag { f16x8 ah[MT], al[MT], bh[NT], bl[NT]; };
    auto load_frag = [&](Frag& f, int stage, int s2) __attribute__((always_inline)) {
        const char* st = smem + stage * STAGE;
        const int sh = ((2 * s2 + hi) ^ x7) * 16, sl = ((4 + 2 * s2 + hi) ^ x7) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f.ah[mt] = *reinterpret_cast<const f16x8*>(st + aoff + mt * 32 * 128 + sh);
            f.al[mt] = *reinterpret_cast<const f16x8*>(st + aoff + mt * 32 * 128 + sl);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f.bh[nt] = *reinterpret_cast<const f16x8*>(st + boff + nt * 32 * 128 + sh);
            f.bl[nt] = *reinterpret_cast<const f16x8*>(st + boff + nt * 32 * 128 + sl);
        }
    };
    f32x16 acc[MT][NT];
    // one split product term: x (a fragment of X) times w (a fragment of W); the store variant swaps the operand roles (D^T = W X^T)
    auto mm = [&](const f16x8& x, const f16x8& w, f32x16& c) __attribute__((always_inline)) {
        if constexpr (POOL) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, c, 0, 0, 0);
        else                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
    };
    auto mma = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                mm(f.al[mt], f.bh[nt], acc[mt][nt]);
                mm(f.ah[mt], f.bl[nt], acc[mt][nt]);
                mm(f.ah[mt], f.bh[nt], acc[mt][nt]);
            }
    };
    auto mma_pair = [&](const Frag& f, int mt, int nt0) __attribute__((always_inline)) {
        mm(f.al[mt], f.bh[nt0],     acc[mt][nt0]);
        mm(f.al[mt], f.bh[nt0 + 1], acc[mt][nt0 + 1]);
        mm(f.ah[mt], f.bl[nt0],     acc[mt][nt0]);
        mm(f.ah[mt], f.bl[nt0 + 1], acc[mt][nt0 + 1]);
        mm(f.ah[mt], f.bh[nt0],     acc[mt][nt0]);
        mm(f.ah[mt], f.bh[nt0 + 1], acc[mt][nt0 + 1]);
    };
    // ---- store variant: the column constants of one tile. Threads 0..255 fetch bias (+ the row bias when all rows of the tile lie
    // in one mesh), threads 256..511 scale and shift; `slow` (block-uniform) = the tile spans several meshes, the row bias is added
    // per row in the epilogue instead. Parameters come from the kernarg segment (scalar loads). ----
    float pv0 = 0.f, pv1 = 0.f;
    auto panel_fetch = [&](int plin, bool& slow) __attribute__((always_inline)) {
        asm volatile("" : "+s"(q));
        const float* bias = q->bias; const float* scale = q->scale; const float* shift = q->shift;
        const float* rowbias = q->rowbias; const int* seg = q->seg;
        const int col = (plin % p.tiles_n) * BN + (tid & (BN - 1));
        slow = false;
        pv0 = 0.f; pv1 = 0.f;
        if (tid < BN) {
            if (bias) pv0 = bias[col];
        } else if (scale) { pv0 = scale[col]; pv1 = shift[col]; }
        if (rowbias != nullptr) {
            const int prow0 = (plin / p.tiles_n) * BM;
            const int s0 = seg[prow0], s1 = seg[min(prow0 + BM, p.M) - 1];       // `seg` is sorted: one mesh iff the ends agree
            slow = s0 != s1;
            if (!slow && tid < BN) pv1 = rowbias[(size_t)s0 * q->ld_rowbias + col];
        }
    };
    auto panel_write = [&](int par) __attribute__((always_inline)) {
        float* pn = pan + par * PANEL;
        if (tid < BN) pn[tid] = pv0 + pv1;
        else { pn[BN + (tid - BN)] = pv0; pn[2 * BN + (tid - BN)] = pv1; }
    };

    // one K-chunk of the stream. `more`: a chunk of this tile follows (hand-over barrier in the middle); `go`: this chunk's DMA slot
    // is used -- for stream chunk g + c + 2, which goes to the stage chunk c is leaving -- with the given bases / chunk index
    unsigned ox[XJ];
    Frag f0, f1;
    auto chunk_iter = [&](int st, auto morec, auto goc, const char* xb, const char* wb, int cc, bool reclamp, int nrow0)
                          __attribute__((always_inline)) {
#ifdef DMAP_ABL_NODMA
        constexpr bool more = decltype(morec)::value != 0, go = false;
#else
        constexpr bool more = decltype(morec)::value != 0, go = decltype(goc)::value != 0;
#endif
#ifdef DMAP_ABL_NOFRAG
        asm volatile("" : "+v"(f1.ah[0]), "+v"(f1.al[0]), "+v"(f1.ah[1]), "+v"(f1.al[1]));
        asm volatile("" : "+v"(f1.bh[0]), "+v"(f1.bl[0]), "+v"(f1.bh[1]), "+v"(f1.bl[1]), "+v"(f1.bh[2]), "+v"(f1.bl[2]), "+v"(f1.bh[3]), "+v"(f1.bl[3]));
#else
        load_frag(f1, st, 1);
#endif
        mma(f0);
        if constexpr (more) {
#ifndef DMAP_ABL_NOWAIT
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunk c+1 landed for this wave (the only DMA in flight)
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my own reads of chunk c have returned ...
            __builtin_amdgcn_s_barrier();                          // ... and everyone's: stage `st` is free
#ifdef DMAP_ABL_NOFRAG
            asm volatile("" : "+v"(f0.ah[0]), "+v"(f0.al[0]), "+v"(f0.ah[1]), "+v"(f0.al[1]));
            asm volatile("" : "+v"(f0.bh[0]), "+v"(f0.bl[0]), "+v"(f0.bh[1]), "+v"(f0.bl[1]), "+v"(f0.bh[2]), "+v"(f0.bl[2]), "+v"(f0.bh[3]), "+v"(f0.bl[3]));
#else
            load_frag(f0, st ^ 1, 0);
#endif
        }
        unsigned o0 = ox[0], o1 = ox[1], o2 = ox[2], o3 = ox[3];
        if (go && reclamp) { o0 = x_offset(0, nrow0); o1 = x_offset(1, nrow0); o2 = x_offset(2, nrow0); o3 = x_offset(3, nrow0); }
        mma_pair(f1, 0, 0); __builtin_amdgcn_sched_barrier(0);
        if constexpr (go) { dma_x(xb, o0, cc, st, 0); dma_x(xb, o1, cc, st, 1); }
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(f1, 0, 2); __builtin_amdgcn_sched_barrier(0);
        if constexpr (go) { dma_x(xb, o2, cc, st, 2); dma_x(xb, o3, cc, st, 3); }
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(f1, 1, 0); __builtin_amdgcn_sched_barrier(0);
        if constexpr (go) { dma_w(wb, cc, st, 0); dma_w(wb, cc, st, 1); }
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(f1, 1, 2); __builtin_amdgcn_sched_barrier(0);
        if constexpr (go) { dma_w(wb, cc, st, 2); dma_w(wb, cc, st, 3); }
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 1> Yes;
    typedef std::integral_constant<int, 0> No;

    // ---- first tile: chunk 0 in flight ----
    int lin = t_lo + bi;
    int g = 0;                                   // stream position of the current tile's chunk 0: chunk c lives in stage (g + c) & 1
    {
        const int row0 = (lin / p.tiles_n) * BM;
        const char* xbase = reinterpret_cast<const char*>(p.X + (size_t)row0 * p.ldx);
        const char* wbase = reinterpret_cast<const char*>(p.W + (size_t)((lin % p.tiles_n) * BN) * p.ldw);
#pragma unroll
        for (int j = 0; j < XJ; ++j) dma_x(xbase, x_offset(j, row0), 0, 0, j);
#pragma unroll
        for (int j = 0; j < WJ; ++j) dma_w(wbase, 0, 0, j);
    }
    bool slow_cur = false, slow_next = false;    // POOL = false: does the current / next tile span several meshes (row bias per row)?
    if constexpr (!POOL) { panel_fetch(lin, slow_cur); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!POOL) { panel_write(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

#pragma unroll 1
    for (int jt = 0; jt < n_my; ++jt) {
        const int tn = lin % p.tiles_n, row0 = (lin / p.tiles_n) * BM;
        const char* xbase = reinterpret_cast<const char*>(p.X + (size_t)row0 * p.ldx);
        const char* wbase = reinterpret_cast<const char*>(p.W + (size_t)(tn * BN) * p.ldw);
#pragma unroll
        for (int j = 0; j < XJ; ++j) ox[j] = x_offset(j, row0);

        __builtin_amdgcn_s_barrier();            // chunk 0 landed for everyone; this tile's panel is visible
#pragma unroll
        for (int j = 0; j < XJ; ++j) dma_x(xbase, ox[j], 1, (g + 1) & 1, j);          // chunk 1: the usual prologue slot
#pragma unroll
        for (int j = 0; j < WJ; ++j) dma_w(wbase, 1, (g + 1) & 1, j);
        if constexpr (POOL) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        } else {
            // accumulators start at bias (+ row bias): register r of tile nt is column 32 nt + (r & 3) + 8 (r >> 2) + 4 hi
            typedef float pf32x4 __attribute__((ext_vector_type(4)));
            const float* pn = pan + (jt & 1) * PANEL + wn * NT * 32 + 4 * hi;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const pf32x4 b4 = *reinterpret_cast<const pf32x4*>(pn + nt * 32 + 8 * g4);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) acc[mt][nt][4 * g4 + qq] = b4[qq];
                }
        }
        load_frag(f0, g & 1, 0);

#pragma unroll 1
        for (int c = 0; c + 2 < nchunk; ++c) chunk_iter((g + c) & 1, Yes{}, Yes{}, xbase, wbase, c + 2, false, 0);
        {
            // the second-to-last chunk's slot fetches chunk 0 of the NEXT tile of the list (at the end of the list this tile's
            // own chunk 0 once more: never read); rows past M are clamped
            const int nlin = jt + 1 < n_my ? lin + nbx : lin;
            const int nrow0 = (nlin / p.tiles_n) * BM;
            const char* nxbase = reinterpret_cast<const char*>(p.X + (size_t)nrow0 * p.ldx);
            const char* nwbase = reinterpret_cast<const char*>(p.W + (size_t)((nlin % p.tiles_n) * BN) * p.ldw);
            // the next tile's column constants: requested two chunks ahead of the epilogue's stores (vmcnt retires in order, so
            // a load issued behind the stores could only be waited for together with them)
            if constexpr (!POOL) panel_fetch(nlin, slow_next);
            chunk_iter((g + nchunk - 2) & 1, Yes{}, Yes{}, nxbase, nwbase, 0, true, nrow0);
            chunk_iter((g + nchunk - 1) & 1, No{}, No{}, nxbase, nwbase, 0, false, 0);
            lin = nlin;
        }

        // ---- tile done: chunk 0 of the next tile must have landed BEFORE the epilogue issues stores (vmcnt is in order) ----
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const int colw0 = tn * BN + wn * NT * 32;
        if constexpr (POOL) {
            asm volatile("" : "+s"(q));
            GemmDmaParams pe;                    // field by field: scalar loads from the kernarg segment (a memcpy becomes VMEM loads)
            pe.M = q->M; pe.N = q->N; pe.bias = q->bias; pe.scale = q->scale; pe.shift = q->shift; pe.relu = q->relu; pe.seg = q->seg;
            pe.pool = q->pool; pe.ld_pool = q->ld_pool;
            // pooled epilogue (scatter_max over meshes), as gemm_dma.hip
            const int rfirst = row0 + wm * 64;
            if (rfirst < pe.M) {
                const int rlast = min(rfirst + 63, pe.M - 1);
                const int s0 = pe.seg[rfirst];
                const bool uni = s0 == pe.seg[rlast];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int col = colw0 + nt * 32 + l31;
                    const bool cok = col < pe.N;
                    const float b = (pe.bias && cok) ? pe.bias[col] : 0.f;
                    const float sc = (pe.scale && cok) ? pe.scale[col] : 1.f;
                    const float shf = (pe.shift && cok) ? pe.shift[col] : 0.f;
                    float m = -INFINITY;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = rfirst + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                            float v = acc[mt][nt][r] + b;
                            if (pe.relu) v = v > 0.f ? v : 0.f;
                            v = v * sc + shf;
                            if (row < pe.M) {
                                if (uni) m = fmaxf(m, v);
                                else if (cok) atomic_max_f32(pe.pool + (size_t)pe.seg[row] * pe.ld_pool + col, v);
                            }
                        }
                    if (uni) {
                        m = fmaxf(m, __shfl_xor(m, 32, 64));
                        if (hi == 0 && cok && m > -INFINITY) atomic_max_f32(pe.pool + (size_t)s0 * pe.ld_pool + col, m);
                    }
                }
            }
        } else {
            panel_write((jt + 1) & 1);           // the next tile's panel (its loads returned with the vmcnt(0) above); published by the
                                                 // barrier at the top of the next tile. This tile's panel: parity jt & 1
            asm volatile("" : "+s"(q));
            GemmDmaParams pe;
            pe.M = q->M; pe.N = q->N; pe.scale = q->scale; pe.relu = q->relu;
            pe.rowbias = q->rowbias; pe.ld_rowbias = q->ld_rowbias; pe.seg = q->seg;
            pe.Y = q->Y; pe.ldy = q->ldy; pe.y16 = q->y16; pe.ovf = q->ovf; pe.dbg = q->dbg;
            const float* pn = pan + (jt & 1) * PANEL;
            if (!(pe.dbg & 1))
                store_tile_regs<MT, NT>(pe, acc, pe.scale ? pn + BN : nullptr, pn + 2 * BN, wm * 64, row0, pe.M, colw0, wn * NT * 32, lane,
                                        slow_cur);
            slow_cur = slow_next;
        }
        g += nchunk;                             // the next tile's chunk 0 sits in stage (g + nchunk) & 1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
#endif
}

static int cu_count() {
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

int launch_gemm16_dmap(const GemmDmaParams& p0, hipStream_t s) {
    GemmDmaParams p = p0;
    static const int dbg = [] { const char* e = getenv("MORIG_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    p.tiles_n = p.N / 256;
    const int T = cdiv(p.M, 256) * p.tiles_n;
    int ncu = cu_count();
    ncu = ncu > 8 ? (ncu / 8) * 8 : 8;
    int avail = ncu - ((reserved_cus() + 7) / 8) * 8;
    if (avail < 8) avail = 8;
    const int grid = T < avail ? ((T + 7) / 8) * 8 : avail;
    if (p.pool) hipLaunchKernelGGL(gemm16_dmap_kernel<true>, dim3(grid), dim3(512), 0, s, p);
    else        hipLaunchKernelGGL(gemm16_dmap_kernel<false>, dim3(grid), dim3(512), 0, s, p);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig
