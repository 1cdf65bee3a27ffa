// PING-PONG variant of the persistent LDS-DMA split-fp16 GEMM (gemm_dmap.hip), 256 x 256 tiles, 8 waves of 64 x 128.
//
// gemm_dmap.hip runs its eight waves through the same code in phase: every wave issues 24 MFMAs, then its fragment reads and
// its LDS-DMA pieces, and the two waves that share a SIMD stall on those side instructions at the same time (counter-measured
// MFMA utilisation 0.62-0.68, 37 % of the wave cycles issue-stalled). Here the two waves of a SIMD work in OPPOSITE roles
// (MI355X_MICROARCH.md, "Two waves per SIMD": matrix beside memory is the complementary pairing): waves 0-3 (group A, one per
// SIMD) and waves 4-7 (group B) alternate between
//   compute(c): the 48 MFMAs of K-chunk c (2 k16 steps x 2 x 4 tiles x 3 split terms), nothing else in the stream;
//   load(c')  : the 24 ds_read_b128 of the NEXT chunk's fragments, this wave's share of the LDS-DMA pieces, the waits;
// one s_barrier between the slots, so a SIMD's matrix pipe always has exactly one wave feeding it while the other wave's LDS
// and VMEM instructions issue into the gaps of that stream instead of in front of its own MFMAs:
//        slot 2c    : A compute(c)               |  B load(c)   + B's pieces of chunk c+1
//        slot 2c+1  : A load(c+1) + A's pieces of chunk c+2  |  B compute(c)
// Ring: the same 2 x 64 KB stages (stage = chunk parity). Chunk c is read by A in slot 2c-1 and by B in slot 2c, so its stage is
// free from slot 2c+1 on; chunk c+2 goes there in slots 2c+1 (A's share) and 2c+2 (B's share) and is first read in slot 2c+3:
// B's pieces have to land inside their own slot, so B issues only a quarter of them (16 of 64) and issues them first.
// A waits for its own pieces at the end of its compute slot (issued a slot earlier: landed long ago).
// The chunk stream runs across tile boundaries as in gemm_dmap.hip; a wave's epilogue is appended to the load slot that
// follows its last compute slot (A: under B's last compute; B: under A's first compute of the next tile).
// Everything else -- swizzled ring image, split layout, register epilogue of the store launches, pooled epilogue -- is
// gemm_dmap.hip's. Reference op: the vertex MLP layers, models/basic_modules.py:31-36, models/rignet.py:56-57.
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

#ifndef PP_AX
#define PP_AX 6
#endif
struct GemmPpHot { int M, K, ldx, ldw, tiles_n; const float* X; const float* W; const int* seg; };
typedef __attribute__((address_space(4))) const GemmDmaParams* pp_kernarg_t;

template <bool POOL>
__global__ __launch_bounds__(512) void gemm16_pp_kernel(const GemmDmaParams) {
#if defined(__HIP_DEVICE_COMPILE__)
    pp_kernarg_t q = (pp_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    const GemmPpHot p = {q->M, q->K, q->ldx, q->ldw, q->tiles_n, q->X, q->W, q->seg};
    constexpr int BM = 256, BN = 256, NT = 4, MT = 2;
    constexpr int WNW = BN / (32 * NT);          // 2 waves along N
    constexpr int STAGE = (BM + BN) * 128;       // 64 KB: X tile + W tile of one 32-column chunk
    constexpr int PANEL = 3 * BN;                // floats: [bias (+ row bias) | scale | shift] of a tile's columns
    constexpr int AX = PP_AX, BX = 8 - AX;           // one-KiB pieces of X (and as many of W) per wave and chunk: group A / group B
    __shared__ __attribute__((aligned(128))) char smem[2 * STAGE + (POOL ? 0 : 2 * PANEL * 4)];
    float* pan = reinterpret_cast<float*>(smem + 2 * STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: the role branches below are s_cbranch, not exec masks
    const bool grpB = wave >= 4;
    const int wg = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WNW, wn = wave % WNW;

    const int tiles_m = (p.M + BM - 1) / BM;
    const int T = tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int t_lo = (int)((long long)T * xcd / 8), t_hi = (int)((long long)T * (xcd + 1) / 8);
    const int n_my = (t_hi - t_lo - bi + nbx - 1) / nbx;
    if (n_my <= 0) return;                       // block-uniform

    const int nchunk = (p.K + 31) / 32;          // >= 2 (launcher)
    const int rsub = lane >> 3, pslot = lane & 7;
    const int id0 = grpB ? 4 * AX + wg * BX : wg * AX;               // this wave's first piece id (a piece = 8 rows x 128 B)
    const int npiece = grpB ? BX : AX;

    // piece `id` (8 rows x 128 B) of chunk cc of the X tile whose first row is row0 / of the W tile at wb -> ring stage `stage`.
    // A lane's source offset is (id * 8 + rsub) * ld * 4 + 16 * (pslot ^ swz), swz = ((id * 8 + rsub) >> 1) & 7 = (4 (id & 1) + (rsub >> 1)) & 7
    // (rule 21: the swizzle is applied to the SOURCE): the id-dependent row part is wave-uniform and goes into the scalar base,
    // the lane part exists in two forms (id even / odd) -- 2 + 2 VGPRs for all pieces instead of one offset per piece (hoisted out
    // of the chunk loop, a dozen of those spilled the accumulators). Rows past M (last row tile only): per-lane clamp, slow form.
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const unsigned vx0 = ((unsigned)rsub * (unsigned)p.ldx + 4u * (pslot ^ ((rsub >> 1) & 7))) * 4u;
    const unsigned vx1 = ((unsigned)rsub * (unsigned)p.ldx + 4u * (pslot ^ ((4 + (rsub >> 1)) & 7))) * 4u;
    const unsigned vw0 = ((unsigned)rsub * (unsigned)p.ldw + 4u * (pslot ^ ((rsub >> 1) & 7))) * 4u;
    const unsigned vw1 = ((unsigned)rsub * (unsigned)p.ldw + 4u * (pslot ^ ((4 + (rsub >> 1)) & 7))) * 4u;
    auto dma_x = [&](const char* xb, int row0, int cc, int stage, int id, int par) __attribute__((always_inline)) {
        unsigned o;
        const char* sb;
        if (row0 + BM <= p.M) {                  // block-uniform
            o = par ? vx1 : vx0;
            sb = xb + cc * 128 + (size_t)(id * 8) * p.ldx * 4;
        } else {
            const int r = id * 8 + rsub;
            int xr = r; if (row0 + xr >= p.M) xr = p.M - 1 - row0;       // rows past M are clamped: never stored
            o = ((unsigned)xr * (unsigned)p.ldx + 4u * (pslot ^ ((r >> 1) & 7))) * 4u;
            sb = xb + cc * 128;
        }
        // scalar base (SGPR pair) + 32-bit lane offset, LDS destination through M0: the builtin only selects the 64-bit-VGPR address
        // form (an address pair and two v_lshl_add_u64 per piece)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(smem_lds + (unsigned)(stage * STAGE + id * 1024)), "v"(o), "s"(sb) : "memory");
    };
    auto dma_w = [&](const char* wb, int cc, int stage, int id, int par) __attribute__((always_inline)) {
        const unsigned o = par ? vw1 : vw0;
        const char* sb = wb + cc * 128 + (size_t)(id * 8) * p.ldw * 4;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(smem_lds + (unsigned)(stage * STAGE + BM * 128 + id * 1024)), "v"(o), "s"(sb) : "memory");
    };
    static_assert(AX % 2 == 0 && BX % 2 == 0, "a wave's first piece id is even: piece parity = j & 1");
    auto dma_share = [&](const char* xb, const char* wb, int row0, int cc, int stage) __attribute__((always_inline)) {
        if (grpB) {
#pragma unroll
            for (int j = 0; j < BX; ++j) { dma_x(xb, row0, cc, stage, id0 + j, j & 1); dma_w(wb, cc, stage, id0 + j, j & 1); }
        } else {
#pragma unroll
            for (int j = 0; j < AX; ++j) { dma_x(xb, row0, cc, stage, id0 + j, j & 1); dma_w(wb, cc, stage, id0 + j, j & 1); }
        }
    };
    (void)npiece;

    const int x7 = (l31 >> 1) & 7;
    const int aoff = (wm * 64 + l31) * 128, boff = BM * 128 + (wn * NT * 32 + l31) * 128;
    struct Frag { f16x8 ah[MT], al[MT], bh[NT], bl[NT]; };
    Frag fr[2];
    auto load_frags = [&](int stage) __attribute__((always_inline)) {
        const char* st = smem + stage * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int sh = ((2 * s2 + hi) ^ x7) * 16, sl = ((4 + 2 * s2 + hi) ^ x7) * 16;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                fr[s2].ah[mt] = *reinterpret_cast<const f16x8*>(st + aoff + mt * 32 * 128 + sh);
                fr[s2].al[mt] = *reinterpret_cast<const f16x8*>(st + aoff + mt * 32 * 128 + sl);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                fr[s2].bh[nt] = *reinterpret_cast<const f16x8*>(st + boff + nt * 32 * 128 + sh);
                fr[s2].bl[nt] = *reinterpret_cast<const f16x8*>(st + boff + nt * 32 * 128 + sl);
            }
        }
    };
    f32x16 acc[MT][NT];
    auto mm = [&](const f16x8& x, const f16x8& w, f32x16& c) __attribute__((always_inline)) {
        if constexpr (POOL) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, c, 0, 0, 0);
        else                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
    };
    // A compute slot is ONE wave feeding the SIMD's matrix pipe: consecutive MFMAs must not wait for each other, so each split
    // term sweeps all eight accumulators of the wave tile before the next term touches them again (dependent MFMAs 8 apart; with
    // the in-phase kernel's 2-apart order a lone wave left the pipe idle between them: first A/B, profiles/r04b_*). Per accumulator
    // the order of its additions is unchanged (k16 step, then lo*hi, hi*lo, hi*hi): results stay bit-identical to gemm_dmap.hip.
    auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const Frag& f = fr[s2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mm(f.al[mt], f.bh[nt], acc[mt][nt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mm(f.ah[mt], f.bl[nt], acc[mt][nt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mm(f.ah[mt], f.bh[nt], acc[mt][nt]);
        }
    };
    // ---- store variant: the column constants of one tile (as gemm_dmap.hip) ----
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
            const int s0 = seg[prow0], s1 = seg[min(prow0 + BM, p.M) - 1];
            slow = s0 != s1;
            if (!slow && tid < BN) pv1 = rowbias[(size_t)s0 * q->ld_rowbias + col];
        }
    };
    auto panel_write = [&](int par) __attribute__((always_inline)) {
        float* pn = pan + par * PANEL;
        if (tid < BN) pn[tid] = pv0 + pv1;
        else { pn[BN + (tid - BN)] = pv0; pn[2 * BN + (tid - BN)] = pv1; }
    };
    auto acc_init = [&](int par) __attribute__((always_inline)) {
        if constexpr (POOL) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        } else {
            typedef float pf32x4 __attribute__((ext_vector_type(4)));
            const float* pn = pan + par * PANEL + wn * NT * 32 + 4 * hi;
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
    };
    // the epilogue of the tile with linear id `elin` whose panel has parity `par`
    auto epilogue = [&](int elin, int par, bool slow) __attribute__((always_inline)) {
        const int tn = elin % p.tiles_n, row0 = (elin / p.tiles_n) * BM;
        const int colw0 = tn * BN + wn * NT * 32;
        asm volatile("" : "+s"(q));
        if constexpr (POOL) {
            GemmDmaParams pe;
            pe.M = q->M; pe.N = q->N; pe.bias = q->bias; pe.scale = q->scale; pe.shift = q->shift; pe.relu = q->relu; pe.seg = q->seg;
            pe.pool = q->pool; pe.ld_pool = q->ld_pool; pe.dbg = q->dbg;
            if (pe.dbg & 1) return;
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
            GemmDmaParams pe;
            pe.M = q->M; pe.N = q->N; pe.scale = q->scale; pe.relu = q->relu;
            pe.rowbias = q->rowbias; pe.ld_rowbias = q->ld_rowbias; pe.seg = q->seg;
            pe.Y = q->Y; pe.ldy = q->ldy; pe.y16 = q->y16; pe.ovf = q->ovf; pe.dbg = q->dbg;
            const float* pn = pan + par * PANEL;
            if (!(pe.dbg & 1))
                store_tile_regs<MT, NT>(pe, acc, pe.scale ? pn + BN : nullptr, pn + 2 * BN, wm * 64, row0, pe.M, colw0, wn * NT * 32, lane, slow);
        }
    };
    auto xbase_of = [&](int tl) __attribute__((always_inline)) {
        return reinterpret_cast<const char*>(p.X + (size_t)((tl / p.tiles_n) * BM) * p.ldx);
    };
    auto wbase_of = [&](int tl) __attribute__((always_inline)) {
        return reinterpret_cast<const char*>(p.W + (size_t)((tl % p.tiles_n) * BN) * p.ldw);
    };

    // ---- prologue: chunk 0 of the first tile in flight (every wave its share), the first panel ----
    int lin = t_lo + bi;
    bool slow_cur = false, slow_next = false, slow_prev = false;
    {
        const int row0 = (lin / p.tiles_n) * BM;
        dma_share(xbase_of(lin), wbase_of(lin), row0, 0, 0);
        if constexpr (!POOL) panel_fetch(lin, slow_cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!POOL) { panel_write(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
    }

    // Both groups run the SAME loop -- per stream chunk k a load slot L(k) = [epilogue of the previous tile and this tile's initial
    // accumulators if k opens a tile; my pieces of chunk k+1; the fragments of chunk k] and a compute slot C(k) -- one slot apart in
    // time: group A does not wait at the barrier that ends its first load slot, so A's C(k) runs beside B's L(k) and A's L(k+1)
    // beside B's C(k). (Fragments are loaded and consumed inside one iteration: nothing but the accumulators is loop-carried.)
    int g = 0;                                   // stream position of the current tile's chunk 0: chunk c lives in stage (g + c) & 1
    int prev_lin = lin;
    bool a_skips = !grpB;                        // group A's one skipped barrier
#pragma unroll 1
    for (int jt = 0; jt < n_my; ++jt) {
        const bool has_next = jt + 1 < n_my;
        const int nlin = has_next ? lin + nbx : lin;
        const char* xb = xbase_of(lin); const char* wb = wbase_of(lin);
        const int row0 = (lin / p.tiles_n) * BM;
        const int cP = nchunk - 2 > 1 ? nchunk - 2 : 1;      // the chunk in whose load slot the NEXT tile's panel is fetched and written
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            const int st = (g + c) & 1;
            const bool panel_now = !POOL && c == cP && has_next;
            // ---------------- load slot L(c) ----------------
#ifdef PP_PRIO
            __builtin_amdgcn_s_setprio(PP_PRIO);             // the loading wave's LDS / VMEM issue ahead of the partner's MFMA stream
#endif
            if (panel_now) panel_fetch(nlin, slow_next);
            if (grpB) {                          // B's pieces must land inside this slot: out first
                if (c + 1 < nchunk) dma_share(xb, wb, row0, c + 1, st ^ 1);
                else if (has_next) dma_share(xbase_of(nlin), wbase_of(nlin), (nlin / p.tiles_n) * BM, 0, st ^ 1);
            }
            if (c == 0) {
                if (jt > 0) epilogue(prev_lin, (jt - 1) & 1, slow_prev);
                acc_init(jt & 1);
            }
            load_frags(st);
            if (!grpB) {
                if (c + 1 < nchunk) dma_share(xb, wb, row0, c + 1, st ^ 1);
                else if (has_next) dma_share(xbase_of(nlin), wbase_of(nlin), (nlin / p.tiles_n) * BM, 0, st ^ 1);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (panel_now && grpB) panel_write((jt + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (!a_skips) __builtin_amdgcn_s_barrier();
            a_skips = false;
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- compute slot C(c) ----------------
#ifdef PP_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            compute();
            __builtin_amdgcn_sched_barrier(0);
            if (!grpB) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my pieces of chunk c+1 (issued in L(c)) and the panel loads
                if (panel_now) { panel_write((jt + 1) & 1); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        g += nchunk;
        prev_lin = lin; slow_prev = slow_cur;
        lin = nlin; slow_cur = slow_next;
    }
    epilogue(prev_lin, (n_my - 1) & 1, slow_prev);                   // the last tile
    if (!grpB) __builtin_amdgcn_s_barrier();                         // (the barrier group A skipped)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // no LDS-DMA may outlive the workgroup
#endif
}

static int pp_cu_count() {
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

int launch_gemm16_pp(const GemmDmaParams& p0, hipStream_t s) {
    GemmDmaParams p = p0;
    static const int dbg = [] { const char* e = getenv("MORIG_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    p.tiles_n = p.N / 256;
    const int T = cdiv(p.M, 256) * p.tiles_n;
    int ncu = pp_cu_count();
    ncu = ncu > 8 ? (ncu / 8) * 8 : 8;
    int avail = ncu - ((reserved_cus() + 7) / 8) * 8;
    if (avail < 8) avail = 8;
    const int grid = T < avail ? ((T + 7) / 8) * 8 : avail;
    if (p.pool) hipLaunchKernelGGL(gemm16_pp_kernel<true>, dim3(grid), dim3(512), 0, s, p);
    else        hipLaunchKernelGGL(gemm16_pp_kernel<false>, dim3(grid), dim3(512), 0, s, p);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig
