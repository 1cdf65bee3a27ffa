// Dense vertex layer on the split-fp16 path when BOTH operands are already in the split layout (X from the
// producing layer's epilogue, W pre-split by the host): the operand tiles need no VALU work at all, so they are
// streamed HBM/L2 -> LDS with global_load_lds (LDS-DMA, 16 bytes per lane, no VGPR round trip) into a 4-stage
// ring, three K-chunks ahead of the MFMAs, with ONE raw s_barrier and a COUNTED s_waitcnt vmcnt per chunk
// (cdna_hip_programming.md section 5: "Pipelining across barriers", rule 21 for the swizzle).
//
// LDS image of a stage: [128 rows][128 B] for X then the same for W, a row = [32 hi | 32 lo] halves of one
// 32-column chunk = 8 slots of 16 B. DMA writes are lane-linear (wave base + lane*16), so the bank-conflict fix
// is an XOR swizzle applied on the SOURCE side: LDS slot p of row r holds logical slot p ^ ((r >> 1) & 7); fragment
// reads apply the same involution. A 16-lane read group (16 distinct rows mod 16) then hits 16 distinct slots.
#include "common.h"
#include "epilogue_store.h"
#include <stdlib.h>
#include <stdlib.h>

namespace morig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// Tile shapes. The split layout doubles the operand bytes per MFMA (hi and lo fragments), so LDS bandwidth --
// fragment reads plus the DMA writes -- is what binds a 128x128 / 64x64-per-wave tile (measured plateau
// ~260 TFLOP/s for every staging scheme). BIG = 256x256 block, 8 waves of 64x128: 3x fewer LDS bytes per MFMA.
// TR (store launches of the 256 x 256 tile): the MFMAs run with the operands swapped (A = W fragment, B = X fragment), the result
// tile is D^T and leaves through the register epilogue (epilogue_store.h: store_tile_regs) -- no LDS transposition; the bias and,
// when the tile lies in one mesh, the row bias are the accumulators' initial value (panel in LDS, filled under the prologue DMA).
// TAIL [r06]: the last `tail_chunks` 32-column chunks of X come from p.Xt[row % tail_rows] (morig_gemm_args.X_tail: a replica-invariant
// block of a concatenated input, e.g. a unit's position-branch features under the keyframe loop, read from ONE copy instead of being
// copied into every replica's row first). A separate instantiation: the four extra lane offsets stay out of the other launches.
template <int BM, int BN, int NT, int DMA_NS, bool TR = false, bool TAIL = false>
__global__ __launch_bounds__((BM / 64) * (BN / (32 * NT)) * 64) void gemm16_dma_kernel(const GemmDmaParams p) {
    constexpr int MT = 2;
    constexpr int WNW = BN / (32 * NT);          // waves along N
    constexpr int NW = (BM / 64) * WNW;          // waves per block
    constexpr int DMA_STAGE = (BM + BN) * 128;   // bytes: X tile + W tile of one 32-column chunk
    constexpr int XJ = BM / 8 / NW, WJ = BN / 8 / NW;     // 1-KiB DMA instructions per wave per chunk (X, W)
    constexpr int PER_CHUNK = XJ + WJ;
    constexpr int TBYTES = NW * EpilogueTile<NT>::FLOATS * 4;     // per-wave transposition tiles of the store epilogue
    constexpr int RING = DMA_NS * DMA_STAGE;
    constexpr int SM0 = (TR || RING > TBYTES) ? RING : TBYTES;
    constexpr int PANEL = TR ? 3 * BN * 4 : 0;                    // TR: [bias (+ row bias) | scale | shift] of the tile's BN columns
    __shared__ __attribute__((aligned(128))) char smem[SM0 + BM * 4 + PANEL];
    int* sseg = reinterpret_cast<int*>(smem + SM0);
    float* pan = reinterpret_cast<float*>(smem + SM0 + BM * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WNW, wn = wave % WNW;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = lin % p.tiles_n, tm = lin / p.tiles_n;
    const int row0 = tm * BM;

#ifdef MORIG_DMA_TRACE
#define DMA_TS(k) do { if ((int)blockIdx.x == (int)(gridDim.x / 2 + 8) && lane == 0) p.trace[(k) * 8 + wave] = __builtin_readcyclecounter(); } while (0)
#else
#define DMA_TS(k) do { } while (0)
#endif
    DMA_TS(0);
    if (!TR && p.seg != nullptr && tid < BM) sseg[tid] = (row0 + tid < p.M) ? p.seg[row0 + tid] : 0;
    // TR: this thread's column constants (loads issued AHEAD of the prologue DMA: vmcnt retires in order, so they are back first)
    float pv_b = 0.f, pv_s = 1.f, pv_t = 0.f;
    bool rb_slow = false;
    if constexpr (TR) {
        if (tid < BN) {
            const int col = tn * BN + tid;
            if (p.bias) pv_b = p.bias[col];
            if (p.scale) { pv_s = p.scale[col]; pv_t = p.shift[col]; }
        }
        if (p.rowbias != nullptr) {
            const int s0 = p.seg[row0], s1 = p.seg[min(row0 + BM, p.M) - 1];      // block-uniform: `seg` is sorted
            rb_slow = s0 != s1;
            if (!rb_slow && tid < BN) pv_b += p.rowbias[(size_t)s0 * p.ld_rowbias + tn * BN + tid];
        }
    }

    // ---- per-lane DMA sources: wave w moves row blocks (8 rows x 128 B = 1 KiB per instruction) XJ*w .. XJ*w+XJ-1.
    // Addresses are (block-uniform base in SGPRs) + (32-bit per-lane byte offset): half the address VGPRs of pointers ----
    unsigned ox[XJ], ow[WJ];
    const int rsub = lane >> 3;                  // row inside the 8-row block
    const int pslot = lane & 7;                  // physical 16-byte slot inside the row
    const char* xbase = reinterpret_cast<const char*>(p.X + (size_t)row0 * p.ldx);
    const char* wbase = reinterpret_cast<const char*>(p.W + (size_t)(tn * BN) * p.ldw);
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int r = (wave * XJ + j) * 8 + rsub;
        const int lslot = pslot ^ ((r >> 1) & 7);  // rule 21: swizzle the SOURCE, keep the LDS destination linear
        int xr = r; if (row0 + xr >= p.M) xr = p.M - 1 - row0;   // clamp: rows past M are never stored
        ox[j] = ((unsigned)xr * (unsigned)p.ldx + 4u * lslot) * 4u;
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int r = (wave * WJ + j) * 8 + rsub;
        ow[j] = ((unsigned)r * (unsigned)p.ldw + 4u * (pslot ^ ((r >> 1) & 7))) * 4u;
    }
    const int nchunk = (p.K + 31) / 32;
    // TAIL: chunk c >= c_tail is chunk c - c_tail of tail row (global row % tail_rows); offsets relative to the tail's base
    unsigned ot[TAIL ? XJ : 1];
    const int c_tail = TAIL ? nchunk - p.tail_chunks : nchunk;
    const char* tbase = reinterpret_cast<const char*>(p.Xt);
    if constexpr (TAIL) {
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int r = (wave * XJ + j) * 8 + rsub;
            int gr = row0 + r; if (gr >= p.M) gr = p.M - 1;
            ot[j] = ((unsigned)(gr % p.tail_rows) * (unsigned)p.ldt + 4u * (unsigned)(pslot ^ ((r >> 1) & 7))) * 4u;
        }
    }
    // the source of X piece j of chunk c: (block-uniform base, lane offset)
    auto x_src = [&](int c, int j) __attribute__((always_inline)) -> const char* {
        unsigned o = ox[j];
        const char* b = xbase + c * 128;
        if constexpr (TAIL) { if (c >= c_tail) { o = ot[j]; b = tbase + (c - c_tail) * 128; } }
        asm volatile("" : "+v"(o));                  // keep (scalar base + lane offset) addressing
        return b + o;
    };
    auto issue_x = [&](int c) {
        char* st = smem + (c % DMA_NS) * DMA_STAGE;
#pragma unroll
        for (int j = 0; j < XJ; ++j)
            __builtin_amdgcn_global_load_lds((glb_void_t*)x_src(c, j), (lds_void_t*)(st + (wave * XJ + j) * 1024), 16, 0, 0);
    };
    auto issue_w = [&](int c) {
        char* st = smem + (c % DMA_NS) * DMA_STAGE;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            unsigned o = ow[j];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((glb_void_t*)(wbase + c * 128 + o), (lds_void_t*)(st + BM * 128 + (wave * WJ + j) * 1024), 16, 0, 0);
        }
    };
    auto issue = [&](int c) { issue_x(c); issue_w(c); };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // prologue: NS-1 chunks in flight (PER_CHUNK DMA instructions per chunk per wave)
#pragma unroll
    for (int c = 0; c < DMA_NS - 1; ++c) if (c < nchunk) issue(c);
    if constexpr (TR) { if (tid < BN) { pan[tid] = pv_b; pan[BN + tid] = pv_s; pan[2 * BN + tid] = pv_t; } }

    const int x7 = (l31 >> 1) & 7;               // 128-B rows: slot position in the 256-B bank window = 8*(r&1) + slot,
                                                 // so XOR-ing with (r>>1)&7 makes any 16 consecutive rows conflict-free
    const int aoff = (wm * 64 + l31) * 128, boff = BM * 128 + (wn * NT * 32 + l31) * 128;
    // software pipeline: the fragments of the NEXT 16-k step are always in flight (ds_read) while the 12 MFMAs
    // of the current step issue, and the wait + barrier for chunk c+1 sit in the MIDDLE of chunk c
    struct Frag { f16x8 ah[MT], al[MT], bh[NT], bl[NT]; };
    auto load_frag = [&](Frag& f, int c, int s2) {
        const char* st = smem + (c % DMA_NS) * DMA_STAGE;
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
    // one split product term: x (a fragment of X) times w (a fragment of W); TR swaps the operand roles (D^T = W X^T)
    auto mm = [&](const f16x8& x, const f16x8& w, f32x16& c) __attribute__((always_inline)) {
        if constexpr (TR) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
        else              c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, w, c, 0, 0, 0);
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                mm(f.al[mt], f.bh[nt], acc[mt][nt]);
#ifndef MORIG_2MFMA
                mm(f.ah[mt], f.bl[nt], acc[mt][nt]);
#endif
                mm(f.ah[mt], f.bh[nt], acc[mt][nt]);
            }
    };
    auto wait_chunk = [&](int c) {               // chunk c landed for this wave: only younger chunks' DMAs outstanding
        int younger = nchunk - 1 - c;
        if (younger > DMA_NS - 2) younger = DMA_NS - 2;
        static_assert(DMA_NS >= 2 && DMA_NS <= 4, "ring depth");
        if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_CHUNK) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // one (mt, nt) pair of accumulators = 6 MFMAs with the dependent ones two slots apart
    auto mma_pair = [&](const Frag& f, int mt, int nt0) __attribute__((always_inline)) {
        mm(f.al[mt], f.bh[nt0],     acc[mt][nt0]);
        mm(f.al[mt], f.bh[nt0 + 1], acc[mt][nt0 + 1]);
#ifndef MORIG_2MFMA           // measurement build: W rounded to fp16, i.e. the x_hi * w_lo product dropped
        mm(f.ah[mt], f.bl[nt0],     acc[mt][nt0]);
        mm(f.ah[mt], f.bl[nt0 + 1], acc[mt][nt0 + 1]);
#endif
        mm(f.ah[mt], f.bh[nt0],     acc[mt][nt0]);
        mm(f.ah[mt], f.bh[nt0 + 1], acc[mt][nt0 + 1]);
    };
    auto issue_one = [&](int c, int j) __attribute__((always_inline)) {       // DMA instruction j of chunk c (X pieces, then W pieces)
        char* st = smem + (c % DMA_NS) * DMA_STAGE;
        if (j < XJ) {
            __builtin_amdgcn_global_load_lds((glb_void_t*)x_src(c, j), (lds_void_t*)(st + (wave * XJ + j) * 1024), 16, 0, 0);
        } else {
            unsigned o = ow[j - XJ];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_global_load_lds((glb_void_t*)(wbase + c * 128 + o), (lds_void_t*)(st + BM * 128 + (wave * WJ + j - XJ) * 1024), 16, 0, 0);
        }
    };
    Frag f0, f1;
    wait_chunk(0);
    if constexpr (TR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my panel writes have landed ...
    __builtin_amdgcn_s_barrier();                                            // ... and everyone's
    if (DMA_NS - 1 < nchunk) issue(DMA_NS - 1);
    if constexpr (TR) {
        // accumulators start at bias (+ row bias): register r of tile nt is column 32 nt + (r & 3) + 8 (r >> 2) + 4 hi
        typedef float pf32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const pf32x4 b4 = *reinterpret_cast<const pf32x4*>(pan + wn * NT * 32 + nt * 32 + 8 * g4 + 4 * hi);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[mt][nt][4 * g4 + q] = b4[q];
            }
    }
    load_frag(f0, 0, 0);
    // Measured on the EdgeConv kernels (tools/gpu_edge_ablate.sh): an LDS-DMA instruction costs the issuing wave ~100 cycles of issue
    // stall, and the two waves of a SIMD leave the chunk barrier together -- with all PER_CHUNK instructions issued back to
    // back right behind it the matrix pipe of every SIMD idles for the whole burst. MORIG_DMA_SPREAD (default) issues them
    // in pairs between groups of 6 MFMAs of the second half-chunk instead (sched_barrier pins the order), so a wave stalled
    // on a DMA slot leaves the pipe to its partner's queued MFMAs.
#ifdef MORIG_DMA_BURST
    constexpr bool SPREAD = false;
#else
    constexpr bool SPREAD = (NT == 4 && MT == 2 && PER_CHUNK == 8);
#endif
    for (int c = 0; c < nchunk; ++c) {
        load_frag(f1, c, 1);
        mma(f0);
        const bool more = c + 1 < nchunk;
        const bool dma = c + DMA_NS < nchunk;
        if (more) {
            wait_chunk(c + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my own reads of chunk c have returned ...
            __builtin_amdgcn_s_barrier();        // ... and everyone's: chunk c lives in registers, stage c%NS is free
            if constexpr (!SPREAD) { if (dma) issue(c + DMA_NS); }
            load_frag(f0, c + 1, 0);
        }
        if constexpr (SPREAD) {
            const bool go = more && dma;
            mma_pair(f1, 0, 0); __builtin_amdgcn_sched_barrier(0);
            if (go) { issue_one(c + DMA_NS, 0); issue_one(c + DMA_NS, 1); }
            __builtin_amdgcn_sched_barrier(0);
            mma_pair(f1, 0, 2); __builtin_amdgcn_sched_barrier(0);
            if (go) { issue_one(c + DMA_NS, 2); issue_one(c + DMA_NS, 3); }
            __builtin_amdgcn_sched_barrier(0);
            mma_pair(f1, 1, 0); __builtin_amdgcn_sched_barrier(0);
            if (go) { issue_one(c + DMA_NS, 4); issue_one(c + DMA_NS, 5); }
            __builtin_amdgcn_sched_barrier(0);
            mma_pair(f1, 1, 2); __builtin_amdgcn_sched_barrier(0);
            if (go) { issue_one(c + DMA_NS, 6); issue_one(c + DMA_NS, 7); }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            mma(f1);
        }
    }

    DMA_TS(1);
    const int colw0 = tn * BN + wn * NT * 32;
    if (!TR && p.pool != nullptr) {
        // ---- pooled epilogue (scatter_max over meshes): `seg` is sorted, so the 64 rows of a wave tile almost always
        // belong to ONE mesh: reduce them in registers (32 values per lane, then the two half-waves) and issue one
        // integer-atomic float max per column; a wave tile that straddles meshes falls back to per-element atomics ----
        const int rfirst = row0 + wm * 64;
        if (rfirst >= p.M) return;
        const int rlast = min(rfirst + 63, p.M - 1);
        const int s0 = p.seg[rfirst];
        const bool uni = s0 == p.seg[rlast];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = colw0 + nt * 32 + l31;
            const bool cok = col < p.N;
            const float b = (p.bias && cok) ? p.bias[col] : 0.f;
            const float sc = (p.scale && cok) ? p.scale[col] : 1.f;
            const float shf = (p.shift && cok) ? p.shift[col] : 0.f;
            float m = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rfirst + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float v = acc[mt][nt][r] + b;
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    v = v * sc + shf;
                    if (row < p.M) {
                        if (uni) m = fmaxf(m, v);
                        else if (cok) atomic_max_f32(p.pool + (size_t)p.seg[row] * p.ld_pool + col, v);
                    }
                }
            if (uni) {
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                if (hi == 0 && cok && m > -INFINITY) atomic_max_f32(p.pool + (size_t)s0 * p.ld_pool + col, m);
            }
        }
        return;
    }
    // ---- epilogue: bias / per-mesh row bias / ReLU / BN affine, fp32 or split-fp16 store (epilogue_store.h) ----
    if constexpr (TR) {
        if (p.dbg & 1) { if (acc[0][0][0] == 12345.678f) p.Y[0] = 1.f; return; }
        DMA_TS(2);
        store_tile_regs<MT, NT>(p, acc, p.scale ? pan + BN : nullptr, pan + 2 * BN, wm * 64, row0, p.M, colw0, wn * NT * 32, lane, rb_slow);
        DMA_TS(3);
        return;
    }
    __syncthreads();                             // every wave is done with the ring; sseg visible
    if (p.dbg & 1) { if (acc[0][0][0] == 12345.678f) p.Y[0] = 1.f; return; }
    DMA_TS(2);
    store_tile_transposed<MT, NT, true>(p, acc, reinterpret_cast<float*>(smem) + wave * EpilogueTile<NT>::FLOATS, sseg,
                                        wm * 64, row0, p.M, colw0, lane);
    DMA_TS(3);
#ifdef MORIG_DMA_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DMA_TS(4);
#endif
}

int launch_gemm16_dma(const GemmDmaParams& p0, int tiles_m128, hipStream_t s) {
    GemmDmaParams p = p0;
    static const int dbg = [] { const char* e = getenv("MORIG_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    static const int mode = [] { const char* e = getenv("MORIG_DMA_TILE"); return e ? atoi(e) : 256; }();
    // persistent variant (gemm_dmap.hip): the next tile's first chunk lands under the epilogue. Measured in one call
    // (profiles/r02_gemm_persistent_ab.txt): -3.5 % at K = 832 -> N = 1024 and on the pooled launches, -0.6 % at K = 1862, but +2.5..4 %
    // on short-K or single-N-tile shapes (its 16-row epilogue slabs and the tile bookkeeping cost more than the ~2.5 us cold start
    // they hide), so it takes the pooled launches and the deep, wide stores only. MORIG_DMA_PERSIST=0 / 1 forces never / always.
    static const int persist = [] { const char* e = getenv("MORIG_DMA_PERSIST"); return e ? (e[0] == '0' ? 0 : 2) : 1; }();
    const bool deep_wide = p.pool != nullptr || (p.K >= 768 && p.N >= 1024);
    // the persistent STORE kernel has only the register epilogue (16-byte vector stores to Y, 16-byte loads of rowbias): an output
    // or row-bias that is not 16-byte aligned stays on the one-tile kernel below, whose <.., false> form stores scalars
    const bool vec_ok = p.pool != nullptr ||
                        ((reinterpret_cast<uintptr_t>(p.Y) & 15) == 0 && (p.ldy & 3) == 0 &&
                         (!p.rowbias || ((reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0 && (p.ld_rowbias & 3) == 0)));
    if (p.Xt && !(mode == 256 && p.N % 256 == 0 && !p.pool)) return MORIG_E_UNSUPPORTED;      // K tails: the 256 x 256 store kernel only
    // [r06] few rows (one to four meshes per launch: M = 4 096 is 16 row tiles of 256): when the launch has at most MORIG_DMA_SMALL_TILES
    // (default 256 = one per CU) 256 x 256 tiles, CUs idle or run a single workgroup -- the 128 x 128 kernel below makes four times as many
    // workgroups of it, two per CU (its extra LDS-DMA pieces per MFMA do not matter where the chip is not full). Served one-mesh forward
    // 1.79 -> 1.66-1.69 ms, two meshes 2.34 -> 2.27, eight 6.29 -> 6.26 (thresholds 0 / 64 / 128 / 256 on one box: profiles/r07g_*)
    static const int small_tiles = [] { const char* e = getenv("MORIG_DMA_SMALL_TILES"); return e ? atoi(e) : 256; }();
    const bool few_tiles = !p.Xt && p.N % 256 == 0 && (long)cdiv(p.M, 256) * (p.N / 256) <= small_tiles;
    if (!few_tiles && !p.Xt && mode == 256 && p.N % 256 == 0 && p.K > 32 && vec_ok && (persist == 2 || (persist == 1 && deep_wide))) {
        // (a ping-pong schedule for these launches -- the two waves of a SIMD alternating 48-MFMA slots and load slots, gemm_pp.hip in
        // commit a267b6e -- was built, bit-identical, and measured 3-11 % slower: profiles/r04b..r04d, DESIGN section 5 [r04])
        if (!p.pool) prof_retag(K_GEMM16_DMAP);       // the pooled kind already names this kernel
        return launch_gemm16_dmap(p0, s);
    }
    if (mode == 256 && p.N % 256 == 0 && !few_tiles) {
        p.tiles_n = p.N / 256;
        const int nb = cdiv(p.M, 256) * p.tiles_n;
#ifdef MORIG_DMA_TRACE
        static unsigned long long* trace_buf = [] { void* b = nullptr; return hipMalloc(&b, 64 * 8) == hipSuccess ? (unsigned long long*)b : nullptr; }();
        p.trace = trace_buf;
#endif
        // stores: the transposed-accumulator kernel with the register epilogue (16-byte aligned output rows); MORIG_GEMM_TR=0 and
        // pooled launches keep the LDS-transposition epilogue
        static const bool want_tr = [] { const char* e = getenv("MORIG_GEMM_TR"); return !(e && e[0] == '0'); }();
        const bool tr = want_tr && !p.pool && (reinterpret_cast<uintptr_t>(p.Y) & 15) == 0 && (p.ldy & 3) == 0 &&
                        (!p.rowbias || ((reinterpret_cast<uintptr_t>(p.rowbias) & 15) == 0 && (p.ld_rowbias & 3) == 0));
        // [r05] measured and NOT adopted (profiles/r05j_shortk_128_tile_ab.txt): the short-K store launches (the [A | B] producers,
        // K = 64 -> 512, 256 -> 1024) on 128 x 128 tiles, two workgroups per CU, so that one's stores run under the other's MFMAs:
        // 0.698 vs 0.615 ms (K = 256) and 0.233 vs 0.177 ms (K = 64), sustained. The 256 CUs are not in phase, so the chip's HBM writes
        // are already smooth (K = 64: 3.8 TB/s written), and the small tile doubles the LDS-DMA pieces per MFMA. MORIG_DMA_SHORTK=<K>
        // selects it for K <= <K> (A/B switch; default 0 = never)
        static const int shortk = [] { const char* e = getenv("MORIG_DMA_SHORTK"); return e ? atoi(e) : 0; }();
        if (tr && !p.Xt && shortk > 0 && p.K <= shortk && p.N % 128 == 0) {
            p.tiles_n = p.N / 128;
            const int nb128 = cdiv(p.M, 128) * p.tiles_n;
            prof_retag(K_GEMM16_DMA128);
            hipLaunchKernelGGL((gemm16_dma_kernel<128, 128, 2, 2, true>), dim3(nb128), dim3(256), 0, s, p);
            MORIG_LAUNCH_CHECK();
            return MORIG_OK;
        }
        if (p.Xt && !tr) return MORIG_E_UNSUPPORTED;
        if (p.Xt)    hipLaunchKernelGGL((gemm16_dma_kernel<256, 256, 4, 2, true, true>), dim3(nb), dim3(512), 0, s, p);
        else if (tr) hipLaunchKernelGGL((gemm16_dma_kernel<256, 256, 4, 2, true>), dim3(nb), dim3(512), 0, s, p);
        else         hipLaunchKernelGGL((gemm16_dma_kernel<256, 256, 4, 2, false>), dim3(nb), dim3(512), 0, s, p);
#ifdef MORIG_DMA_TRACE
        {
            unsigned long long h[64];
            if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h, p.trace, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "DMA_TRACE K=%d N=%d:", p.K, p.N);
                for (int w = 0; w < 8; w += 3) { fprintf(stderr, " w%d", w); for (int q = 1; q < 5; ++q) fprintf(stderr, " %lld", (long long)(h[q * 8 + w] - h[(q - 1) * 8 + w])); }
                fprintf(stderr, "\n");
            }
        }
#endif
    } else {
        p.tiles_n = cdiv(p.N, 128);
        prof_retag(K_GEMM16_DMA128);
        // 2-stage ring = 64 KB: TWO workgroups per CU, one's prologue / epilogue under the other's main loop
        // (measured 25 % faster than a 4-stage ring at one workgroup per CU)
        hipLaunchKernelGGL((gemm16_dma_kernel<128, 128, 2, 2, false>), dim3(tiles_m128 * p.tiles_n), dim3(256), 0, s, p);
    }
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig
