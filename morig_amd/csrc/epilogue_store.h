// Store epilogue shared by the GEMM kernels (tile_gemm.hip MODE_STORE, gemm_dma.hip): bias / per-mesh row bias / ReLU /
// BN affine, then an fp32 or split-fp16 store.
//
// A lane owns one COLUMN of an MFMA result tile, so storing straight from the accumulators costs one 4-byte (or two
// 2-byte) store instruction per element: 128 per lane for a 64x128 wave tile, measured at 25-57 % of the whole GEMM.
// Here every wave transposes its tile through a private LDS region, 32 rows at a time (the operand ring is dead by
// then; same-wave LDS traffic executes in order, so no barrier is involved). A lane then owns 4 (fp32) or 8 (split)
// ADJACENT columns of a row: the per-column constants live in registers, the row bias is re-read only when the mesh id
// changes, and the result leaves as 16-byte stores, 1 KB per wave instruction.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace morig {

typedef float ep_f32x16 __attribute__((ext_vector_type(16)));
typedef float ep_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ep_f16x8 __attribute__((ext_vector_type(8)));

// SLAB rows of a 32-row MFMA tile are transposed at a time (32, or 16 where the scratch has to fit beside a live ring stage:
// gemm_dmap.hip); PAD floats between rows (4: conflict-free reads; 0: 2-way ds_write_b32 conflicts, which cost nothing)
template <int NT, int SLAB = 32, int PAD = 4> struct EpilogueTile {
    static constexpr int CW = NT * 32, LDT = CW + PAD, FLOATS = SLAB * LDT;
};

// acc: the wave's MT x NT accumulator tiles (C/D layout of v_mfma_f32_32x32x*: col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)); T: this wave's EpilogueTile<NT>::FLOATS floats of LDS;
// rl_base: block-local row of the wave tile's first row (index into sseg); row0: global row of block-local row 0.
template <int MT, int NT, bool ALLOW16, class P, int SLAB = 32, int PAD = 4>
__device__ __forceinline__ void store_tile_transposed(const P& p, ep_f32x16 (&acc)[MT][NT], float* T, const int* sseg,
                                                      int rl_base, int row0, int Mlim, int colw0, int lane) {
    static_assert(SLAB == 32 || SLAB == 16, "slab = whole or half MFMA row tile");
    constexpr int CW = EpilogueTile<NT, SLAB, PAD>::CW, LDT = EpilogueTile<NT, SLAB, PAD>::LDT;
    constexpr int VW = 4, LPR = CW / VW, RPI = 64 / LPR;            // fp32: 4 columns per lane
    constexpr int VW16 = 8, LPR16 = CW / VW16, RPI16 = 64 / LPR16;  // split: 8 columns = 16 B of hi + 16 B of lo
    constexpr int NV = ALLOW16 ? 8 : 4;
    const int l31 = lane & 31, hi = lane >> 5;
    const bool y16 = ALLOW16 && p.y16 != 0;
    const bool vec_ok = (reinterpret_cast<size_t>(p.Y) & 15) == 0 && (p.ldy & 3) == 0;
    const int cg = y16 ? (lane % LPR16) * VW16 : (lane % LPR) * VW;   // first of my columns inside the wave tile
    const int rsel = y16 ? lane / LPR16 : lane / LPR;
    const int col0 = colw0 + cg;
    const int nv = y16 ? VW16 : VW;
    float cb[NV], cs[NV], ch[NV], rbv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool ok = v < nv && col0 + v < p.N;
        cb[v] = (ok && p.bias) ? p.bias[col0 + v] : 0.f;
        cs[v] = (ok && p.scale) ? p.scale[col0 + v] : 1.f;
        ch[v] = (ok && p.shift) ? p.shift[col0 + v] : 0.f;
        rbv[v] = 0.f;
    }
    int rb_seg = -1;
    bool ovf = false;
    const float lo_bound = p.relu ? 0.f : -INFINITY;
#pragma unroll
    for (int ms = 0; ms < MT * (32 / SLAB); ++ms) {
        const int mt = ms / (32 / SLAB), hs = ms % (32 / SLAB);       // half-slab hs: accumulator registers with (r >> 3) == hs
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (SLAB == 32 || (r >> 3) == hs)
#ifndef EPI_NO_LDSW
                    T[((r & 3) + 8 * ((r >> 2) & (SLAB / 8 - 1)) + 4 * hi) * LDT + nt * 32 + l31] = acc[mt][nt][r];
#else
                    asm volatile("" :: "v"(acc[mt][nt][r]));
#endif
        const int nit = y16 ? SLAB / RPI16 : SLAB / RPI;
        const int rstep = y16 ? RPI16 : RPI;
        // The copy-out loop must contain NO global load: vmcnt counts loads and stores in order, so waiting for a row-bias
        // load issued after the previous row's stores waits for those stores too (measured: ~0.7 us per row pass, the whole
        // cost of this epilogue). `seg` is sorted, so the rows of a slab share one mesh id unless the slab straddles a mesh
        // boundary: the row bias is fetched once, before the loop; only straddling slabs take the per-row path.
        bool per_row = false;
        if (p.rowbias) {
            const int rl0 = rl_base + mt * 32 + hs * SLAB + rsel;
            const int sg0 = sseg[rl0], sg1 = sseg[rl0 + (nit - 1) * rstep];
            if (sg0 != sg1) per_row = true;
            else if (sg0 != rb_seg) {
                rb_seg = sg0;
#pragma unroll
                for (int q = 0; q < NV; ++q)
                    rbv[q] = (q < nv && col0 + q < p.N) ? p.rowbias[(size_t)sg0 * p.ld_rowbias + col0 + q] : 0.f;
            }
        }
        float cbr[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) cbr[q] = cb[q] + rbv[q];        // rbv stays 0 without a row bias
        auto pass = [&](int it, auto check_seg) __attribute__((always_inline)) {
            const int rloc = it * rstep + rsel;                       // row inside the slab
            const int rl = rl_base + mt * 32 + hs * SLAB + rloc;      // row inside the block tile
            const int row = row0 + rl;
            float v[NV];
#ifdef EPI_NO_LDSR
            const ep_f32x4 t0 = {acc[mt][0][0], acc[mt][0][1], acc[mt][0][2], acc[mt][0][3]};
#else
            const ep_f32x4 t0 = *reinterpret_cast<const ep_f32x4*>(T + rloc * LDT + cg);
#endif
            v[0] = t0[0]; v[1] = t0[1]; v[2] = t0[2]; v[3] = t0[3];
            if constexpr (ALLOW16) {
                if (y16) {
#ifdef EPI_NO_LDSR
                    const ep_f32x4 t1 = {acc[mt][1][0], acc[mt][1][1], acc[mt][1][2], acc[mt][1][3]};
#else
                    const ep_f32x4 t1 = *reinterpret_cast<const ep_f32x4*>(T + rloc * LDT + cg + 4);
#endif
                    v[4] = t1[0]; v[5] = t1[1]; v[6] = t1[2]; v[7] = t1[3];
                }
            }
            if (row >= Mlim) return;
            if constexpr (decltype(check_seg)::value) {
                const int sg = sseg[rl];
                if (sg != rb_seg) {
                    rb_seg = sg;
#pragma unroll
                    for (int q = 0; q < NV; ++q)
                        rbv[q] = (q < nv && col0 + q < p.N) ? p.rowbias[(size_t)sg * p.ld_rowbias + col0 + q] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                float add = cb[q];
                if constexpr (decltype(check_seg)::value) add += rbv[q]; else add = cbr[q];   // bias + row bias, pre-added per slab
                v[q] = fmaxf(v[q] + add, lo_bound) * cs[q] + ch[q];                          // ReLU as a lower bound (-inf: none)
            }
            if (!y16) {
                float* o = p.Y + (size_t)row * p.ldy + col0;
                if (vec_ok && col0 + VW <= p.N) { ep_f32x4 w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<ep_f32x4*>(o) = w; }
                else { for (int q = 0; q < VW; ++q) if (col0 + q < p.N) o[q] = v[q]; }
            } else if constexpr (ALLOW16) {
                // split layout: each 32-column chunk is [32 hi halves | 32 lo halves]; my 8 columns never straddle a chunk.
                // hi = fp16(v) truncated (one v_cvt_pkrtz per pair), lo = fp16(v - hi) rounded to nearest: hi + lo == v to ~2^-22
                // lo through v_fma_mix (f32 v * 1.0 - f16 hi, rounded into one half of the destination): bit-identical to the
                // cvt / sub / cvt sequence (v - hi is exact in fp32), 3 VALU per pair instead of 6 -- this epilogue is VALU-bound
                typedef __fp16 ep_h2 __attribute__((ext_vector_type(2)));
                typedef float ep_b32x4 __attribute__((ext_vector_type(4)));
                ep_b32x4 hw, lw;
                float am = 0.f;
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    const ep_h2 h = __builtin_amdgcn_cvt_pkrtz(v[q], v[q + 1]);
                    const float hb = __builtin_bit_cast(float, h);
                    float lb;
                    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(v[q]), "v"(hb));
                    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(v[q + 1]), "v"(hb));
                    hw[q >> 1] = hb; lw[q >> 1] = lb;
                    am = fmaxf(am, fmaxf(fabsf(v[q]), fabsf(v[q + 1])));
                }
                const ep_f16x8 hv = __builtin_bit_cast(ep_f16x8, hw), lv = __builtin_bit_cast(ep_f16x8, lw);
                if (!(am < 65000.f)) ovf = true;
                char* o = reinterpret_cast<char*>(p.Y + (size_t)row * p.ldy) + (col0 >> 5) * 128 + (col0 & 31) * 2;
#ifdef EPI_NO_STORE
                if (am == 12345.678f)
#endif
                if (vec_ok && col0 + VW16 <= p.N) {
                    *reinterpret_cast<ep_f16x8*>(o) = hv;
                    *reinterpret_cast<ep_f16x8*>(o + 64) = lv;
                } else {
                    for (int q = 0; q < 8; ++q) if (col0 + q < p.N) {
                        reinterpret_cast<_Float16*>(o)[q] = hv[q]; reinterpret_cast<_Float16*>(o + 64)[q] = lv[q];
                    }
                }
            }
        };
        if (!per_row) { for (int it = 0; it < nit; ++it) pass(it, std::false_type{}); }
        else          { for (int it = 0; it < nit; ++it) pass(it, std::true_type{}); }
    }
    if (ovf) *p.ovf = 1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Register epilogue for TRANSPOSED accumulators (gemm_dma.hip / gemm_dmap.hip store launches). The kernels issue their MFMAs
// with the operands swapped -- A = W fragment, B = X fragment -- so the result tile is D^T: a lane owns ONE OUTPUT ROW
// (m = lane & 31) and, per 32 x 32 tile, the 16 columns n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): four groups of 4 ADJACENT
// columns. One v_permlane32_swap per register pair exchanges groups between the two half-waves, after which a lane holds 8
// consecutive columns twice (16 P + 8 (lane >> 5) ..+7, P = 0, 1): its results leave as 16-byte stores straight from the
// accumulators -- no LDS transposition, no barrier in front of the epilogue, no scratch to reserve in the operand ring.
// The bias (and the per-mesh row bias when the tile lies in one mesh) is the accumulators' INITIAL value, so what is left per
// element is ReLU / BN affine when the layer has them (constants: broadcast ds_read_b128 from a 3 x BN float panel) and the
// fp16 (hi, lo) split for split-layout outputs.
//   acc[mt][nt][r] (before the exchange) = Y[row0 + rl_base + 32 mt + (lane & 31)][colw0 + 32 nt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)]
//   ps / pt: LDS panels (scale, shift) indexed by the block-local column, or nullptr; cl0 = block-local column of colw0
//   rb_slow: the tile spans several meshes, the row bias was NOT folded into the accumulators: added here per row (rare)
// CT: the launch-uniform switches (output layout, ReLU, column affine; no per-row bias left to add) as template constants -- the
// epilogue tests them once per 8-column piece, 16-32 pieces per wave and tile: ~100-146 scalar branches per tile in the disassembly
template <int MT, int NT, class P, bool CT, bool Y16, bool RELU, bool AFF>
__device__ __forceinline__ void store_tile_regs_t(const P& p, ep_f32x16 (&acc)[MT][NT], const float* ps, const float* pt,
                                                  int rl_base, int row0, int Mlim, int colw0, int cl0, int lane, bool rb_slow) {
    const int l31 = lane & 31, hi = lane >> 5;
    const bool y16 = CT ? Y16 : (p.y16 != 0);
    const bool relu = CT ? RELU : (p.relu != 0), aff = CT ? AFF : (ps != nullptr);
    if (CT) rb_slow = false;
    bool ovf = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = row0 + rl_base + mt * 32 + l31;
        const bool ok = row < Mlim;
        float* yrow = p.Y + (size_t)(ok ? row : 0) * p.ldy;
        const float* rbrow = nullptr;
        if (rb_slow && ok) rbrow = p.rowbias + (size_t)p.seg[row] * p.ld_rowbias;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {                   // registers 8 P2 + q  <->  8 P2 + 4 + q of the other half-wave
                const int a = (q >> 2) * 8 + (q & 3), b = a + 4;
                // (scalar copies first: __builtin_bit_cast applied to an ext-vector ELEMENT expression casts the whole vector and
                // takes element 0 with this compiler -- every swap then exchanged register 0 with itself)
                const float fa = acc[mt][nt][a], fb = acc[mt][nt][b];
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(fa), __float_as_uint(fb), false, false);
                const unsigned ua = sw[0], ub = sw[1];
                acc[mt][nt][a] = __uint_as_float(ua);
                acc[mt][nt][b] = __uint_as_float(ub);
            }
#pragma unroll
            for (int P2 = 0; P2 < 2; ++P2) {
                const int c8 = nt * 32 + 16 * P2 + 8 * hi;   // my 8 columns inside the wave tile
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = acc[mt][nt][8 * P2 + q];
                if (rbrow) {
                    const ep_f32x4 r0 = *reinterpret_cast<const ep_f32x4*>(rbrow + colw0 + c8);
                    const ep_f32x4 r1 = *reinterpret_cast<const ep_f32x4*>(rbrow + colw0 + c8 + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] += r0[q]; v[4 + q] += r1[q]; }
                }
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (aff) {
                    const ep_f32x4 s0 = *reinterpret_cast<const ep_f32x4*>(ps + cl0 + c8), s1 = *reinterpret_cast<const ep_f32x4*>(ps + cl0 + c8 + 4);
                    const ep_f32x4 t0 = *reinterpret_cast<const ep_f32x4*>(pt + cl0 + c8), t1 = *reinterpret_cast<const ep_f32x4*>(pt + cl0 + c8 + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = v[q] * s0[q] + t0[q]; v[4 + q] = v[4 + q] * s1[q] + t1[q]; }
                }
                if (!ok) continue;
                const int col = colw0 + c8;
                if (!y16) {
                    ep_f32x4 w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
                    // (plain stores: with the nontemporal hint the short-K launches ran 40 % SLOWER, profiles/r03e_epilogue_nt_ab.txt)
                    *reinterpret_cast<ep_f32x4*>(yrow + col) = w0;
                    *reinterpret_cast<ep_f32x4*>(yrow + col + 4) = w1;
                } else {
                    // split layout: each 32-column chunk is [32 hi halves | 32 lo halves]; my 8 columns never straddle a chunk
                    typedef __fp16 ep_h2 __attribute__((ext_vector_type(2)));
                    typedef float ep_b32x4 __attribute__((ext_vector_type(4)));
                    ep_b32x4 hw, lw;
                    float am = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        const ep_h2 h = __builtin_amdgcn_cvt_pkrtz(v[q], v[q + 1]);
                        const float hb = __builtin_bit_cast(float, h);
                        float lb;
                        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(v[q]), "v"(hb));
                        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(v[q + 1]), "v"(hb));
                        hw[q >> 1] = hb; lw[q >> 1] = lb;
                        am = fmaxf(am, fmaxf(fabsf(v[q]), fabsf(v[q + 1])));
                    }
                    if (!(am < 65000.f)) ovf = true;
                    char* o = reinterpret_cast<char*>(yrow) + (col >> 5) * 128 + (col & 31) * 2;
                    *reinterpret_cast<ep_f16x8*>(o) = __builtin_bit_cast(ep_f16x8, hw);
                    *reinterpret_cast<ep_f16x8*>(o + 64) = __builtin_bit_cast(ep_f16x8, lw);
                }
            }
        }
    }
    if (ovf) *p.ovf = 1;
}

template <int MT, int NT, class P>
__device__ __forceinline__ void store_tile_regs(const P& p, ep_f32x16 (&acc)[MT][NT], const float* ps, const float* pt,
                                                int rl_base, int row0, int Mlim, int colw0, int cl0, int lane, bool rb_slow) {
#ifdef MORIG_EPI_GENERIC                              // measurement variant: the one body with run-time switches
    store_tile_regs_t<MT, NT, P, false, false, false, false>(p, acc, ps, pt, rl_base, row0, Mlim, colw0, cl0, lane, rb_slow);
#else
    if (rb_slow) { store_tile_regs_t<MT, NT, P, false, false, false, false>(p, acc, ps, pt, rl_base, row0, Mlim, colw0, cl0, lane, rb_slow); return; }
    const int key = (p.y16 != 0 ? 1 : 0) | (p.relu != 0 ? 2 : 0) | (ps != nullptr ? 4 : 0);          // launch-uniform
#define MORIG_EPI_CASE(K, Y, R, A) case K: store_tile_regs_t<MT, NT, P, true, Y, R, A>(p, acc, ps, pt, rl_base, row0, Mlim, colw0, cl0, lane, false); break;
    switch (key) {
        MORIG_EPI_CASE(0, false, false, false) MORIG_EPI_CASE(1, true, false, false) MORIG_EPI_CASE(2, false, true, false)
        MORIG_EPI_CASE(3, true, true, false) MORIG_EPI_CASE(4, false, false, true) MORIG_EPI_CASE(5, true, false, true)
        MORIG_EPI_CASE(6, false, true, true) MORIG_EPI_CASE(7, true, true, true)
    }
#undef MORIG_EPI_CASE
#endif
}

}  // namespace morig
