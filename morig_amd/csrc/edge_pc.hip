// Fused EdgeConv, wave-specialised variant for the wide layers (H = 128, 256) on the split-fp16 path.
//
// The symmetric tile kernel (tile_gemm.hip) alternates "all waves stage the operand tile" and "all waves issue
// MFMAs", separated by barriers: on the split-fp16 path the MFMA phase is so short (3 x 32-cycle MFMAs per 16 k)
// that the VALU work of producing the operand tile -- gather two per-vertex rows, add, ReLU, split into fp16
// hi/lo -- dominates and cannot overlap it. Here a 512-thread workgroup splits into
//     waves 0-3  consumers: ds_read fragments + MFMA only (accumulators live here),
//     waves 4-7  producers: global gathers (two K-chunks in flight in registers), VALU, ds_write,
// meeting at ONE barrier per K-chunk on a two-stage LDS ring. Matrix pipe and VALU/LDS/VMEM pipes of every
// SIMD are then busy at the same time (MI355X_MICROARCH.md: MFMA and VALU are separate pipes; waves of
// different roles co-issue). Epilogue (segmented max over destination segments) as in tile_gemm.hip,
// with all 8 waves scanning.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace morig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));

template <int H>
__global__ __launch_bounds__(512, (H == 128 ? 4 : 2)) void edge_pc_kernel(const EdgePcParams p) {   // H=128: two workgroups per CU
    constexpr int BM = 128, KC = 32, LDB = 144;         // LDB: bytes per LDS row = [32 hi | 32 lo | 16 pad]
    constexpr int NT = H / 64;                          // consumer wave tile: 64 rows x H/2 cols
    constexpr int MT = 2;
    constexpr int NCHUNK = H / KC;
    constexpr int STAGE = (BM + H) * LDB;               // bytes per ring stage
    constexpr int ZC = 64, ZLD = ZC + 1;
    constexpr int SMB = (2 * STAGE > BM * ZLD * 4) ? 2 * STAGE : BM * ZLD * 4;
    __shared__ __attribute__((aligned(16))) char smem[SMB + BM * 4];
    int* sseg = reinterpret_cast<int*>(smem + SMB);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;

    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int rep = lin / p.tiles_per_rep;
    const int tm = lin - rep * p.tiles_per_rep;
    const int row0 = tm * BM;
    const int Etot = p.rowptr[p.n_nodes];
    if (row0 >= Etot) return;                           // block-uniform

    f32x16 acc[MT][NT];
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = (wave & 3) >> 1, wn = wave & 1;

    if (producer) {
        // ---------------- producers: 256 threads, thread (row = pt/8 + 32 i, 16-byte piece = pt%8) ----------------
        const int pt = tid - 256;
        const int lrow = pt >> 3, lkq = pt & 7;
        const float* pa[4]; const float* pb[4]; bool va[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + 32 * i, row = row0 + r;
            va[i] = row < Etot;
            const int d = va[i] ? p.dstS[row] : -1;
            const int s = va[i] ? p.srcS[row] : 0;
            const size_t base = (size_t)rep * p.rep_in;
            pa[i] = p.A + (base + (va[i] ? d : 0)) * p.lda + 4 * lkq;
            pb[i] = p.B + (base + s) * p.ldb + 4 * lkq;
            if (lkq == 0) sseg[r] = d;
        }
        const float* pw = p.W + (size_t)lrow * p.ldw + 4 * lkq;
        constexpr int PW = H / 32;                      // W passes of 32 rows
        f32x4 ra[2][4], rb[2][4], rw[2][PW];
        auto fetch = [&](int c, auto setc) {
            constexpr int S = decltype(setc)::value;
            const int k0 = c * KC;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                ra[S][i] = z; rb[S][i] = z;
                if (va[i] && !(p.dbg & 4)) {
                    ra[S][i] = *reinterpret_cast<const f32x4*>(pa[i] + k0);
                    rb[S][i] = *reinterpret_cast<const f32x4*>(pb[i] + k0);
                }
            }
#pragma unroll
            for (int i = 0; i < PW; ++i) if (!(p.dbg & 8)) rw[S][i] = *reinterpret_cast<const f32x4*>(pw + (size_t)i * 32 * p.ldw + k0);
        };
        auto stage = [&](int c, auto setc) {
            constexpr int S = decltype(setc)::value;
            char* sA = smem + (c & 1) * STAGE;
            char* sB = sA + BM * LDB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(ra[S][i][q] + rb[S][i][q], 0.f);     // invalid rows were fetched as 0
                typedef float b32x2 __attribute__((ext_vector_type(2)));
                b32x2 h, l;
                float h0, h1, l0, l1;
                split_pair_f16(v[0], v[1], h0, l0);
                split_pair_f16(v[2], v[3], h1, l1);
                h[0] = h0; h[1] = h1; l[0] = l0; l[1] = l1;
                const float amax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));               // v >= 0 after the ReLU
                if (!(amax < 65000.f)) *p.ovf = 1;
                char* rowp = sA + (lrow + 32 * i) * LDB + 8 * lkq;
                if (p.dbg & 16) continue;
                *reinterpret_cast<b32x2*>(rowp) = h;
                *reinterpret_cast<b32x2*>(rowp + 64) = l;
            }
            if (p.dbg & 8) return;
#pragma unroll
            for (int i = 0; i < PW; ++i) *reinterpret_cast<f32x4*>(sB + (lrow + 32 * i) * LDB + 16 * lkq) = rw[S][i];
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        fetch(0, S0{});
        if (NCHUNK > 1) fetch(1, S1{});
#pragma unroll
        for (int c = 0; c < NCHUNK; c += 2) {
            if ((p.dbg & 32) && c >= 2) break;
            stage(c, S0{});
            if (c + 2 < NCHUNK) fetch(c + 2, S0{});
            __syncthreads();                            // B_c
            if (c + 1 < NCHUNK) {
                stage(c + 1, S1{});
                if (c + 3 < NCHUNK) fetch(c + 3, S1{});
                __syncthreads();                        // B_{c+1}
            }
        }
    } else {
        // ---------------- consumers: 4 waves as 2 (rows) x 2 (cols), fragments + MFMA only ----------------
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            if ((p.dbg & 32) && c >= 2) break;
            __syncthreads();                            // B_c: chunk c is in stage c&1
            if (p.dbg & 2) continue;
            const char* sA = smem + (c & 1) * STAGE;
            const char* sB = sA + BM * LDB;
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
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    }
            }
        }
    }

    // ---------------- epilogue: segmented max over the tile's destination segments (all 8 waves scan) ----------------
    float* Z = reinterpret_cast<float*>(smem);
    __syncthreads();                                    // last chunk consumed; sseg visible to everyone
    if (p.dbg & 1) { if (!producer && acc[0][0][0] == 12345.678f) p.Y[0] = 1.f; return; }
    const bool first_cont = p.rowptr[sseg[0]] < row0;
    bool last_cont = false;
    if (row0 + BM < Etot) last_cont = p.rowptr[sseg[BM - 1] + 1] > row0 + BM;
    if (p.quad) {
        // ---- 4-aligned segments (MORIG_CSR_PAD4): every lane's 4 consecutive accumulator rows (r&3) belong to one
        // segment, so they are reduced in registers first; the LDS tile holds 32 quad-rows x H columns in ONE pass ----
        constexpr int ZQ = H + 1;
        constexpr int G = 512 / H, RGQ = 32 / G, EXTQ = 8;
        if (!producer) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = wn * NT * 32 + nt * 32 + l31;
                const float b = p.bias[col], sc = p.scale[col], sh = p.shift[col];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float m = fmaxf(acc[mt][nt][4 * q] + b, 0.f) * sc + sh;
#pragma unroll
                        for (int r = 1; r < 4; ++r) m = fmaxf(m, fmaxf(acc[mt][nt][4 * q + r] + b, 0.f) * sc + sh);
                        Z[(wm * 16 + mt * 8 + 2 * q + hi) * ZQ + col] = m;
                    }
            }
        }
        __syncthreads();
        const int col = tid % H, zg = tid / H;              // zg is wave-uniform (H is a multiple of 64)
        const int q0 = __builtin_amdgcn_readfirstlane(zg * RGQ);
        const float* zcolp = Z + col;
        float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + col;
        auto flush = [&](int sg, float m, int qs, int qend) {
            float* o = obase + (size_t)sg * p.ldy;
            const bool partial = (qs == 0 && first_cont) || (qend == 32 && last_cont);
            if (partial) atomic_max_f32(o, m); else *o = m;
        };
        int cur = __builtin_amdgcn_readfirstlane((zg > 0) ? sseg[4 * q0 - 1] : -2);
        bool open = false, done = false;
        float m = 0.f; int qs = 0, qnext = q0;
#pragma unroll
        for (int bt = 0; bt < (RGQ + EXTQ) / 8; ++bt) {
            const int qb0 = q0 + bt * 8;
            if (done || qb0 >= 32) break;
            float zv[8]; int sv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sv[i] = __builtin_amdgcn_readfirstlane(sseg[4 * (qb0 + i)]); zv[i] = zcolp[(qb0 + i) * ZQ]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (!done) {
                    if (sv[i] != cur) {
                        if (open) flush(cur, m, qs, qb0 + i);
                        if (bt * 8 + i >= RGQ) { open = false; done = true; }
                        else { cur = sv[i]; open = cur >= 0; m = zv[i]; qs = qb0 + i; }
                    } else if (open) m = fmaxf(m, zv[i]);
                }
            }
            qnext = qb0 + 8;
        }
        if (open && !done) {
            int q = qnext;
            while (q < 32 && sseg[4 * q] == cur) { m = fmaxf(m, zcolp[q * ZQ]); ++q; }
            flush(cur, m, qs, q);
        }
        return;
    }
    constexpr int NPASS = NT;                           // one fragment column (2 consumer column-waves x 32) per pass
    constexpr int RG = 16, EXT = 32;                    // 8 row groups of 16
    const int zc = tid & 63, zg = tid >> 6;
    const int r0 = __builtin_amdgcn_readfirstlane(zg * RG);
    auto run_pass = [&](auto cb_const) {
        constexpr int cb = decltype(cb_const)::value;
        if (cb > 0) __syncthreads();                    // previous pass's scan done
        if (!producer) {
            const int col = wn * NT * 32 + cb * 32 + l31;
            const float b = p.bias[col], sc = p.scale[col], sh = p.shift[col];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float v = fmaxf(acc[mt][cb][r] + b, 0.f);
                    Z[rl * ZLD + wn * 32 + l31] = v * sc + sh;
                }
        }
        __syncthreads();
        const int col = (zc >> 5) * NT * 32 + cb * 32 + (zc & 31);
        const float* zcolp = Z + zc;
        float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + col;
        auto flush = [&](int sg, float m, int rs, int rend) {
            float* o = obase + (size_t)sg * p.ldy;
            const bool partial = (rs == 0 && first_cont) || (rend == BM && last_cont);
            if (partial) atomic_max_f32(o, m); else *o = m;
        };
        int cur = __builtin_amdgcn_readfirstlane((zg > 0) ? sseg[r0 - 1] : -2);
        bool open = false, done = false;
        float m = 0.f; int rs = 0, rnext = r0;
#pragma unroll
        for (int bt = 0; bt < (RG + EXT) / 16; ++bt) {
            const int rb0 = r0 + bt * 16;
            if (done || rb0 >= BM) break;
            float zv[16]; int sv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_readfirstlane(sseg[rb0 + i]); zv[i] = zcolp[(rb0 + i) * ZLD]; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (!done) {
                    if (sv[i] != cur) {
                        if (open) flush(cur, m, rs, rb0 + i);
                        if (bt * 16 + i >= RG) { open = false; done = true; }
                        else { cur = sv[i]; open = cur >= 0; m = zv[i]; rs = rb0 + i; }
                    } else if (open) m = fmaxf(m, zv[i]);
                }
            }
            rnext = rb0 + 16;
        }
        if (open && !done) {
            int r = rnext;
            while (r < BM && sseg[r] == cur) { m = fmaxf(m, zcolp[r * ZLD]); ++r; }
            flush(cur, m, rs, r);
        }
    };
    run_pass(std::integral_constant<int, 0>{});
    if constexpr (NPASS > 1) run_pass(std::integral_constant<int, 1>{});
    if constexpr (NPASS > 2) run_pass(std::integral_constant<int, 2>{});
    if constexpr (NPASS > 3) run_pass(std::integral_constant<int, 3>{});
}

int launch_edge_pc(const EdgePcParams& p0, int nblocks, hipStream_t s) {
    EdgePcParams p = p0;
    static const int dbg = [] { const char* e = getenv("MORIG_DEBUG_FLAGS"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    if (p.H == 256) hipLaunchKernelGGL((edge_pc_kernel<256>), dim3(nblocks), dim3(512), 0, s, p);
    else if (p.H == 128) hipLaunchKernelGGL((edge_pc_kernel<128>), dim3(nblocks), dim3(512), 0, s, p);
    else return MORIG_E_UNSUPPORTED;
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig
