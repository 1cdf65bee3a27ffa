// The one MFMA tile engine behind both hot operators of the MoRig forward path:
//
//   morig_gemm      dense vertex layer   Y = s*act(X W^T + b + rowbias[seg]) + t   [+ per-mesh column max]
//   morig_edgeconv  fused EdgeConv       out[i] = max_{j->i} s2*relu(W2 (s1*relu(A_i + B_j) + t1) + b2) + t2
//
// Both are a [rows x K] x [K x N] contraction on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma
// chain) with a 128-row workgroup tile; they differ only in how the A operand tile is PRODUCED
// (dense rows vs gather-add-ReLU-affine of two per-vertex rows named by the CSR) and how the
// accumulator tile is CONSUMED (store vs segmented max over rows).
//
// Layout facts used below (cdna_hip_programming.md section 3, MI355X_MICROARCH.md LDS table):
//   * 32x32x2 f32 MFMA: A operand lane l -> A[i = l&31][k = l>>5]; B operand lane l -> B[k = l>>5][j = l&31];
//     D register r of lane l -> D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
//   * K order inside a group of 8 is permuted so that ONE ds_read_b128 feeds 4 consecutive MFMAs:
//     MFMA m of group q multiplies k = 8q+m (lanes 0-31) and k = 8q+4+m (lanes 32-63). Both operands
//     use the same permutation, so the sum is the same set of products.
//   * LDS rows are padded to KC+4 floats: for ds_read_b128 the 16-lane service groups then touch
//     16 distinct 16-byte slots (36*r mod 64 and 20*r mod 64 are 4*(odd*r mod 16)) -> conflict-free.
#include "common.h"
#include "epilogue_store.h"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

namespace morig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));        // what __builtin_amdgcn_cvt_pkrtz returns

// Arithmetic modes of the contraction:
//   PREC_F32   : v_mfma_f32_32x32x2_f32, exact fp32 products (157 TFLOP/s peak).
//   PREC_F16X3 : every fp32 operand is split x = hi + lo with hi = fp16(x), lo = fp16(x - hi); the product
//                is  hi*hi + hi*lo + lo*hi  on v_mfma_f32_32x32x16_f16 with fp32 accumulation (3 MFMAs per
//                16 k = 5.3x the fp32-MFMA rate). fp16 products are exact in fp32, the dropped lo*lo term
//                and the rounding of lo are both ~2^-22 relative: fp32-class accuracy. |x| must stay below
//                the fp16 range; the loader raises `ovf` otherwise and the host re-runs in PREC_F32.
//   PREC_BF16X6 [r06]: both fp32 operands are split IN THE KERNEL into three bf16 limbs x = b1 + b2 + b3 (8 + 8 + 8 mantissa bits, float32's
//                exponent range: no range condition) and the product is the six terms down to 2^-16 of the leading one --
//                b1 b1 + b1 b2 + b2 b1 + b2 b2 + b1 b3 + b3 b1 -- on v_mfma_f32_32x32x16_bf16: float32-class products (the dropped terms are
//                <= 2^-24 relative) at 16/6 = 2.7x the fp32-MFMA rate. Takes plain fp32 X and W (weights that change every step need no
//                host-side image): the train-mode forward contractions and the exact path behind the range guard.
enum { PREC_F32 = 0, PREC_F16X3 = 1, PREC_BF16X3 = 2, PREC_BF16X6 = 3 };   // BF16X3 [r04]: the same 3-MFMA split on bf16 halves (float32 exponent range: the
                                                          // backward contractions, whose operands are gradients); dense fp32-X stores only

// LOAD_EDGE3: an EdgeConv whose vertex input has 3 channels (positions; the keyframe flow of motionNet's first unit): instead of
//             gathering the per-vertex first-layer terms A[dst], B[src] (2 x 4 H bytes per edge row) the loader gathers the two
//             endpoints' 3 inputs (2 x 16 bytes) and evaluates the first Linear for its 4 hidden channels in registers
//             (relu((Wa - Wb) x_i + Wb x_j + b): 28 fused multiply-adds per thread and row).
enum { LOAD_DENSE = 0, LOAD_EDGE = 1, LOAD_EDGE3 = 2 };
enum { MODE_STORE = 0, MODE_POOL = 1, MODE_EDGEMAX = 2 };

struct TileParams {
    int M, N, K;                 // rows (edge: capacity), logical out width, contraction length
    const float* W; int ldw;
    const float* bias; const float* scale; const float* shift; int relu;
    // dense loader
    const float* X; int ldx;
    // edge loader
    const float* A; int lda; const float* B; int ldb;
    const float* X3; int ldx3; const float* W1a; const float* W1b; const float* b1;     // LOAD_EDGE3: inputs [rows][ldx3 >= 4], [Hpad][4] weights
    const int* rowptr; const int* srcS; const int* dstS; int n_nodes; int rep_in; int rep_out; int tiles_per_rep;
    const float* s1; const float* t1;
    // epilogue
    const float* rowbias; int ld_rowbias; const int* seg;
    float* Y; int ldy;           // store target / edge-max target / pool target
    int tiles_n;
    int* ovf;                    // PREC_F16X3: set to 1 when an operand leaves the fp16 range
    int x16, y16;                // PREC_F16X3 dense: X already in split-fp16 layout / store Y in split-fp16 layout
    int dbg;                     // ablation switches for tools/microbench.py (MORIG_DEBUG_FLAGS; 0 in production)
};
enum { DBG_NO_EPILOGUE = 1, DBG_NO_MFMA = 2, DBG_NO_GATHER = 4, DBG_NO_WLOAD = 8, DBG_NO_STAGE = 16 };

static int debug_flags() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MORIG_DEBUG_FLAGS"); v = e ? atoi(e) : 0; }
    return v;
}

// The narrow EdgeConv tiles are latency chains (ids -> gathers -> one or two K-chunks -> scan): occupancy is what hides
// them, so their register budget is capped for 4 (fp32, KC = 16: 6) waves per SIMD (measured -17..21 % at H = 32).
// The dense fp32-X GEMM tile (BN = 128, KC = 32) likewise runs better at 3 waves per SIMD than at 2.
// FAST [r04]: every row of every tile exists (M a multiple of 128), K is a multiple of KC and no debug switch is set -- decided by the
// launcher; the row / column tests, zero fills and switch tests compile out (dense store GEMMs with fp32 X: the GCU units' MLPs of the
// forward, dX of the training step; the counters of the general form: 7.0 VALU + 3.6 SALU instructions per MFMA, matrix pipe 39 % busy).
template <int BN, int KC, int LOAD, int MODE, int PREC, bool FAST = false>
__global__ __launch_bounds__(256, ((BN == 256 && PREC == PREC_F32) ? 2 : (BN == 32 && LOAD != LOAD_DENSE) ? ((PREC == PREC_F32 && KC == 16) ? 6 : 4) : (BN == 64 && LOAD == LOAD_EDGE && PREC == PREC_F16X3) ? 4 : (PREC == PREC_BF16X6) ? (BN >= 256 ? 1 : 2) : (BN == 128 && KC == 32 && LOAD == LOAD_DENSE && MODE != MODE_EDGEMAX && PREC != PREC_F32) ? 3 :
                               (BN == 64 && LOAD == LOAD_DENSE && MODE == MODE_STORE && PREC != PREC_F32) ? 4 : 1)) void tile_kernel(const TileParams p) {
    constexpr int BM = 128;
    constexpr int WN = (BN >= 128) ? 2 : 1;
    constexpr int WM = 4 / WN;
    constexpr int MT = BM / WM / 32;
    constexpr int NT = BN / WN / 32;
    constexpr int LDK = (PREC == PREC_BF16X6 ? (KC / 32) * 48 : KC) + 4;      // floats per LDS row (BF16X6: three limbs = 192 B per 32-column chunk)
    constexpr int TPR = KC / 4;                 // loader threads per tile row
    constexpr int RPP = 256 / TPR;              // tile rows per loader pass
    constexpr int PA = BM / RPP;
    constexpr int PB = (BN + RPP - 1) / RPP;
    constexpr int ZC = BN < 64 ? BN : 64;       // epilogue column block
    constexpr int ZLD = ZC + 1;
    constexpr int SM_MAIN = (BM + BN) * LDK;
    constexpr int SM_Z = (MODE == MODE_STORE) ? 4 * EpilogueTile<NT>::FLOATS      // store: per-wave transposition tiles
                       : (MODE == MODE_EDGEMAX && BN == 32) ? BM * 36 : BM * ZLD;      // narrow EdgeConv: 16-byte Z rows
    constexpr int SM = SM_MAIN > SM_Z ? SM_MAIN : SM_Z;
    static_assert(BN % 32 == 0 && KC % 8 == 0, "tile shape");
    static_assert(PREC == PREC_F32 || KC % 32 == 0, "the split-fp16 LDS image is [hi 32 halves | lo 32 halves] per 32-column chunk");

    __shared__ __attribute__((aligned(16))) float smem[SM + BM];
    float* sA = smem;
    float* sB = smem + BM * LDK;
    int* sseg = reinterpret_cast<int*>(smem + SM);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn, rep;
    if (LOAD == LOAD_DENSE) { tn = lin % p.tiles_n; tm = lin / p.tiles_n; rep = 0; }
    else { rep = lin / p.tiles_per_rep; tm = lin - rep * p.tiles_per_rep; tn = 0; }
    const int row0 = tm * BM;

    constexpr bool IS_EDGE = LOAD != LOAD_DENSE;                               // gathered edge rows (LOAD_EDGE, LOAD_EDGE3)
    constexpr bool EDGE_ROWS = IS_EDGE || (MODE == MODE_EDGEMAX);               // tile rows = sorted edges
    // Edge tiles: the edge count, this thread's edge ids and the ids just outside the tile are all fetched before anything
    // waits (rows are clamped to the arrays' capacity p.M, validity is applied afterwards): ONE memory latency at tile
    // start instead of a chain of two or three. The neighbour ids decide in the epilogue whether the first / last
    // segment is shared with another tile (rowptr look-ups there were two dependent global loads on its critical path).
    int Etot = 0;
    int dprev = -2, dafter = -3;
    int ld_d[BM / (256 / (KC / 4))], ld_s[BM / (256 / (KC / 4))];
    if (EDGE_ROWS) {
        Etot = p.rowptr[p.n_nodes];
        const int lr = tid / (KC / 4);
#pragma unroll
        for (int i = 0; i < BM / (256 / (KC / 4)); ++i) {
            const int rc = min(row0 + lr + i * (256 / (KC / 4)), p.M - 1);
            ld_d[i] = p.dstS[rc];
            ld_s[i] = IS_EDGE ? p.srcS[rc] : 0;
        }
        if (MODE == MODE_EDGEMAX) {
            const int dp = p.dstS[max(row0 - 1, 0)], da = p.dstS[min(row0 + BM, p.M - 1)];
            if (row0 > 0) dprev = dp;
            if (row0 + BM < Etot) dafter = da;
        }
        if (row0 >= Etot) return;               // block-uniform
    }
    const int Mlim = EDGE_ROWS ? Etot : p.M;

    // ---- loader set-up -------------------------------------------------------------------
    const int lrow = tid / TPR, lkq = tid % TPR;
    const float* pa[PA];
    const float* pb[PA];                        // edge only
    bool va[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int r = lrow + i * RPP;
        const int row = row0 + r;
        if (LOAD == LOAD_DENSE) {
            va[i] = FAST || row < Mlim;
            pa[i] = p.X + (size_t)(va[i] ? row : 0) * p.ldx + 4 * lkq;
            pb[i] = nullptr;
            if (MODE == MODE_EDGEMAX) {
                if (lkq == 0) sseg[r] = va[i] ? ld_d[i] : -1;
            } else if (MODE != MODE_STORE || p.seg != nullptr) {
                if (lkq == 0) sseg[r] = (va[i] && p.seg) ? p.seg[row] : -1;
            }
        } else {
            va[i] = row < Etot;
            const int d = va[i] ? ld_d[i] : -1;
            const int s = va[i] ? ld_s[i] : 0;
            const size_t base = (size_t)rep * p.rep_in;
            if (LOAD == LOAD_EDGE3) {
                pa[i] = p.X3 + (base + (va[i] ? d : 0)) * p.ldx3;
                pb[i] = p.X3 + (base + s) * p.ldx3;
            } else {
                pa[i] = p.A + (base + (va[i] ? d : 0)) * p.lda + 4 * lkq;
                pb[i] = p.B + (base + s) * p.ldb + 4 * lkq;
            }
            if (lkq == 0) sseg[r] = d;
        }
    }
    const float* pw = p.W + (size_t)(tn * BN + lrow) * p.ldw + 4 * lkq;

    // LOAD_EDGE3: first-layer rows [channel][8] = {W1a row (3), b1, W1b row (3), 0} in LDS (28 per-thread registers would spill the
    // 128-VGPR budget that keeps 4 waves per SIMD); read channel by channel while staging, reused over the thread's PA rows
    __shared__ __attribute__((aligned(16))) float w3[LOAD == LOAD_EDGE3 ? 32 * 8 : 4];
    if (LOAD == LOAD_EDGE3) {
        if (tid < 32) {
            w3[tid * 8 + 0] = p.W1a[tid * 4 + 0]; w3[tid * 8 + 1] = p.W1a[tid * 4 + 1]; w3[tid * 8 + 2] = p.W1a[tid * 4 + 2]; w3[tid * 8 + 3] = p.b1[tid];
            w3[tid * 8 + 4] = p.W1b[tid * 4 + 0]; w3[tid * 8 + 5] = p.W1b[tid * 4 + 1]; w3[tid * 8 + 6] = p.W1b[tid * 4 + 2]; w3[tid * 8 + 7] = 0.f;
        }
    }
    f32x4 ra[PA], rb[PA], rw[PB];
    f32x4 rs1 = {1.f, 1.f, 1.f, 1.f}, rt1 = {0.f, 0.f, 0.f, 0.f};      // hidden-layer affine of this thread's 4 k's
    auto fetch = [&](int k0) {
        if (LOAD == LOAD_EDGE3) {                           // one chunk (K = H = 32): both endpoints' inputs, 16 bytes each
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                ra[i] = z; rb[i] = z;
                if (va[i] && !(p.dbg & DBG_NO_GATHER)) {
                    ra[i] = *reinterpret_cast<const f32x4*>(pa[i]);
                    rb[i] = *reinterpret_cast<const f32x4*>(pb[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                if ((BN % RPP == 0 || lrow + i * RPP < BN) && !(p.dbg & DBG_NO_WLOAD))
                    rw[i] = *reinterpret_cast<const f32x4*>(pw + (size_t)i * RPP * p.ldw + k0);
            }
            return;
        }
        if (LOAD == LOAD_EDGE && p.s1 != nullptr && k0 + 4 * lkq < p.K) {   // NULL: affine already folded into W2/b2
            rs1 = *reinterpret_cast<const f32x4*>(p.s1 + k0 + 4 * lkq);
            rt1 = *reinterpret_cast<const f32x4*>(p.t1 + k0 + 4 * lkq);
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (!FAST) { ra[i] = z; rb[i] = z; }
            // split-fp16 X: a 16-byte piece holds halves of its whole 32-column chunk -> guard per chunk
            const int k = (PREC == PREC_F16X3 && LOAD == LOAD_DENSE && p.x16) ? ((k0 + 4 * lkq) & ~31) : (k0 + 4 * lkq);
            if (FAST || (va[i] && k < p.K && !(p.dbg & DBG_NO_GATHER))) {
                ra[i] = *reinterpret_cast<const f32x4*>(pa[i] + k0);
                if (IS_EDGE) rb[i] = *reinterpret_cast<const f32x4*>(pb[i] + k0);
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if ((BN % RPP == 0 || lrow + i * RPP < BN) && (FAST || !(p.dbg & DBG_NO_WLOAD)))
                rw[i] = *reinterpret_cast<const f32x4*>(pw + (size_t)i * RPP * p.ldw + k0);
        }
    };
    // BF16X6: four fp32 values -> their three bf16 limbs, 4 halves (8 bytes) each into the [b1 | b2 | b3] thirds of the row's 192-byte chunk
    auto split3_store = [&](char* rowp, const f32x4& v) __attribute__((always_inline)) {
        typedef float b32x2 __attribute__((ext_vector_type(2)));
        b32x2 l1, l2, l3;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float h, r0, r1, m, t;
            split_pair_bf16_rem(v[2 * q], v[2 * q + 1], h, r0, r1);       // h = bf16 pair, r = the exact remainders
            l1[q] = h;
            float s0, s1;
            split_pair_bf16_rem(r0, r1, m, s0, s1);
            l2[q] = m;
            split_pair_bf16(s0, s1, t, h);                                // third limb (h: the fourth, dropped)
            l3[q] = t;
        }
        *reinterpret_cast<b32x2*>(rowp) = l1;
        *reinterpret_cast<b32x2*>(rowp + 64) = l2;
        *reinterpret_cast<b32x2*>(rowp + 128) = l3;
    };
    auto stage = [&](int k0) {
        const int k = k0 + 4 * lkq;
        f32x4 v3[PA];
        if (LOAD == LOAD_EDGE3) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(&w3[(4 * lkq + c) * 8]);        // W1a row, b1
                const f32x4 ws = *reinterpret_cast<const f32x4*>(&w3[(4 * lkq + c) * 8 + 4]);    // W1b row
#pragma unroll
                for (int i = 0; i < PA; ++i) {
                    float h = wa[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) h = fmaf(wa[j], ra[i][j], fmaf(ws[j], rb[i][j], h));
                    v3[i][c] = va[i] ? fmaxf(h, 0.f) : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            f32x4 v = ra[i];
            if (LOAD == LOAD_EDGE3) {
                v = v3[i];
            } else if (LOAD == LOAD_EDGE) {
                if (p.s1 != nullptr) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float h = v[c] + rb[i][c];
                        h = h > 0.f ? h : 0.f;
                        v[c] = va[i] ? (h * rs1[c] + rt1[c]) : 0.f;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c] + rb[i][c], 0.f);      // invalid rows were fetched as 0
                }
            } else if (!FAST && k0 + KC > p.K && !(PREC == PREC_F16X3 && p.x16)) {   // only the last chunk can cross K
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = (k + c < p.K) ? v[c] : 0.f;
            }
            if (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(&sA[(lrow + i * RPP) * LDK + 4 * lkq]) = v;
            } else if (PREC == PREC_BF16X6) {
                split3_store(reinterpret_cast<char*>(sA) + (lrow + i * RPP) * (LDK * 4) + 192 * (lkq >> 3) + 8 * (lkq & 7), v);
            } else if (PREC == PREC_F16X3 && LOAD == LOAD_DENSE && p.x16) {
                // the producer already wrote [32 halves hi | 32 halves lo] per 32-column chunk: plain copy
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(sA) + (lrow + i * RPP) * (LDK * 4) + 16 * lkq) = ra[i];
            } else {
                // hi = fp16(x) (round-toward-zero, 2 per instruction), lo = fp16(x - hi): x - hi is exact in fp32
                // (hi may truncate: lo absorbs it exactly; lo itself is rounded to nearest-even so the split is unbiased; common.h)
                typedef float b32x2 __attribute__((ext_vector_type(2)));
                b32x2 h, l;
                float h0, h1, l0, l1;
                if (PREC == PREC_BF16X3) {
                    split_pair_bf16(v[0], v[1], h0, l0);
                    split_pair_bf16(v[2], v[3], h1, l1);
                } else {
                    split_pair_f16(v[0], v[1], h0, l0);
                    split_pair_f16(v[2], v[3], h1, l1);
                    const float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    if (!(amax < 65000.f)) *p.ovf = 1;                 // also catches NaN
                }
                h[0] = h0; h[1] = h1; l[0] = l0; l[1] = l1;
                char* rowp = reinterpret_cast<char*>(sA) + (lrow + i * RPP) * (LDK * 4) + 128 * (lkq >> 3) + 8 * (lkq & 7);
                *reinterpret_cast<b32x2*>(rowp) = h;
                *reinterpret_cast<b32x2*>(rowp + 64) = l;
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            if (BN % RPP == 0 || lrow + i * RPP < BN) {
                if (PREC == PREC_BF16X6) {                  // plain fp32 weights in, three limbs out (columns past K are zero in the packed rows)
                    f32x4 w = rw[i];
                    if (!FAST && (p.dbg & DBG_NO_WLOAD)) w = f32x4{0.f, 0.f, 0.f, 0.f};
                    split3_store(reinterpret_cast<char*>(sB) + (lrow + i * RPP) * (LDK * 4) + 192 * (lkq >> 3) + 8 * (lkq & 7), w);
                } else {
                    *reinterpret_cast<f32x4*>(&sB[(lrow + i * RPP) * LDK + 4 * lkq]) = rw[i];
                }
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- main loop: register prefetch of chunk c+1 under the MFMAs of chunk c ---------------
    const int nchunk = (p.K + KC - 1) / KC;
    fetch(0);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();                        // previous chunk's fragment reads are done
        if (FAST || !(p.dbg & DBG_NO_STAGE) || c == 0) stage(c * KC);
        __syncthreads();
        if (c + 1 < nchunk) fetch((c + 1) * KC);
        if (!FAST && (p.dbg & DBG_NO_MFMA)) continue;
        if (PREC == PREC_BF16X6) {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            const char* a0 = reinterpret_cast<const char*>(sA) + (wm * MT * 32 + l31) * (LDK * 4) + 16 * hi;
            const char* b0 = reinterpret_cast<const char*>(sB) + (wn * NT * 32 + l31) * (LDK * 4) + 16 * hi;
#pragma unroll
            for (int st2 = 0; st2 < KC / 16; ++st2) {
                const int off = 192 * (st2 >> 1) + 32 * (st2 & 1);     // 32-column chunk, 16-k step inside each 64-byte limb block
                bf16x8 x1[MT], x2[MT], x3[MT], w1[NT], w2[NT], w3[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    x1[mt] = *reinterpret_cast<const bf16x8*>(a0 + mt * 32 * LDK * 4 + off);
                    x2[mt] = *reinterpret_cast<const bf16x8*>(a0 + mt * 32 * LDK * 4 + off + 64);
                    x3[mt] = *reinterpret_cast<const bf16x8*>(a0 + mt * 32 * LDK * 4 + off + 128);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    w1[nt] = *reinterpret_cast<const bf16x8*>(b0 + nt * 32 * LDK * 4 + off);
                    w2[nt] = *reinterpret_cast<const bf16x8*>(b0 + nt * 32 * LDK * 4 + off + 64);
                    w3[nt] = *reinterpret_cast<const bf16x8*>(b0 + nt * 32 * LDK * 4 + off + 128);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {           // smallest terms first
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3[mt], w1[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[mt], w3[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2[mt], w2[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2[mt], w1[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[mt], w2[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1[mt], w1[nt], acc[mt][nt], 0, 0, 0);
                    }
            }
            continue;
        }
        if (PREC != PREC_F32) {
            const char* a0 = reinterpret_cast<const char*>(sA) + (wm * MT * 32 + l31) * (LDK * 4) + 16 * hi;
            const char* b0 = reinterpret_cast<const char*>(sB) + (wn * NT * 32 + l31) * (LDK * 4) + 16 * hi;
#pragma unroll
            for (int st2 = 0; st2 < KC / 16; ++st2) {
                const int off = 128 * (st2 >> 1) + 32 * (st2 & 1);     // 32-column chunk, 16-k step inside it
                f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    ah[mt] = *reinterpret_cast<const f16x8*>(a0 + mt * 32 * LDK * 4 + off);
                    al[mt] = *reinterpret_cast<const f16x8*>(a0 + mt * 32 * LDK * 4 + off + 64);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt] = *reinterpret_cast<const f16x8*>(b0 + nt * 32 * LDK * 4 + off);
                    bl[nt] = *reinterpret_cast<const f16x8*>(b0 + nt * 32 * LDK * 4 + off + 64);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (PREC == PREC_BF16X3) {
                            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                            const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[mt]), xl = __builtin_bit_cast(bf16x8, al[mt]);
                            const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[nt]), wl = __builtin_bit_cast(bf16x8, bl[nt]);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, acc[mt][nt], 0, 0, 0);
                        } else {
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        }
                    }
            }
            continue;
        }
        const float* a0 = sA + (wm * MT * 32 + l31) * LDK + 4 * hi;
        const float* b0 = sB + (wn * NT * 32 + l31) * LDK + 4 * hi;
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(a0 + mt * 32 * LDK + kk * 8);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const f32x4*>(b0 + nt * 32 * LDK + kk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------
    if (!FAST && (p.dbg & DBG_NO_EPILOGUE)) { if (acc[0][0][0] == 12345.678f) p.Y[0] = 1.f; return; }
    const int colw0 = tn * BN + wn * NT * 32;   // first global column of this wave
    if (MODE == MODE_STORE) {
        __syncthreads();                        // all waves are done with the operand tiles; sseg visible
        store_tile_transposed<MT, NT, PREC == PREC_F16X3>(p, acc, smem + wave * EpilogueTile<NT>::FLOATS, sseg,
                                                          wm * MT * 32, row0, Mlim, colw0, lane);
        return;
    }

    // Segmented max over tile rows. The accumulator tile goes through LDS in column blocks of ZC (all four
    // waves write one or two fragments per pass); thread (column, row group) then preloads 16 rows at a time
    // (independent LDS reads, no dependent chain) and runs the segmented scan in registers.
    float* Z = smem;
    bool first_cont = false, last_cont = false;
    if (MODE == MODE_EDGEMAX) {
        // sseg was written before the main loop; every thread passed >= 1 barrier since
        first_cont = sseg[0] == dprev;
        last_cont = sseg[BM - 1] == dafter;
    }
    if constexpr (MODE == MODE_EDGEMAX && BN == 32) {
        // ---- narrow layers (H = 16, 32): the scan is half of the kernel if every thread walks a row group plus the tail
        // of its last segment (17-row segments over 16-row groups: ~3x redundant reads). Here wave 0 lists the segment
        // starts of the tile (two ballots over the 128 destination ids), and a slot of 8 threads -- 4 adjacent columns
        // each, 16-byte LDS reads -- reduces exactly the rows of one segment, slots taking segments round-robin.
        constexpr int ZL = 36;                      // floats per Z row (16-byte rows)
        __shared__ int sstart[BM + 2];              // start row of segment k; [nseg] = BM
        __shared__ int snseg;
        __syncthreads();                            // main-loop reads done
        {
            const int col = tn * BN + l31;
            const float b = (p.bias && col < p.N) ? p.bias[col] : 0.f;
            const float sc = (p.scale && col < p.N) ? p.scale[col] : 1.f;
            const float sh = (p.shift && col < p.N) ? p.shift[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;     // MT = NT = 1 for BN = 32
                float v = acc[0][0][r] + b;
                if (p.relu) v = v > 0.f ? v : 0.f;
                Z[rl * ZL + l31] = v * sc + sh;
            }
        }
        if (wave == 0) {
            const int r0 = lane, r1 = lane + 64;
            const bool f0 = r0 == 0 || sseg[r0] != sseg[r0 - 1];
            const bool f1 = sseg[r1] != sseg[r1 - 1];
            const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (f0) sstart[__popcll(m0 & below)] = r0;
            if (f1) sstart[__popcll(m0) + __popcll(m1 & below)] = r1;
            if (lane == 0) { const int ns = __popcll(m0) + __popcll(m1); snseg = ns; sstart[ns] = BM; }
        }
        __syncthreads();
        const int c4 = (tid & 7) * 4, slot = tid >> 3;                          // 8 threads x 4 columns per segment, 32 slots
        const int col0 = tn * BN + c4;
        if (col0 >= p.N) return;
        const int nseg = snseg;
        float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + col0;
        const bool vec_ok = (reinterpret_cast<uintptr_t>(p.Y) & 15) == 0 && (p.ldy & 3) == 0 && col0 + 4 <= p.N;
        for (int k = slot; k < nseg; k += 32) {
            const int rs = sstart[k], re = sstart[k + 1];
            const int sg = sseg[rs];
            if (sg < 0) continue;                                               // rows past the last edge
            f32x4 m = *reinterpret_cast<const f32x4*>(Z + rs * ZL + c4);
            for (int r = rs + 1; r < re; r += 4) {          // four independent reads in flight (rows past the end repeat the last)
                f32x4 z[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) z[u] = *reinterpret_cast<const f32x4*>(Z + min(r + u, re - 1) * ZL + c4);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], z[u][q]);
            }
            float* o = obase + (size_t)sg * p.ldy;
            const bool partial = (rs == 0 && first_cont) || (re == BM && last_cont);
            if (partial) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (col0 + q < p.N) atomic_max_f32(o + q, m[q]);
            } else if (vec_ok) {
                *reinterpret_cast<f32x4*>(o) = m;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (col0 + q < p.N) o[q] = m[q];
            }
        }
        return;
    }
    constexpr int NTP = ((ZC / 32) / WN) >= 1 ? ((ZC / 32) / WN) : 1;    // fragments per wave per pass
    constexpr int NPASS = NT / NTP;
    static_assert(NTP * WN * 32 == ZC && NPASS * NTP == NT, "epilogue column blocking");
    constexpr int G = 256 / ZC;                 // row groups in the reduce phase
    constexpr int RG = BM / G;                  // 32 (ZC = 64) or 16 (ZC = 32)
    const int zc = tid % ZC, zg = tid / ZC;
    const int r0 = (ZC == 64) ? __builtin_amdgcn_readfirstlane(zg * RG) : zg * RG;
    // MODE_EDGEMAX: segment starts of the tile's 128 rows as two ballots, once for all column passes
    unsigned long long seg_start_lo = 0, seg_start_hi = 0, seg_own = 0; int seg_id_own = 0;
    if constexpr (MODE == MODE_EDGEMAX && ZC == 64) {
        const int s_lo = sseg[lane], s_hi = sseg[lane + 64];
        const int p_lo = sseg[lane > 0 ? lane - 1 : 0], p_hi = sseg[lane + 63];
        seg_start_lo = __ballot(lane == 0 || s_lo != p_lo); seg_start_hi = __ballot(s_hi != p_hi);
        const bool up = r0 >= 64;
        const unsigned long long valid = up ? __ballot(s_hi >= 0) : __ballot(s_lo >= 0);
        seg_own = (up ? seg_start_hi : seg_start_lo) & valid & (((1ull << RG) - 1ull) << (r0 & 63));
        seg_id_own = up ? s_hi : s_lo;
    }
    // one instantiation per column pass: `cb` must be a compile-time constant so that acc[][] keeps
    // static register indices (a runtime-indexed accumulator array would live in scratch memory)
    auto run_pass = [&](auto cb_const) {
        constexpr int cb = decltype(cb_const)::value;
        __syncthreads();                        // main-loop reads / previous pass's scan done
#pragma unroll
        for (int j = 0; j < NTP; ++j) {
            constexpr int dummy = 0; (void)dummy;
            const int nt = cb * NTP + j;
            const int col = tn * BN + wn * NT * 32 + nt * 32 + l31;
            const float b = (p.bias && col < p.N) ? p.bias[col] : 0.f;
            const float sc = (p.scale && col < p.N) ? p.scale[col] : 1.f;
            const float sh = (p.shift && col < p.N) ? p.shift[col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * MT * 32 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    float v = acc[mt][nt][r] + b;
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    Z[rl * ZLD + (wn * NTP + j) * 32 + l31] = v * sc + sh;
                }
        }
        __syncthreads();
        const int wz = zc / (NTP * 32), jz = (zc >> 5) % NTP;
        const int col = tn * BN + wz * NT * 32 + (cb * NTP + jz) * 32 + (zc & 31);
        if (col >= p.N) return;
        const float* zcolp = Z + zc;
        if (MODE == MODE_POOL) {
            // every output is shared with other tiles -> all atomic; each group reduces its own rows
            int cur = -2; bool open = false; float m = 0.f;
#pragma unroll
            for (int bt = 0; bt < RG / 16; ++bt) {
                const int rb0 = r0 + bt * 16;
                float zv[16]; int sv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sv[i] = (ZC == 64) ? __builtin_amdgcn_readfirstlane(sseg[rb0 + i]) : sseg[rb0 + i];
                    zv[i] = zcolp[(rb0 + i) * ZLD];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (sv[i] != cur) {
                        if (open) atomic_max_f32(p.Y + (size_t)cur * p.ldy + col, m);
                        cur = sv[i]; open = cur >= 0; m = zv[i];
                    } else if (open) m = fmaxf(m, zv[i]);
                }
            }
            if (open) atomic_max_f32(p.Y + (size_t)cur * p.ldy + col, m);
        } else {
            // a wave (zg = wave id: ZC == 64 here) owns the segments that START in its row group and follows them to their
            // end. Segment bookkeeping by ballots (lane l looks at rows l and l + 64), then a walk over the set bits with four
            // LDS reads in flight per step -- not a per-row state machine (edge_pp.hip, same scheme).
            if constexpr (ZC == 64) {                          // (BN = 32 takes the narrow path above; its pooled variant the branch above)
            float* obase = p.Y + (size_t)rep * p.rep_out * p.ldy + col;
            const bool upper = r0 >= 64;
            unsigned long long mine = seg_own;
            while (mine) {                                                   // wave-uniform
                const int bl = __builtin_ctzll(mine);
                mine &= mine - 1ull;
                const unsigned long long above = bl < 63 ? ~((2ull << bl) - 1ull) : 0ull;
                int rs, re;
                if (upper) {
                    const unsigned long long later = seg_start_hi & above;
                    rs = 64 + bl; re = later ? 64 + __builtin_ctzll(later) : BM;
                } else {
                    const unsigned long long later = seg_start_lo & above;
                    rs = bl; re = later ? __builtin_ctzll(later) : (seg_start_hi ? 64 + __builtin_ctzll(seg_start_hi) : BM);
                }
                const int sg = __builtin_amdgcn_readlane(seg_id_own, bl);
                float m = zcolp[rs * ZLD];
                for (int r = rs + 1; r < re; r += 4) {
                    const int l = re - 1;
                    const float z0 = zcolp[r * ZLD], z1 = zcolp[min(r + 1, l) * ZLD], z2 = zcolp[min(r + 2, l) * ZLD], z3 = zcolp[min(r + 3, l) * ZLD];
                    m = fmaxf(fmaxf(m, z0), fmaxf(fmaxf(z1, z2), z3));
                }
                float* o = obase + (size_t)sg * p.ldy;
                if ((rs == 0 && first_cont) || (re == BM && last_cont)) atomic_max_f32(o, m); else *o = m;
            }
            }
        }
    };
    run_pass(std::integral_constant<int, 0>{});
    if constexpr (NPASS > 1) run_pass(std::integral_constant<int, 1>{});
    if constexpr (NPASS > 2) run_pass(std::integral_constant<int, 2>{});
    if constexpr (NPASS > 3) run_pass(std::integral_constant<int, 3>{});
    static_assert(NPASS <= 4, "column passes");
}

// Rows of `out` whose segment straddles a tile boundary are combined with integer-atomic float max
// by the two (or more) tiles involved: only THOSE rows need the identity pattern 0xFFFFFFFF beforehand.
// [r05] a LANE per boundary for the index chain (rowptr[n] -> dstS[e] -> rowptr[d]: three dependent loads, which one wave per
// boundary paid 330 k times per launch on the geo graph), then the wave walks the rows its lanes found and stores them 16 bytes per
// lane ([r04] had one wave per boundary: 17 us per launch on average, 0.3 ms per step).
// Tiles are numbered over the replicas, t = replica * tpr + tile (tpr = ceil(E' / tile_rows), E' read on the device); a persistent
// kernel that carries an open segment from tile to tile inside RUNS of `run` consecutive tiles (edge_rl.hip / edge_ws.hip) shares rows
// only where a run ends: candidate g = 1, 2, ... is the boundary in front of tile t = g * run (run = 1: every tile boundary).
// -> the output row (replica * rep_out + destination) whose segment straddles that boundary, or -1.
__device__ __forceinline__ int straddled_row(const int* __restrict__ rowptr, const int* __restrict__ dstS, int n_nodes, int tile_rows,
                                             int run, int replicas, int rep_out, int g, int n_cand, bool last_only) {
    if (g > n_cand) return -1;
    const int Etot = rowptr[n_nodes];
    const int tpr = (Etot + tile_rows - 1) / tile_rows;
    if (tpr <= 0) return -1;
    const long long t = (long long)g * run;
    const int rep = (int)(t / tpr), tile = (int)(t - (long long)rep * tpr);
    if (rep >= replicas || tile == 0) return -1;                      // a replica starts here: nothing above it
    const int e = tile * tile_rows;                                   // < Etot
    const int d = dstS[e];
    if (rowptr[d] >= e) return -1;                                    // a segment starts exactly on the boundary: no sharing
    // several shared boundaries inside one long segment: the LAST one stands for the row (the next one of this replica is run tiles on)
    if (last_only && (long long)e + (long long)run * tile_rows < rowptr[d + 1]) return -1;
    return rep * rep_out + d;
}

// One boundary pass may serve TWO launches [r06]: the template-graph and the geodesic-graph EdgeConv of a unit write disjoint column
// blocks of the same rows, their passes are independent of each other's kernels, and at one mesh per forward a pass is a launch plus one
// dependent index chain (7-9 us, 26 of them = 12 % of the forward): blockIdx.y selects the job.
struct BoundaryJob { const int* rowptr; const int* dstS; int n_nodes, H; float* out; int ldo, rep_out, tile_rows, run, replicas, n_cand; int* ovf; };
struct BoundaryJobs { BoundaryJob j[2]; };

__global__ __launch_bounds__(256) void init_boundary_rows_kernel(const BoundaryJobs js) {
    const BoundaryJob& b = js.j[blockIdx.y];
    if ((int)blockIdx.x * 256 >= b.n_cand) return;
    const int* __restrict__ rowptr = b.rowptr; const int* __restrict__ dstS = b.dstS;
    float* __restrict__ out = b.out;
    const int H = b.H, ldo = b.ldo;
    const int lane = threadIdx.x & 63;
    const int d = straddled_row(rowptr, dstS, b.n_nodes, b.tile_rows, b.run, b.replicas, b.rep_out, blockIdx.x * 256 + threadIdx.x + 1, b.n_cand, false);
    const bool vec = ((reinterpret_cast<uintptr_t>(out) | (uintptr_t)(ldo * 4) | (uintptr_t)(H * 4)) & 15) == 0;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned long long todo = __ballot(d >= 0);
    while (todo) {                                                    // wave-uniform
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        float* orow = out + (size_t)__builtin_amdgcn_readlane(d, l) * ldo;
        if (vec) {
            u32x4* o4 = reinterpret_cast<u32x4*>(orow);
            for (int c = lane; c < H / 4; c += 64) o4[c] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        } else {
            unsigned* o = reinterpret_cast<unsigned*>(orow);
            for (int c = lane; c < H; c += 64) o[c] = 0xFFFFFFFFu;
        }
    }
}

static BoundaryJob boundary_job(const int* rowptr, const int* dstS, int n_nodes, int edge_capacity, int H, float* out, int ldo, int rep_out,
                                int slots, int tile_rows, int run, int* ovf) {
    BoundaryJob b = {};
    b.rowptr = rowptr; b.dstS = dstS; b.n_nodes = n_nodes; b.H = H; b.out = out; b.ldo = ldo; b.rep_out = rep_out;
    b.tile_rows = tile_rows; b.run = run; b.replicas = slots; b.ovf = ovf;
    const int n_cand = cdiv((long)cdiv(edge_capacity, tile_rows) * slots, run) - 1;      // upper bound (capacity >= E')
    b.n_cand = n_cand > 0 ? n_cand : 0;
    return b;
}

static int init_boundary_rows(const BoundaryJob* jobs, int n, hipStream_t s) {
    BoundaryJobs js = {};
    int m = 0, most = 0;
    for (int i = 0; i < n; ++i) if (jobs[i].n_cand > 0) { js.j[m++] = jobs[i]; most = jobs[i].n_cand > most ? jobs[i].n_cand : most; }
    if (m == 0) return MORIG_OK;
    hipLaunchKernelGGL(init_boundary_rows_kernel, dim3(cdiv(most, 256), m), dim3(256), 0, s, js);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

static int init_boundary_rows(const int* rowptr, const int* dstS, int n_nodes, int edge_capacity, int H, float* out, int ldo,
                              int rep_out, int slots, hipStream_t s, int tile_rows = 128, int run = 1) {
    const BoundaryJob b = boundary_job(rowptr, dstS, n_nodes, edge_capacity, H, out, ldo, rep_out, slots, tile_rows, run, nullptr);
    return init_boundary_rows(&b, 1, s);
}

// out_split launches (edge_rl.hip / edge_ws.hip): whole segments leave the kernel as split-fp16 halves, the shared rows above as fp32
// atomics -- the only rows still in fp32 afterwards. This pass rewrites exactly those rows in place (a 32-column chunk occupies the
// same 128 bytes in both layouts: a lane reads 4 adjacent columns, then stores their 4 hi and 4 lo halves over the chunk; a wave
// instruction covers whole chunks, so every read of a chunk precedes every write to it). A segment that straddles several shared
// boundaries is converted at the LAST one (once). Same lane-per-candidate index phase as the pass above; the rows are taken four at a
// time so that four loads are in flight.
__global__ __launch_bounds__(256) void split_boundary_rows_kernel(const BoundaryJobs js) {
    const BoundaryJob& b = js.j[blockIdx.y];
    if ((int)blockIdx.x * 64 >= b.n_cand) return;
    const int* __restrict__ rowptr = b.rowptr; const int* __restrict__ dstS = b.dstS;
    float* __restrict__ out = b.out;
    int* __restrict__ ovf = b.ovf;
    const int H = b.H, ldo = b.ldo;
    // 16 candidates per wave (lanes 0..15 run the index chain): a wave converts its rows one group of four after the other, a load
    // round trip each -- with 64 candidates per wave the launch was one long latency chain (24 us on average for 5 k rows)
    const int lane = threadIdx.x & 63;
    const int g = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + lane + 1;
    const int d = lane < 16 ? straddled_row(rowptr, dstS, b.n_nodes, b.tile_rows, b.run, b.replicas, b.rep_out, g, b.n_cand, true) : -1;
    unsigned long long todo = __ballot(d >= 0);
    float am = 0.f;
    bool bad = false;                                                 // a NaN: fmaxf drops NaN operands, so the amax below never sees one
    while (todo) {                                                    // wave-uniform
        int dr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int l = todo ? __builtin_ctzll(todo) : -1;
            todo &= todo - 1;                                         // (0 stays 0)
            dr[i] = l >= 0 ? __builtin_amdgcn_readlane(d, l) : -1;
        }
        for (int c0 = 0; c0 < H; c0 += 256) {
            const int c = c0 + 4 * lane;
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (dr[i] >= 0 && c < H) v[i] = *reinterpret_cast<const float4*>(out + (size_t)dr[i] * ldo + c);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (dr[i] >= 0 && c < H) {
                    float h0, l0, h1, l1;
                    split_pair_f16(v[i].x, v[i].y, h0, l0);
                    split_pair_f16(v[i].z, v[i].w, h1, l1);
                    am = fmaxf(am, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
                    bad |= !(v[i].x == v[i].x && v[i].y == v[i].y && v[i].z == v[i].z && v[i].w == v[i].w);
                    char* oc = reinterpret_cast<char*>(out + (size_t)dr[i] * ldo) + (c >> 5) * 128 + (c & 31) * 2;
                    *reinterpret_cast<float2*>(oc) = make_float2(h0, h1);
                    *reinterpret_cast<float2*>(oc + 64) = make_float2(l0, l1);
                }
        }
    }
    // range guard on the converted rows; a NaN -- a result, or a shared row still at the identity pattern 0xFFFFFFFF because no atomic
    // reached it (would be a bug: both tiles of a boundary write the row) -- is raised by its own test (fmaxf hides it)
    if (bad || !(am < 65000.f)) *ovf = 1;
}

static int split_boundary_rows(const BoundaryJob* jobs, int n, hipStream_t s) {
    BoundaryJobs js = {};
    int m = 0, most = 0;
    for (int i = 0; i < n; ++i) if (jobs[i].n_cand > 0) { js.j[m++] = jobs[i]; most = jobs[i].n_cand > most ? jobs[i].n_cand : most; }
    if (m == 0) return MORIG_OK;
    hipLaunchKernelGGL(split_boundary_rows_kernel, dim3(cdiv(most, 64), m), dim3(256), 0, s, js);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

// ------------------------------------------------------------------------------------------------
template <int BN, int KC, int LOAD, int MODE, int PREC = PREC_F32>
static int launch_tile(const TileParams& p0, int nblocks, hipStream_t s) {
    TileParams p = p0;
    p.dbg = debug_flags();
    if constexpr (BN == 128 && KC == 32 && LOAD == LOAD_DENSE && MODE == MODE_STORE) {
        static const bool no_fast = getenv("MORIG_TILE_NO_FAST") != nullptr;
        if (p.dbg == 0 && !no_fast && p.M % 128 == 0 && p.K % KC == 0) {
            hipLaunchKernelGGL((tile_kernel<BN, KC, LOAD, MODE, PREC, true>), dim3(nblocks), dim3(256), 0, s, p);
            MORIG_LAUNCH_CHECK();
            return MORIG_OK;
        }
    }
    hipLaunchKernelGGL((tile_kernel<BN, KC, LOAD, MODE, PREC>), dim3(nblocks), dim3(256), 0, s, p);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// scatter_max semantics for segments that received no row (a graph id without vertices): torch_scatter fills 0; the
// integer-atomic max leaves its identity pattern 0xFFFFFFFF (a NaN) there, which would then flag every split of the row
__global__ void pool_identity_to_zero_kernel(float* __restrict__ pool, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned* u = reinterpret_cast<unsigned*>(pool);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) if (u[i] == 0xFFFFFFFFu) u[i] = 0u;
}
static int pool_finalize(float* pool, int n_seg, int ld_pool, hipStream_t s) {
    const int64_t n = (int64_t)n_seg * ld_pool;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pool_identity_to_zero_kernel, dim3((int)blocks), dim3(256), 0, s, pool, n);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig

using namespace morig;

extern "C" int morig_gemm(const morig_gemm_args* a_in, void* stream) {
    morig_gemm_args mine;
    if (!take_args(a_in, mine, MORIG_GEMM_ARGS_V3_SIZE)) return MORIG_E_INVALID;
    const morig_gemm_args* a = &mine;
    if (!a->X || !a->W) return MORIG_E_INVALID;
    if (a->M < 0 || a->N <= 0 || a->K <= 0) return MORIG_E_INVALID;
    if (a->M == 0) return MORIG_OK;
    if ((a->ldx & 3) || (a->ldw & 3) || !aligned16(a->X) || !aligned16(a->W)) return MORIG_E_INVALID;
    const bool tail = a->X_tail != nullptr;
    if (tail) {
        // K tail: split chunks from X_tail[row % tail_rows] (include/morig_hip.h); everything about it is checked here, the kernel choice below
        if (!a->x_split || !a->W_split || a->tail_cols <= 0 || (a->tail_cols & 31) || (a->K & 31) || a->K <= a->tail_cols || a->tail_rows <= 0 ||
            (a->ld_tail & 31) || a->ld_tail < a->tail_cols || (reinterpret_cast<uintptr_t>(a->X_tail) & 127) ||
            (double)a->tail_rows * a->ld_tail * 4.0 >= 4.0e9) return MORIG_E_INVALID;
    }
    if (a->ldx < (((a->K - (tail ? a->tail_cols : 0)) + 3) & ~3) || a->ldw < ((a->K + 31) & ~31)) return MORIG_E_INVALID;
    const bool pool = a->pool != nullptr;
    if (tail && pool) return MORIG_E_UNSUPPORTED;
    if (!pool && !a->Y) return MORIG_E_INVALID;
    if (pool && a->Y) return MORIG_E_UNSUPPORTED;           // one consumer per launch
    if ((pool || a->rowbias) && !a->seg) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);

    TileParams p = {};
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.W = a->W; p.ldw = a->ldw;
    p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.relu = a->relu;
    p.X = a->X; p.ldx = a->ldx;
    p.rowbias = a->rowbias; p.ld_rowbias = a->ld_rowbias; p.seg = a->seg;
    const int tiles_m = cdiv(a->M, 128);
    const double flops = 2.0 * a->M * (double)a->N * a->K;
    const double bytes = 4.0 * ((double)a->M * a->K + (double)a->N * a->K + (pool ? 0.0 : (double)a->M * a->N));

    // [r06] MORIG_SPLIT_BF16X6 without an image: the exact path on three bf16 limbs per operand (split in the kernel)
    const bool x6 = a->W_split == nullptr && a->w_split_format == MORIG_SPLIT_BF16X6;
    if (x6 && (a->x_split || a->y_split)) return MORIG_E_INVALID;
    const bool f16 = a->W_split != nullptr;
    // the bf16 split (W_split holds bf16 halves; bf16 has float32's exponent range, so there is no range guard to report through):
    // fp32 X, fp32 Y, plain stores -- the backward contractions. Selected by w_split_format, never inferred from a missing overflow
    // word: a caller that forgets the word on an fp16 image gets MORIG_E_INVALID, not silently unguarded results
    if (f16 && a->w_split_format != MORIG_SPLIT_F16 && a->w_split_format != MORIG_SPLIT_BF16) return MORIG_E_INVALID;
    const bool bf16 = f16 && a->w_split_format == MORIG_SPLIT_BF16;
    if (f16 && !bf16 && a->overflow == nullptr) return MORIG_E_INVALID;
    if (f16) {
        if (!aligned16(a->W_split)) return MORIG_E_INVALID;
        if (bf16 && (pool || a->x_split || a->y_split)) return MORIG_E_UNSUPPORTED;
        p.W = static_cast<const float*>(a->W_split); p.ovf = a->overflow;
    }
    // 64-deep K chunks halve the barriers per MFMA on the split-fp16 path (needs weights padded to 64 in K)
    const bool kc64 = f16 && a->K >= 64 && (a->ldw & 63) == 0 && a->ldw >= ((a->K + 63) & ~63) && !getenv("MORIG_KC32");
    if (a->x_split || a->y_split) {
        if (!f16) return MORIG_E_INVALID;                        // split activations only exist on the split-fp16 path
        if (a->x_split && ((a->ldx & 31) || a->ldx < (((a->K - (tail ? a->tail_cols : 0)) + 31) & ~31) || (reinterpret_cast<uintptr_t>(a->X) & 127))) return MORIG_E_INVALID;
        if (a->y_split && (pool || (a->ldy & 31) || (reinterpret_cast<uintptr_t>(a->Y) & 127))) return MORIG_E_INVALID;
        p.x16 = a->x_split ? 1 : 0; p.y16 = a->y_split ? 1 : 0;
    }
    if (pool) {
        if (a->n_seg <= 0 || a->ld_pool < a->N) return MORIG_E_INVALID;
        // identity of the integer-atomic float max
        MORIG_HIP_TRY(hipMemsetAsync(a->pool, 0xFF, (size_t)a->n_seg * a->ld_pool * sizeof(float), s));
        if (f16 && a->x_split && a->N > 64 && !getenv("MORIG_NO_DMA")) {
            // requires `seg` sorted within the matrix (PyG batch vectors are): see gemm_dma.hip
            GemmDmaParams q = {};
            q.M = a->M; q.N = a->N; q.K = a->K; q.X = a->X; q.ldx = a->ldx; q.W = p.W; q.ldw = p.ldw;
            q.bias = a->bias; q.scale = a->scale; q.shift = a->shift; q.relu = a->relu; q.seg = a->seg;
            q.pool = a->pool; q.ld_pool = a->ld_pool; q.ovf = a->overflow;
            ProfScope ps(K_GEMM16_POOL, s, flops, bytes);
            const int st = launch_gemm16_dma(q, tiles_m, s);
            return st != MORIG_OK ? st : pool_finalize(a->pool, a->n_seg, a->ld_pool, s);
        }
        p.Y = a->pool; p.ldy = a->ld_pool;
        p.tiles_n = cdiv(a->N, 128);
        ProfScope ps(f16 ? K_GEMM16_POOL : K_GEMM_POOL, s, flops, bytes);
        int st;
        if (x6) st = launch_tile<128, 32, LOAD_DENSE, MODE_POOL, PREC_BF16X6>(p, tiles_m * p.tiles_n, s);
        else if (f16 && kc64 && getenv("MORIG_KC64")) st = launch_tile<128, 64, LOAD_DENSE, MODE_POOL, PREC_F16X3>(p, tiles_m * p.tiles_n, s);
        else st = f16 ? launch_tile<128, 32, LOAD_DENSE, MODE_POOL, PREC_F16X3>(p, tiles_m * p.tiles_n, s)
                      : launch_tile<128, 32, LOAD_DENSE, MODE_POOL>(p, tiles_m * p.tiles_n, s);
        return st != MORIG_OK ? st : pool_finalize(a->pool, a->n_seg, a->ld_pool, s);
    }
    p.Y = a->Y; p.ldy = a->ldy;
    if (f16 && a->x_split && a->N > 64 && !getenv("MORIG_NO_DMA")) {
        // both operands pre-split: LDS-DMA pipeline (gemm_dma.hip)
        GemmDmaParams q = {};
        q.M = a->M; q.N = a->N; q.K = a->K; q.X = a->X; q.ldx = a->ldx; q.W = p.W; q.ldw = p.ldw;
        q.bias = a->bias; q.scale = a->scale; q.shift = a->shift; q.relu = a->relu;
        q.rowbias = a->rowbias; q.ld_rowbias = a->ld_rowbias; q.seg = a->seg;
        q.Y = a->Y; q.ldy = a->ldy; q.y16 = a->y_split ? 1 : 0; q.tiles_n = cdiv(a->N, 128); q.ovf = a->overflow;
        if (tail) { q.Xt = a->X_tail; q.ldt = a->ld_tail; q.tail_rows = a->tail_rows; q.tail_chunks = a->tail_cols / 32; }
        ProfScope ps(K_GEMM16_DMA, s, flops, bytes);
        return launch_gemm16_dma(q, tiles_m, s);
    }
    if (tail) return MORIG_E_UNSUPPORTED;
    // [r06] a few rows (per-mesh vectors): weights spread over the chip, plain fp32 FMAs (vertex_ops.hip); a->W is the fp32 matrix on
    // every path, so the fast and the exact mode run the same kernel here
    if (!bf16 && !a->rowbias && !a->x_split && !a->y_split && few_rows_gemm_takes(a->M, a->N, a->K, a->ldx, a->ldw)) {
        ProfScope ps(K_MISC, s, flops, bytes);
        return launch_few_rows_gemm(a->X, a->ldx, a->M, a->W, a->ldw, a->N, a->K, a->bias, a->scale, a->shift, a->relu, a->Y, a->ldy, s);
    }
    if (bf16) {
        ProfScope ps(K_MISC, s, flops, bytes);
        if (a->N > 64) { p.tiles_n = cdiv(a->N, 128); return launch_tile<128, 32, LOAD_DENSE, MODE_STORE, PREC_BF16X3>(p, tiles_m * p.tiles_n, s); }
        p.tiles_n = 1;
        if (a->N > 32) return launch_tile<64, 32, LOAD_DENSE, MODE_STORE, PREC_BF16X3>(p, tiles_m, s);
        return launch_tile<32, 32, LOAD_DENSE, MODE_STORE, PREC_BF16X3>(p, tiles_m, s);
    }
    if (x6) {
        ProfScope ps(a->N > 64 ? K_GEMM_BN128 : (a->N > 32 ? K_GEMM_BN64 : K_GEMM_BN32), s, flops, bytes);
        if (a->N > 64) { p.tiles_n = cdiv(a->N, 128); return launch_tile<128, 32, LOAD_DENSE, MODE_STORE, PREC_BF16X6>(p, tiles_m * p.tiles_n, s); }
        p.tiles_n = 1;
        if (a->N > 32) return launch_tile<64, 32, LOAD_DENSE, MODE_STORE, PREC_BF16X6>(p, tiles_m, s);
        return launch_tile<32, 32, LOAD_DENSE, MODE_STORE, PREC_BF16X6>(p, tiles_m, s);
    }
    if (a->N > 64) {
        p.tiles_n = cdiv(a->N, 128);
        ProfScope ps(f16 ? K_GEMM16_BN128 : K_GEMM_BN128, s, flops, bytes);
        // fp32 X on the split-fp16 path: the 32-deep variant at 3 workgroups per CU (152 VGPRs, 37 KB LDS) beats the 64-deep one
        // at 2 per CU by ~20 % (measured; 4 per CU spills). MORIG_KC64 restores the old choice.
        static const bool want64 = getenv("MORIG_KC64") != nullptr;
        if (f16 && kc64 && want64) return launch_tile<128, 64, LOAD_DENSE, MODE_STORE, PREC_F16X3>(p, tiles_m * p.tiles_n, s);
        return f16 ? launch_tile<128, 32, LOAD_DENSE, MODE_STORE, PREC_F16X3>(p, tiles_m * p.tiles_n, s)
                   : launch_tile<128, 32, LOAD_DENSE, MODE_STORE>(p, tiles_m * p.tiles_n, s);
    } else if (a->N > 32) {
        p.tiles_n = 1;
        ProfScope ps(f16 ? K_GEMM16_BN64 : K_GEMM_BN64, s, flops, bytes);
        return f16 ? launch_tile<64, 32, LOAD_DENSE, MODE_STORE, PREC_F16X3>(p, tiles_m, s)
                   : launch_tile<64, 32, LOAD_DENSE, MODE_STORE>(p, tiles_m, s);
    } else {
        p.tiles_n = 1;
        ProfScope ps(f16 ? K_GEMM16_BN32 : K_GEMM_BN32, s, flops, bytes);
        return f16 ? launch_tile<32, 32, LOAD_DENSE, MODE_STORE, PREC_F16X3>(p, tiles_m, s)
                   : launch_tile<32, 32, LOAD_DENSE, MODE_STORE>(p, tiles_m, s);
    }
}

static int edge_common(const morig_edgeconv_args* a, TileParams& p) {
    if (!a->A || !a->B || !a->rowptr || !a->src_sorted || !a->dst_sorted || !a->W2 || !a->out) return MORIG_E_INVALID;
    if ((a->s1 == nullptr) != (a->t1 == nullptr) || !a->b2 || !a->s2 || !a->t2) return MORIG_E_INVALID;
    if (a->n_nodes <= 0 || a->replicas <= 0 || a->edge_capacity <= 0) return MORIG_E_INVALID;
    if ((a->lda & 3) || (a->ldb & 3) || (a->ldw & 3) || !aligned16(a->A) || !aligned16(a->B) || !aligned16(a->W2) ||
        !aligned16(a->s1) || !aligned16(a->t1)) return MORIG_E_INVALID;
    if (a->ldo < a->H || a->lda < a->H || a->ldb < a->H || a->ldw < a->H) return MORIG_E_INVALID;
    p.M = a->edge_capacity; p.N = a->H; p.K = a->H;
    p.W = a->W2; p.ldw = a->ldw;
    p.bias = a->b2; p.scale = a->s2; p.shift = a->t2; p.relu = 1;
    p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb;
    p.rowptr = a->rowptr; p.srcS = a->src_sorted; p.dstS = a->dst_sorted;
    p.n_nodes = a->n_nodes; p.rep_in = a->in_rep_stride; p.rep_out = a->out_rep_stride;
    p.s1 = a->s1; p.t1 = a->t1;
    p.Y = a->out; p.ldy = a->ldo;
    p.tiles_per_rep = cdiv(a->edge_capacity, 128);
    p.tiles_n = 1;
    if (a->W2_split) {
        if (!a->overflow || !aligned16(a->W2_split) || a->H < 32) return MORIG_E_INVALID;
        p.W = static_cast<const float*>(a->W2_split); p.ovf = a->overflow;
    }
    return MORIG_OK;
}

extern "C" int morig_edge_hidden(const morig_edgeconv_args* a_in, void* stream) {
    morig_edgeconv_args mine;
    if (!take_args(a_in, mine, MORIG_EDGECONV_ARGS_V3_SIZE)) return MORIG_E_INVALID;
    const morig_edgeconv_args* a = &mine;
    TileParams p = {};
    const int st = edge_common(a, p);
    if (st != MORIG_OK) return st;
    if (a->replicas != 1 || a->out_split) return MORIG_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const double E = (double)(a->edge_count > 0 ? a->edge_count : a->edge_capacity);
    const double flops = 2.0 * E * a->H * (double)a->H;
    const bool f16 = a->W2_split != nullptr;
    ProfScope ps(f16 ? K_POINTCONV16 : K_POINTCONV, s, flops, 4.0 * 3.0 * E * a->H);
    if (!f16 && a->exact_arith == 1) {            // the exact path on three bf16 limbs (MORIG_SPLIT_BF16X6)
        switch (a->H) {
            case 32:  return launch_tile<32, 32, LOAD_EDGE, MODE_STORE, PREC_BF16X6>(p, p.tiles_per_rep, s);
            case 64:  return launch_tile<64, 32, LOAD_EDGE, MODE_STORE, PREC_BF16X6>(p, p.tiles_per_rep, s);
            case 128: return launch_tile<128, 32, LOAD_EDGE, MODE_STORE, PREC_BF16X6>(p, p.tiles_per_rep, s);
            case 256: return launch_tile<256, 32, LOAD_EDGE, MODE_STORE, PREC_BF16X6>(p, p.tiles_per_rep, s);
            default: break;
        }
    }
    switch (a->H) {
        case 16:  return launch_tile<32, 16, LOAD_EDGE, MODE_STORE>(p, p.tiles_per_rep, s);     // fp32 MFMA (no split image below 32)
        case 32:  return f16 ? launch_tile<32, 32, LOAD_EDGE, MODE_STORE, PREC_F16X3>(p, p.tiles_per_rep, s)
                             : launch_tile<32, 32, LOAD_EDGE, MODE_STORE>(p, p.tiles_per_rep, s);
        case 64:  return f16 ? launch_tile<64, 32, LOAD_EDGE, MODE_STORE, PREC_F16X3>(p, p.tiles_per_rep, s)
                             : launch_tile<64, 32, LOAD_EDGE, MODE_STORE>(p, p.tiles_per_rep, s);
        case 128: return f16 ? launch_tile<128, 32, LOAD_EDGE, MODE_STORE, PREC_F16X3>(p, p.tiles_per_rep, s)
                             : launch_tile<128, 32, LOAD_EDGE, MODE_STORE>(p, p.tiles_per_rep, s);
        case 256: return f16 ? launch_tile<256, 32, LOAD_EDGE, MODE_STORE, PREC_F16X3>(p, p.tiles_per_rep, s)
                             : launch_tile<256, 16, LOAD_EDGE, MODE_STORE>(p, p.tiles_per_rep, s);
        default: return MORIG_E_UNSUPPORTED;
    }
}

extern "C" int morig_segmax_gemm(const morig_segmax_args* a_in, void* stream) {
    morig_segmax_args mine;
    if (!take_args(a_in, mine, MORIG_SEGMAX_ARGS_V3_SIZE)) return MORIG_E_INVALID;
    const morig_segmax_args* a = &mine;
    if (!a->X || !a->W || !a->rowptr || !a->dst_sorted || !a->out) return MORIG_E_INVALID;
    if (a->N <= 0 || a->K <= 0 || a->n_nodes <= 0 || a->edge_capacity <= 0) return MORIG_E_INVALID;
    if ((a->ldx & 3) || (a->ldw & 3) || !aligned16(a->X) || !aligned16(a->W)) return MORIG_E_INVALID;
    if (a->ldx < ((a->K + 3) & ~3) || a->ldw < ((a->K + 31) & ~31) || a->ldo < a->N) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    TileParams p = {};
    p.M = a->edge_capacity; p.N = a->N; p.K = a->K;
    p.W = a->W; p.ldw = a->ldw;
    p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.relu = a->relu;
    p.X = a->X; p.ldx = a->ldx;
    p.rowptr = a->rowptr; p.dstS = a->dst_sorted; p.n_nodes = a->n_nodes;
    p.Y = a->out; p.ldy = a->ldo;
    p.tiles_per_rep = cdiv(a->edge_capacity, 128);
    p.tiles_n = 1;
    { const int st = init_boundary_rows(a->rowptr, a->dst_sorted, a->n_nodes, a->edge_capacity, a->N, a->out, a->ldo, 0, 1, s);
      if (st != MORIG_OK) return st; }
    const double E = (double)(a->edge_count > 0 ? a->edge_count : a->edge_capacity);
    const bool f16 = a->W_split != nullptr;
    if (f16) {
        if (!a->overflow || !aligned16(a->W_split)) return MORIG_E_INVALID;
        p.W = static_cast<const float*>(a->W_split); p.ovf = a->overflow;
    }
    ProfScope ps(f16 ? K_POINTCONV16 : K_POINTCONV, s, 2.0 * E * a->N * (double)a->K, 4.0 * E * a->K);
    if (a->N <= 32)       return f16 ? launch_tile<32, 32, LOAD_DENSE, MODE_EDGEMAX, PREC_F16X3>(p, p.tiles_per_rep, s)
                                     : launch_tile<32, 32, LOAD_DENSE, MODE_EDGEMAX>(p, p.tiles_per_rep, s);
    else if (a->N <= 64)  return f16 ? launch_tile<64, 32, LOAD_DENSE, MODE_EDGEMAX, PREC_F16X3>(p, p.tiles_per_rep, s)
                                     : launch_tile<64, 32, LOAD_DENSE, MODE_EDGEMAX>(p, p.tiles_per_rep, s);
    else if (a->N <= 128) return f16 ? launch_tile<128, 32, LOAD_DENSE, MODE_EDGEMAX, PREC_F16X3>(p, p.tiles_per_rep, s)
                                     : launch_tile<128, 32, LOAD_DENSE, MODE_EDGEMAX>(p, p.tiles_per_rep, s);
    else if (a->N <= 256) return f16 ? launch_tile<256, 32, LOAD_DENSE, MODE_EDGEMAX, PREC_F16X3>(p, p.tiles_per_rep, s)
                                     : launch_tile<256, 16, LOAD_DENSE, MODE_EDGEMAX>(p, p.tiles_per_rep, s);
    return MORIG_E_UNSUPPORTED;
}

// which kernel a morig_edgeconv launch takes: H = 256 / 128 on the split-fp16 path go to the specialised kernels; with a 4-aligned CSR
// the W2-stationary one (edge_ws.hip), whose H = 256 tiles are 64 rows. MORIG_EDGE_KERNEL=pp|pc keeps the older kernels (A/B runs).
struct EdgePlan { bool wide, one_shot, pp_ok, use_rl, use_ws, mix, split_ok; int tile_rows, run; };
static EdgePlan edge_plan(const morig_edgeconv_args* a) {
    EdgePlan pl = {};
    const bool f16 = a->W2_split != nullptr;
    static const char* ek = getenv("MORIG_EDGE_KERNEL");
    static const bool one_shot = ek && ek[0] == 'p' && ek[1] == 'c';
    static const bool want_pp = ek && ek[0] == 'p' && ek[1] == 'p';
    pl.one_shot = one_shot;
    pl.wide = f16 && (a->H == 256 || a->H == 128) && a->s1 == nullptr && !getenv("MORIG_NO_EDGE_PC");
    // the persistent kernels store 16-byte result vectors and address gathered rows with 32-bit byte offsets
    pl.pp_ok = (reinterpret_cast<uintptr_t>(a->out) & 15) == 0 && (a->ldo & 3) == 0 &&
               (double)a->n_nodes * a->lda * 4.0 < 4.0e9 && (double)a->n_nodes * a->ldb * 4.0 < 4.0e9;
    // H = 128 has half the MFMA work per gathered byte: measured on par with / behind the producer-consumer kernel, which stays
    // the default there (MORIG_WS128=1 selects edge_ws.hip for it too)
    static const bool ws128 = [] { const char* e = getenv("MORIG_WS128"); return e && e[0] == '1'; }();
    // H = 128 with a 4-aligned CSR: the row-local kernel (edge_rl.hip: W2 resident in LDS, eight independent waves, no barrier in the
    // main loop; 64-row tiles). MORIG_RL128=0 keeps the producer-consumer kernel (A/B runs)
    static const bool rl128 = [] { const char* e = getenv("MORIG_RL128"); return !(e && e[0] == '0'); }();
    const bool quad_ok = pl.wide && !one_shot && !want_pp && pl.pp_ok && a->quad_aligned && (a->lda & 3) == 0 && (a->ldb & 3) == 0;
    pl.use_rl = quad_ok && a->H == 128 && rl128 && !ws128;
    // [r06] H = 256 on a MORIG_CSR_MIN4 CSR (segments of >= 4 rows, NOT 4-aligned): the mixed-quad form of the W2-stationary kernel, which
    // exists for split rows only (edge_ws.hip <256, true, true>). MORIG_EDGE_MIX=0: such a CSR takes the generic kernels (A/B runs)
    static const bool want_mix = [] { const char* e = getenv("MORIG_EDGE_MIX"); return !(e && e[0] == '0'); }();
    pl.mix = pl.wide && !one_shot && !want_pp && pl.pp_ok && !a->quad_aligned && a->seg_min4 && a->H == 256 && want_mix &&
             (a->lda & 3) == 0 && (a->ldb & 3) == 0;
    pl.use_ws = (quad_ok && !pl.use_rl && (a->H == 256 || ws128)) || (pl.mix && a->out_split);
    pl.tile_rows = ((pl.use_ws && a->H == 256) || pl.use_rl) ? 64 : 128;   // edge_ws.hip at H = 256, edge_rl.hip: 64-row tiles
    // the two kernels above carry a segment that is still open at the end of a tile into the next tile of the same RUN of consecutive
    // tiles (one wave / one workgroup works through a run): only rows that straddle a run boundary are shared through atomics.
    // MORIG_EDGE_RUN=<tiles> (1 = every tile boundary is shared, the [r04] behaviour)
    // Measured (profiles/r05n_*): runs of 4 .. 8 tiles gain 0.5 % of the step over 1, 16 and more lose to the coarser work split --
    // so at most 8, and short enough that every wave (edge_rl.hip, 8 per CU) / workgroup (edge_ws.hip) still gets >= 16 runs
    static const int run_env = [] { const char* e = getenv("MORIG_EDGE_RUN"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 4096 ? 4096 : v); }();
    pl.run = 1;
    if (pl.use_rl || (pl.use_ws && a->out_split)) {              // (edge_ws.hip carries segments in its split-rows form only)
        const long tiles = (long)cdiv(a->edge_capacity, pl.tile_rows) * a->replicas;
        const long units = pl.use_rl ? 2048 : 256;
        while (pl.run * 2 <= run_env && tiles / (units * 16) >= pl.run * 2) pl.run *= 2;
    }
    // split-fp16 results: the two kernels above, chunk-aligned output window (MORIG_EDGE_SPLIT_OUT=0: never, A/B runs)
    static const bool no_split = [] { const char* e = getenv("MORIG_EDGE_SPLIT_OUT"); return e && e[0] == '0'; }();
    pl.split_ok = (pl.use_rl || pl.use_ws || pl.mix) && !no_split && a->overflow != nullptr && (a->ldo & 31) == 0 &&
                  (reinterpret_cast<uintptr_t>(a->out) & 127) == 0;
    return pl;
}

extern "C" int morig_edgeconv_can_split_out(const morig_edgeconv_args* a_in) {
    morig_edgeconv_args mine;
    if (!take_args(a_in, mine, MORIG_EDGECONV_ARGS_V3_SIZE)) return 0;
    const morig_edgeconv_args* a = &mine;
    TileParams p = {};
    if (edge_common(a, p) != MORIG_OK) return 0;
    return edge_plan(a).split_ok ? 1 : 0;
}

extern "C" int morig_edgeconv(const morig_edgeconv_args* a_in, void* stream) {
    morig_edgeconv_args mine;
    if (!take_args(a_in, mine, MORIG_EDGECONV_ARGS_V3_SIZE)) return MORIG_E_INVALID;
    const morig_edgeconv_args* a = &mine;
    TileParams p = {};
    const int st = edge_common(a, p);
    if (st != MORIG_OK) return st;
    if (a->in_rep_stride < 0 || (a->replicas > 1 && a->out_rep_stride < a->n_nodes)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nblocks = p.tiles_per_rep * a->replicas;
    const bool f16 = a->W2_split != nullptr;
    const EdgePlan pl = edge_plan(a);
    const bool wide = pl.wide, one_shot = pl.one_shot, pp_ok = pl.pp_ok, use_rl = pl.use_rl, use_ws = pl.use_ws;
    if (a->out_split && !pl.split_ok) return MORIG_E_UNSUPPORTED;

    // tile-straddling target segments combine through integer-atomic float max: identity in exactly those rows -- of this launch and,
    // with init_with, of the partner launch that follows (which then comes with skip_init)
    auto job_of = [](const morig_edgeconv_args* q, const EdgePlan& qp) {
        return boundary_job(q->rowptr, q->dst_sorted, q->n_nodes, q->edge_capacity, q->H, q->out, q->ldo, q->out_rep_stride, q->replicas,
                            qp.tile_rows, qp.run, q->overflow);
    };
    // the partner's arguments, checked as its own call will check them (nothing of it is launched here but the pass over its rows)
    morig_edgeconv_args other;
    auto partner = [&](const morig_edgeconv_args* q_in, EdgePlan& qp) -> int {
        if (!take_args(q_in, other, MORIG_EDGECONV_ARGS_V3_SIZE)) return MORIG_E_INVALID;
        TileParams tp = {};
        const int stp = edge_common(&other, tp);
        if (stp != MORIG_OK) return stp;
        if (other.in_rep_stride < 0 || (other.replicas > 1 && other.out_rep_stride < other.n_nodes)) return MORIG_E_INVALID;
        qp = edge_plan(&other);
        if (other.out_split && !qp.split_ok) return MORIG_E_UNSUPPORTED;
        return MORIG_OK;
    };
    if (!a->skip_init) {
        BoundaryJob jobs[2];
        int nj = 0;
        jobs[nj++] = job_of(a, pl);
        if (a->init_with) {
            EdgePlan qp;
            const int stp = partner(a->init_with, qp);
            if (stp != MORIG_OK) return stp;
            jobs[nj++] = job_of(&other, qp);
        }
        const int st2 = init_boundary_rows(jobs, nj, s);
        if (st2 != MORIG_OK) return st2;
    } else if (a->init_with) return MORIG_E_INVALID;

    const double E = (double)(a->edge_count > 0 ? a->edge_count : a->edge_capacity) * a->replicas;
    const double flops = 2.0 * E * a->H * (double)a->H;
    // compulsory bytes: every row of the [A | B] operand tables and every result row ONCE (the per-edge re-reads of gathered rows are L2
    // hits: what they cost shows as traffic ABOVE this figure in bench.py's roofline.traffic_over_algorithmic)
    const double bytes = 4.0 * a->H * (double)a->n_nodes * (2.0 * (a->in_rep_stride > 0 ? a->replicas : 1) + a->replicas);
    // H = 256 / 128: wave-specialised kernel (1.35x / 1.25x over the symmetric one; H = 128 runs two workgroups per CU)
    if (wide) {
        EdgePcParams q = {};
        q.H = a->H; q.W = p.W; q.ldw = p.ldw; q.bias = p.bias; q.scale = p.scale; q.shift = p.shift;
        q.A = p.A; q.lda = p.lda; q.B = p.B; q.ldb = p.ldb;
        q.rowptr = p.rowptr; q.srcS = p.srcS; q.dstS = p.dstS; q.n_nodes = p.n_nodes;
        q.rep_in = p.rep_in; q.rep_out = p.rep_out; q.tiles_per_rep = p.tiles_per_rep; q.replicas = a->replicas;
        q.Y = p.Y; q.ldy = p.ldy; q.ovf = p.ovf; q.quad = a->quad_aligned ? 1 : 0; q.min4 = a->seg_min4 ? 1 : 0;
        q.y16 = a->out_split ? 1 : 0; q.run = pl.run;
        ProfScope ps(a->H == 256 ? K_EDGE16_H256 : K_EDGE16_H128, s, flops, bytes);
        if (use_rl || use_ws) {
            int st3;
            if (use_rl) {
                prof_retag(K_EDGE16_H128_RL);
                st3 = launch_edge_rl(q, cdiv(a->edge_capacity, 64) * a->replicas, s);
            } else {
                if (a->H == 128) prof_retag(K_EDGE16_H128_WS);
                st3 = launch_edge_ws(q, cdiv(a->edge_capacity, a->H == 256 ? 64 : 128) * a->replicas, s);
            }
            if (st3 != MORIG_OK || !a->out_split) return (st3 == MORIG_OK && (a->split_with || a->skip_split)) ? MORIG_E_INVALID : st3;
            // the rows two tiles share were combined as fp32 atomics: into the split layout now (skip_split: the partner's pass does it,
            // behind its own kernel; split_with: this pass also converts the partner's rows, whose kernel ran in front of this one)
            if (a->skip_split) return a->split_with ? MORIG_E_INVALID : MORIG_OK;
            BoundaryJob jobs[2];
            int nj = 0;
            jobs[nj++] = job_of(a, pl);
            if (a->split_with) {
                EdgePlan qp;
                const int stp = partner(a->split_with, qp);
                if (stp != MORIG_OK) return stp;
                if (!other.out_split || !(qp.use_rl || qp.use_ws)) return MORIG_E_INVALID;
                jobs[nj++] = job_of(&other, qp);
            }
            return split_boundary_rows(jobs, nj, s);
        }
        if (a->split_with || a->skip_split) return MORIG_E_INVALID;      // (only the split-rows kernels have a conversion pass)
        if (one_shot || !pp_ok) { prof_retag(K_EDGE16_PC); return launch_edge_pc(q, nblocks, s); }
        if (a->H == 256) prof_retag(K_EDGE16_H256_PP);
        return launch_edge_pp(q, nblocks, s);
    }
    if (a->split_with || a->skip_split) return MORIG_E_INVALID;
    // [r06] H = 32 on the split-fp16 path with the hidden BatchNorm folded (packing.fold_hidden_affine: every pack of the networks): the
    // persistent 32-wide kernel with gathered [A | B] rows (edge_x3.hip <true>); MORIG_X3_TILE=1 keeps the tile engine (A/B runs)
    static const bool x3_tile = [] { const char* e = getenv("MORIG_X3_TILE"); return e && e[0] == '1'; }();
    if (f16 && a->H == 32 && !a->s1 && !x3_tile && (a->ldo & 3) == 0 && aligned16(a->out) && (a->lda & 3) == 0 && (a->ldb & 3) == 0 &&
        aligned16(a->b2) && a->ldw >= 32) {
        EdgeX3Params q = {};
        q.A = a->A; q.lda = a->lda; q.B = a->B; q.ldb = a->ldb;
        q.W2s = static_cast<const float*>(a->W2_split); q.ldw = a->ldw;
        q.bias = a->b2; q.scale = a->s2; q.shift = a->t2;
        q.rowptr = a->rowptr; q.srcS = a->src_sorted; q.dstS = a->dst_sorted; q.n_nodes = a->n_nodes; q.cap = a->edge_capacity;
        q.rep_in = a->in_rep_stride; q.rep_out = a->out_rep_stride; q.replicas = a->replicas;
        q.Y = a->out; q.ldy = a->ldo; q.ovf = a->overflow;
        ProfScope ps(K_EDGE16_H32, s, flops, bytes);
        return launch_edge_x3(q, nblocks, s);
    }
    if (f16) {
        switch (a->H) {
            case 32:  { ProfScope ps(K_EDGE16_H32, s, flops, bytes);  return launch_tile<32, 32, LOAD_EDGE, MODE_EDGEMAX, PREC_F16X3>(p, nblocks, s); }
            case 64:  { ProfScope ps(K_EDGE16_H64, s, flops, bytes);  return launch_tile<64, 32, LOAD_EDGE, MODE_EDGEMAX, PREC_F16X3>(p, nblocks, s); }
            case 128: { ProfScope ps(K_EDGE16_H128, s, flops, bytes); return launch_tile<128, 32, LOAD_EDGE, MODE_EDGEMAX, PREC_F16X3>(p, nblocks, s); }
            case 256: { ProfScope ps(K_EDGE16_H256, s, flops, bytes); return launch_tile<256, 32, LOAD_EDGE, MODE_EDGEMAX, PREC_F16X3>(p, nblocks, s); }
            default: return MORIG_E_UNSUPPORTED;
        }
    }
    if (a->exact_arith == 1) {                    // the exact path on three bf16 limbs (MORIG_SPLIT_BF16X6)
        switch (a->H) {
            case 32:  { ProfScope ps(K_EDGE_H32, s, flops, bytes);  return launch_tile<32, 32, LOAD_EDGE, MODE_EDGEMAX, PREC_BF16X6>(p, nblocks, s); }
            case 64:  { ProfScope ps(K_EDGE_H64, s, flops, bytes);  return launch_tile<64, 32, LOAD_EDGE, MODE_EDGEMAX, PREC_BF16X6>(p, nblocks, s); }
            case 128: { ProfScope ps(K_EDGE_H128, s, flops, bytes); return launch_tile<128, 32, LOAD_EDGE, MODE_EDGEMAX, PREC_BF16X6>(p, nblocks, s); }
            case 256: { ProfScope ps(K_EDGE_H256, s, flops, bytes); return launch_tile<256, 32, LOAD_EDGE, MODE_EDGEMAX, PREC_BF16X6>(p, nblocks, s); }
            default: break;
        }
    }
    switch (a->H) {
        case 16:  { ProfScope ps(K_EDGE_H16, s, flops, bytes);  return launch_tile<32, 16, LOAD_EDGE, MODE_EDGEMAX>(p, nblocks, s); }
        case 32:  { ProfScope ps(K_EDGE_H32, s, flops, bytes);  return launch_tile<32, 32, LOAD_EDGE, MODE_EDGEMAX>(p, nblocks, s); }
        case 64:  { ProfScope ps(K_EDGE_H64, s, flops, bytes);  return launch_tile<64, 32, LOAD_EDGE, MODE_EDGEMAX>(p, nblocks, s); }
        case 128: { ProfScope ps(K_EDGE_H128, s, flops, bytes); return launch_tile<128, 32, LOAD_EDGE, MODE_EDGEMAX>(p, nblocks, s); }
        case 256: { ProfScope ps(K_EDGE_H256, s, flops, bytes); return launch_tile<256, 16, LOAD_EDGE, MODE_EDGEMAX>(p, nblocks, s); }
        default: return MORIG_E_UNSUPPORTED;
    }
}

// EdgeConv whose vertex input has 3 channels: first Linear evaluated in the loader from the gathered endpoints (LOAD_EDGE3), then the
// 32-wide second layer + max exactly as morig_edgeconv. Replaces `morig_gemm` (K = 3 -> [A | B]) + `morig_edgeconv` for the position
// branches (models/basic_modules.py:193-195: nn_pos([pos_i, pos_j - pos_i])) and for motionNet's first unit (nn_x on the 3-channel flow).
extern "C" int morig_edgeconv_x3(const morig_edgeconv_x3_args* a_in, void* stream) {
    morig_edgeconv_x3_args mine;
    if (!take_args(a_in, mine, MORIG_EDGECONV_X3_ARGS_V3_SIZE)) return MORIG_E_INVALID;
    const morig_edgeconv_x3_args* a = &mine;
    if (!a->X || !a->W1a || !a->W1b || !a->b1 || !a->rowptr || !a->src_sorted || !a->dst_sorted || !a->W2 || !a->out) return MORIG_E_INVALID;
    if (!a->b2 || !a->s2 || !a->t2) return MORIG_E_INVALID;
    if (a->H != 32) return MORIG_E_UNSUPPORTED;
    if (a->n_nodes <= 0 || a->replicas <= 0 || a->edge_capacity <= 0 || a->ldx < 4 || (a->ldx & 3) || !aligned16(a->X)) return MORIG_E_INVALID;
    if ((a->ldw & 3) || a->ldw < 32 || a->ldo < 32 || !aligned16(a->W2) || !aligned16(a->W1a) || !aligned16(a->W1b)) return MORIG_E_INVALID;
    if (a->in_rep_stride < 0 || (a->replicas > 1 && a->out_rep_stride < a->n_nodes)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    TileParams p = {};
    p.M = a->edge_capacity; p.N = 32; p.K = 32;
    p.W = a->W2; p.ldw = a->ldw;
    p.bias = a->b2; p.scale = a->s2; p.shift = a->t2; p.relu = 1;
    p.X3 = a->X; p.ldx3 = a->ldx; p.W1a = a->W1a; p.W1b = a->W1b; p.b1 = a->b1;
    p.rowptr = a->rowptr; p.srcS = a->src_sorted; p.dstS = a->dst_sorted;
    p.n_nodes = a->n_nodes; p.rep_in = a->in_rep_stride; p.rep_out = a->out_rep_stride;
    p.Y = a->out; p.ldy = a->ldo;
    p.tiles_per_rep = cdiv(a->edge_capacity, 128);
    p.tiles_n = 1;
    const bool f16 = a->W2_split != nullptr;
    if (f16) {
        if (!a->overflow || !aligned16(a->W2_split)) return MORIG_E_INVALID;
        p.W = static_cast<const float*>(a->W2_split); p.ovf = a->overflow;
    }
    const int nblocks = p.tiles_per_rep * a->replicas;
    if (!a->skip_init) {
        BoundaryJob jobs[2];
        int nj = 0;
        jobs[nj++] = boundary_job(a->rowptr, a->dst_sorted, a->n_nodes, a->edge_capacity, 32, a->out, a->ldo, a->out_rep_stride, a->replicas, 128, 1, nullptr);
        if (a->init_with) {                                // the partner launch that follows (with skip_init): its rows in the same pass
            morig_edgeconv_x3_args o;
            if (!take_args(a->init_with, o, MORIG_EDGECONV_X3_ARGS_V3_SIZE)) return MORIG_E_INVALID;
            if (!o.rowptr || !o.dst_sorted || !o.out || o.H != 32 || o.n_nodes <= 0 || o.replicas <= 0 || o.edge_capacity <= 0 || o.ldo < 32 ||
                (o.replicas > 1 && o.out_rep_stride < o.n_nodes)) return MORIG_E_INVALID;
            jobs[nj++] = boundary_job(o.rowptr, o.dst_sorted, o.n_nodes, o.edge_capacity, 32, o.out, o.ldo, o.out_rep_stride, o.replicas, 128, 1, nullptr);
        }
        const int st2 = init_boundary_rows(jobs, nj, s);
        if (st2 != MORIG_OK) return st2;
    } else if (a->init_with) return MORIG_E_INVALID;
    const double E = (double)(a->edge_count > 0 ? a->edge_count : a->edge_capacity) * a->replicas;
    // algorithmic work: second layer 2 H^2 + first layer 2 * 6 * H per edge row; bytes: the two gathered 16-byte inputs
    ProfScope ps(f16 ? K_EDGE16_X3 : K_EDGE_X3, s, E * (2.0 * 32 * 32 + 12.0 * 32), 32.0 * E);
    // split-fp16 path: the persistent kernel (edge_x3.hip); MORIG_X3_TILE=1 keeps the one-tile-per-workgroup engine (A/B). The
    // exact-fp32 path (MORIG_PRECISION=f32, or the re-run after a range overflow) stays on the tile engine.
    static const bool want_tile = [] { const char* e = getenv("MORIG_X3_TILE"); return e && e[0] == '1'; }();
    if (f16 && !want_tile && (a->ldo & 3) == 0 && aligned16(a->out)) {
        EdgeX3Params q = {};
        q.X = a->X; q.ldx = a->ldx; q.W1a = a->W1a; q.W1b = a->W1b; q.b1 = a->b1;
        q.W2s = static_cast<const float*>(a->W2_split); q.ldw = a->ldw;
        q.bias = a->b2; q.scale = a->s2; q.shift = a->t2;
        q.rowptr = a->rowptr; q.srcS = a->src_sorted; q.dstS = a->dst_sorted; q.n_nodes = a->n_nodes; q.cap = a->edge_capacity;
        q.rep_in = a->in_rep_stride; q.rep_out = a->out_rep_stride; q.replicas = a->replicas;
        q.Y = a->out; q.ldy = a->ldo; q.ovf = a->overflow;
        prof_retag(K_EDGE16_X3P);
        return launch_edge_x3(q, nblocks, s);
    }
    return f16 ? launch_tile<32, 32, LOAD_EDGE3, MODE_EDGEMAX, PREC_F16X3>(p, nblocks, s)
               : launch_tile<32, 32, LOAD_EDGE3, MODE_EDGEMAX>(p, nblocks, s);
}
