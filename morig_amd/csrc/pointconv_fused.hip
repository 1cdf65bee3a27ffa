// Fused PointConv for the CorrNet set-abstraction levels (models/basic_modules.py:66-86, PyG PointConv.message / aggr='max'):
//
//   out[c] = max over the edges (j -> c) of  s3 * relu(W3 z2 + b3) + t3,   z2 = relu(W2 relu(A_c + B_j) + b2)
//
// straight from the ball query's slot table (centre c owns 64 slots of source indices, -1 = unused), with the graph
// normalisation PointConv applies folded in: pairs with source index == centre index are dropped and one self loop (c, c)
// is added (remove_self_loops / add_self_loops on the bipartite pair, the quirk morig_csr_from_slots reproduces).
//
// Why not the two-pass path (morig_edge_hidden -> morig_segmax_gemm): the per-edge hidden rows Z [E][H] went through HBM
// (sa1: 8.5 M edges x 128 B written and read back), each pass ran a narrow tile at ~45 TFLOP/s, and the segmented scan
// works for arbitrary CSRs although a centre's edges are already 64 contiguous slots. Here:
//   * a WAVE owns 32 slot rows at a time (half a centre) and carries them through the whole chain alone: gather + add +
//     ReLU + split into its private LDS rows X1 -> layer 2 on MFMA, computed TRANSPOSED (D = W2 . X1^T, a lane owns an edge
//     and 4 adjacent features per register group) so that the result goes back into the same LDS rows as the split-fp16
//     operand X2 with 8-byte writes -> layer 3 on MFMA the usual way round (a lane owns a feature column, the 32 edges
//     sit in its registers) -> the max over edges is 15 register maxima + one cross-half exchange. No workgroup barrier
//     after the weights are in LDS, no atomics, no scan, nothing per edge leaves the CU.
//   * unused and dropped slots are filled with the self-loop source c: max is idempotent, so duplicates are harmless and
//     every row is a valid edge. A centre whose 64 slots are all kept has no free slot for its self loop, so the self loops
//     of all centres run as a second, tiny pass (32 centres per wave step, read-max-write of `out`): 1.5 % extra rows.
//   * W2 and W3 (split-fp16 images, BN2's affine folded into W3 / b3 by the host) stay resident in LDS for the whole
//     persistent launch; b2 and b3 are the accumulators' initial values.
// Arithmetic: the same 3-MFMA split products as every other contraction, same range guard.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace morig {

typedef float pf_f32x16 __attribute__((ext_vector_type(16)));
typedef float pf_f32x4 __attribute__((ext_vector_type(4)));
typedef float pf_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pf_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 pf_h2 __attribute__((ext_vector_type(2)));

struct PcfParams {
    const float* A; int lda;                     // per-centre first-layer rows [M][lda]   (-W1p pos_c)
    const float* B; int ldb;                     // per-source first-layer rows [nsrc][ldb] (W1 [x_j | pos_j] + b1)
    const long long* slots;                      // [M][64] source indices, -1 = unused
    int M, nsrc;
    const float* W2; int ldw2; const float* b2;  // split image [H][ldw2], bias [H]
    const float* W3; int ldw3; const float* b3; const float* s3; const float* t3; int relu3;   // split image [H3][ldw3]
    float* out; int ldo;
    int* ovf; int* status;
};

// v_max_f32 without the canonicalising self-max clang puts in front of fmaxf() when an operand comes out of an MFMA or a load
// (IEEE sNaN quieting: one extra VALU instruction per element in epilogues that are VALU-bound). NaN operands: the other one
// is returned, like fmaxf; the range guard flags NaNs before anything is compared.
// NEVER on an MFMA result: the matrix pipe's write -> VALU read hazard is resolved by the compiler (s_nop), and it does not look
// into inline assembly -- the first registers read after the last MFMA came back without its contribution. Accumulators go
// through clamp_lo() (v_med3_f32 from the builtin: hazard-tracked, and not canonicalised either).
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (upper bound FLT_MAX, not +inf: with +inf the compiler folds the median back into a canonicalising fmax; anything near it has
// tripped the range guard long before)
__device__ __forceinline__ float clamp_lo(float x, float lo) { return __builtin_amdgcn_fmed3f(x, lo, 3.4028234e38f); }

// hi = fp16(v) truncated, lo = fp16(v - hi) rounded to nearest (v_fma_mix: f32 v * 1.0 - f16 hi): the split of epilogue_store.h
__device__ __forceinline__ void split4(const float (&v)[4], pf_f32x2& hw, pf_f32x2& lw, float& am) {
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
        const pf_h2 h = __builtin_amdgcn_cvt_pkrtz(v[q], v[q + 1]);
        const float hb = __builtin_bit_cast(float, h);
        float lb;
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(v[q]), "v"(hb));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(v[q + 1]), "v"(hb));
        hw[q >> 1] = hb; lw[q >> 1] = lb;
        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(am) : "v"(v[q]), "v"(v[q + 1]));
    }
}

template <int H, int H3, int NW, bool SELF>
__global__ __launch_bounds__(NW * 64, (H <= 32 ? 3 : 2)) void pcf_kernel(const PcfParams p) {
    constexpr int LDR = (H + 4) * 4;             // bytes per LDS row: H/32 chunks of [32 hi halves | 32 lo halves] + 16 B pad
    constexpr int NT1 = H / 32, NT3 = H3 / 32, KS = H / 16;
    constexpr int LPR = H / 4;                   // lanes per gathered row (16 B each)
    constexpr int RPI = 64 / LPR;                // rows per gather instruction
    constexpr int NG = 32 / RPI;                 // gather instructions per 32-row step
    static_assert(H % 32 == 0 && H3 % 64 == 0 && H <= 64, "layer widths");
    __shared__ __attribute__((aligned(16))) char smem[(H + H3 + 32 * NW) * LDR + H * 4];
    char* sW2 = smem;
    char* sW3 = smem + H * LDR;
    const float* sb2 = reinterpret_cast<const float*>(smem + (H + H3 + 32 * NW) * LDR);   // accumulator seed of layer 2

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    char* X = smem + (H + H3) * LDR + wave * 32 * LDR;           // this wave's 32 operand rows

    // ---- weights -> LDS (once per workgroup) ----
    for (int i = tid; i < (H + H3) * LPR; i += NW * 64) {
        const int r = i / LPR, q = i % LPR;
        const float* src = r < H ? p.W2 + (size_t)r * p.ldw2 : p.W3 + (size_t)(r - H) * p.ldw3;
        *reinterpret_cast<pf_f32x4*>(smem + r * LDR + 16 * q) = *reinterpret_cast<const pf_f32x4*>(src + 4 * q);
    }
    if (tid < H) reinterpret_cast<float*>(smem + (H + H3 + 32 * NW) * LDR)[tid] = p.b2[tid];
    float b3r[NT3], s3r[NT3], t3r[NT3];          // layer 3: lane owns feature nt*32 + l31
#pragma unroll
    for (int nt = 0; nt < NT3; ++nt) {
        b3r[nt] = p.b3 ? p.b3[nt * 32 + l31] : 0.f;
        s3r[nt] = p.s3 ? p.s3[nt * 32 + l31] : 1.f;
        t3r[nt] = p.t3 ? p.t3[nt * 32 + l31] : 0.f;
    }
    const float lo3 = p.relu3 ? 0.f : -INFINITY;
    __syncthreads();

    const int q4 = lane % LPR, rsub = lane / LPR;                // gather: my 4 columns, my row inside an instruction
    char* xst = X + rsub * LDR + (q4 >> 3) * 128 + (q4 & 7) * 8; // where my staged quad goes (hi halves; lo at +64)
    float amax = 0.f;

    // ---- the chain for the 32 rows in X ----
    auto stage = [&](const pf_f32x4 (&raw)[NG], const pf_f32x4 (&arow)[SELF ? NG : 1]) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const pf_f32x4 a = arow[SELF ? i : 0];
            const float v[4] = {vmax(raw[i][0] + a[0], 0.f), vmax(raw[i][1] + a[1], 0.f), vmax(raw[i][2] + a[2], 0.f),
                                vmax(raw[i][3] + a[3], 0.f)};
            pf_f32x2 hw, lw;
            split4(v, hw, lw, amax);
            *reinterpret_cast<pf_f32x2*>(xst + i * RPI * LDR) = hw;
            *reinterpret_cast<pf_f32x2*>(xst + i * RPI * LDR + 64) = lw;
        }
        // the rows are written as float pairs and read back as half vectors: without this the compiler is free to move the reads
        // of layer 2 above these writes (type-based aliasing); the hardware keeps a wave's LDS operations in order
        asm volatile("" ::: "memory");
    };
    const char* xr = X + l31 * LDR + 16 * hi;
    // layer 2, transposed: D[feature][edge] = W2 . X1^T, then ReLU, split, back into my rows as the operand of layer 3
    auto layer2 = [&]() {
        pf_f32x16 acc1[NT1];
        // register r of tile nt is feature nt*32 + 8*(r>>2) + 4*hi + (r&3): four adjacent biases per register group
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const pf_f32x4 b = *reinterpret_cast<const pf_f32x4*>(sb2 + nt * 32 + 8 * g + 4 * hi);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc1[nt][4 * g + q] = b[q];
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = 128 * (ks >> 1) + 32 * (ks & 1);
            const pf_f16x8 xh = *reinterpret_cast<const pf_f16x8*>(xr + off);
            const pf_f16x8 xl = *reinterpret_cast<const pf_f16x8*>(xr + off + 64);
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) {
                const char* wr = sW2 + (nt * 32 + l31) * LDR + 16 * hi + off;
                const pf_f16x8 wh = *reinterpret_cast<const pf_f16x8*>(wr);
                const pf_f16x8 wl = *reinterpret_cast<const pf_f16x8*>(wr + 64);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc1[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc1[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc1[nt], 0, 0, 0);
            }
            asm volatile("" ::: "memory");           // keep the fragment reads of later steps from being hoisted (register pressure)
        }
        // same-wave LDS traffic is ordered: the operand reads above are done before these writes land
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v[4] = {clamp_lo(acc1[nt][4 * g], 0.f), clamp_lo(acc1[nt][4 * g + 1], 0.f), clamp_lo(acc1[nt][4 * g + 2], 0.f),
                                    clamp_lo(acc1[nt][4 * g + 3], 0.f)};
                pf_f32x2 hw, lw;
                split4(v, hw, lw, amax);
                char* o = X + l31 * LDR + nt * 128 + 16 * g + 8 * hi;
                *reinterpret_cast<pf_f32x2*>(o) = hw;
                *reinterpret_cast<pf_f32x2*>(o + 64) = lw;
            }
        asm volatile("" ::: "memory");
    };
    // layer 3 for two feature tiles at a time (accumulators: 32 registers): D[edge][feature] = X2 . W3^T
    constexpr int NP = NT3 / 2;
    auto layer3 = [&](auto pc, pf_f32x16 (&acc3)[2]) {
        constexpr int np = decltype(pc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[j][r] = b3r[2 * np + j];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = 128 * (ks >> 1) + 32 * (ks & 1);
            const pf_f16x8 xh = *reinterpret_cast<const pf_f16x8*>(xr + off);
            const pf_f16x8 xl = *reinterpret_cast<const pf_f16x8*>(xr + off + 64);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* wr = sW3 + ((2 * np + j) * 32 + l31) * LDR + 16 * hi + off;
                const pf_f16x8 wh = *reinterpret_cast<const pf_f16x8*>(wr);
                const pf_f16x8 wl = *reinterpret_cast<const pf_f16x8*>(wr + 64);
                acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, acc3[j], 0, 0, 0);
                acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, acc3[j], 0, 0, 0);
                acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, acc3[j], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
    };

    const int gw = blockIdx.x * NW + wave, GW = gridDim.x * NW;
    if constexpr (SELF) {
        // ---- self loops: row j of a step is the edge (c0 + j -> c0 + j); out = max(out, message) ----
        for (int c0 = gw * 32; c0 < p.M; c0 += GW * 32) {
            pf_f32x4 raw[NG], arow[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int c = min(c0 + i * RPI + rsub, p.M - 1);
                raw[i] = *reinterpret_cast<const pf_f32x4*>(p.B + (size_t)c * p.ldb + 4 * q4);
                arow[i] = *reinterpret_cast<const pf_f32x4*>(p.A + (size_t)c * p.lda + 4 * q4);
            }
            stage(raw, arow);
            layer2();
            auto half = [&](auto pc) {
                constexpr int np = decltype(pc)::value;
                pf_f32x16 acc3[2];
                layer3(pc, acc3);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (c < p.M) {
                            float* o = p.out + (size_t)c * p.ldo + (2 * np + j) * 32 + l31;
                            const float v = clamp_lo(acc3[j][r], lo3) * s3r[2 * np + j] + t3r[2 * np + j];
                            *o = vmax(*o, v);
                        }
                    }
            };
            half(std::integral_constant<int, 0>{});
            if constexpr (NP > 1) half(std::integral_constant<int, 1>{});
        }
    } else {
        // ---- slot rows: a centre per iteration, 32 slots per step; the rows of the next step are in flight under the MFMAs ----
        auto sources = [&](int c) -> int {           // lane l: the source of slot l of centre c (filler: the self-loop source c)
            const long long v = p.slots[(size_t)c * 64 + lane];
            if (v >= p.nsrc) *p.status = 1;
            return (v >= 0 && v < p.nsrc && v != c) ? (int)v : c;
        };
        auto gather = [&](pf_f32x4 (&raw)[NG], int src, int half) {
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int s = __shfl(src, 32 * half + i * RPI + rsub, 64);
                raw[i] = *reinterpret_cast<const pf_f32x4*>(p.B + (size_t)s * p.ldb + 4 * q4);
            }
        };
        auto step = [&](float (&m)[NT3], bool first) {            // layers 2 and 3 of the 32 staged rows, folded into the running max
            layer2();
            auto half = [&](auto pc) {
                constexpr int np = decltype(pc)::value;
                pf_f32x16 acc3[2];
                layer3(pc, acc3);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float mm = first ? -INFINITY : m[2 * np + j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) mm = vmax(mm, clamp_lo(acc3[j][r], lo3) * s3r[2 * np + j] + t3r[2 * np + j]);
                    m[2 * np + j] = mm;
                }
            };
            half(std::integral_constant<int, 0>{});
            if constexpr (NP > 1) half(std::integral_constant<int, 1>{});
        };
        int c = gw;
        if (c < p.M) {
            int src = sources(c);
            pf_f32x4 arow[1] = {*reinterpret_cast<const pf_f32x4*>(p.A + (size_t)c * p.lda + 4 * q4)};
            pf_f32x4 raw[NG];
            gather(raw, src, 0);
            while (true) {
                const int cn = c + GW;
                const bool more = cn < p.M;
                int src_n = 0;
                pf_f32x4 arow_n[1] = {arow[0]};
                if (more) {                                   // the next centre's slot row and A row: a whole centre ahead of their use
                    src_n = sources(cn);
                    arow_n[0] = *reinterpret_cast<const pf_f32x4*>(p.A + (size_t)cn * p.lda + 4 * q4);
                }
                float m[NT3];
                stage(raw, arow);
                gather(raw, src, 1);
                step(m, true);
                stage(raw, arow);
                if (more) gather(raw, src_n, 0);
                step(m, false);
#pragma unroll
                for (int nt = 0; nt < NT3; ++nt) {
                    const float o = vmax(m[nt], __shfl_xor(m[nt], 32, 64));
                    if (hi == 0) p.out[(size_t)c * p.ldo + nt * 32 + l31] = o;
                }
                if (!more) break;
                c = cn; src = src_n; arow[0] = arow_n[0];
            }
        }
    }
    if (!(amax < 65000.f)) *p.ovf = 1;           // also catches NaN
}

static int cu_count() {
    static const int n = [] { int d = 0, c = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d); return c; }();
    return n;
}

template <int H, int H3>
static int launch_pcf(const PcfParams& p, hipStream_t s) {
    // H = 64: one workgroup of 8 waves per CU (122 KB of LDS, 230 VGPRs); H = 32: three workgroups of 4 waves (32 KB, <= 168 VGPRs)
    constexpr int NW = H <= 32 ? 4 : 8;
    const int per_cu = H <= 32 ? 3 : 1;
    const int units = (p.M + NW - 1) / NW;
    int grid = cu_count() * per_cu;
    if (grid > units) grid = units;
    hipLaunchKernelGGL((pcf_kernel<H, H3, NW, false>), dim3(grid), dim3(NW * 64), 0, s, p);
    MORIG_LAUNCH_CHECK();
    const int sunits = (p.M + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL((pcf_kernel<H, H3, NW, true>), dim3(sunits < grid ? sunits : grid), dim3(NW * 64), 0, s, p);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig

using namespace morig;

extern "C" int morig_pointconv_fused(const morig_pointconv_args* a_in, void* stream) {
    morig_pointconv_args mine;
    if (!take_args(a_in, mine, MORIG_POINTCONV_ARGS_V3_SIZE)) return MORIG_E_INVALID;
    const morig_pointconv_args* a = &mine;
    if (!a->A || !a->B || !a->slots || !a->W2_split || !a->b2 || !a->W3_split || !a->out || !a->overflow || !a->status)
        return MORIG_E_INVALID;
    if (a->n_centres <= 0 || a->n_src < a->n_centres || a->max_nbrs != 64) return MORIG_E_INVALID;
    if (a->lda < a->H || a->ldb < a->H || (a->lda & 3) || (a->ldb & 3) || a->ldw2 < a->H || a->ldw3 < a->H || (a->ldw2 & 3) || (a->ldw3 & 3) ||
        a->ldo < a->H3)
        return MORIG_E_INVALID;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a->A) | reinterpret_cast<uintptr_t>(a->B) | reinterpret_cast<uintptr_t>(a->W2_split) |
                         reinterpret_cast<uintptr_t>(a->W3_split) | reinterpret_cast<uintptr_t>(a->b2);
    if (al & 15) return MORIG_E_INVALID;
    PcfParams p = {};
    p.A = a->A; p.lda = a->lda; p.B = a->B; p.ldb = a->ldb;
    p.slots = reinterpret_cast<const long long*>(a->slots); p.M = a->n_centres; p.nsrc = a->n_src;
    p.W2 = static_cast<const float*>(a->W2_split); p.ldw2 = a->ldw2; p.b2 = a->b2;
    p.W3 = static_cast<const float*>(a->W3_split); p.ldw3 = a->ldw3; p.b3 = a->b3; p.s3 = a->s3; p.t3 = a->t3; p.relu3 = a->relu3;
    p.out = a->out; p.ldo = a->ldo; p.ovf = a->overflow; p.status = a->status;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const double E = (double)a->n_centres * 65.0;
    ProfScope ps(K_POINTCONV16, s, 2.0 * E * a->H * ((double)a->H + a->H3), 4.0 * E * a->H + 8.0 * E + 4.0 * a->n_centres * (double)a->H3);
    if (a->H == 32 && a->H3 == 64) return launch_pcf<32, 64>(p, s);
    if (a->H == 64 && a->H3 == 128) return launch_pcf<64, 128>(p, s);
    return MORIG_E_UNSUPPORTED;
}
