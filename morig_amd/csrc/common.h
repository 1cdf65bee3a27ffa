// Shared internals of libmorig_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/morig_hip.h"

namespace morig {

// profiling kinds (index into the live event accounting; names in prof.hip)
enum Kind : int {
    K_GEMM_BN128 = 0, K_GEMM_BN64, K_GEMM_BN32, K_GEMM_POOL,
    K_EDGE_H16, K_EDGE_H32, K_EDGE_H64, K_EDGE_H128, K_EDGE_H256,
    K_CSR, K_COPY, K_ROWNORM, K_ATTN, K_MISC,
    K_FPS, K_BALL, K_POINTCONV, K_KNN_INTERP, K_COSINE_NN,
    K_GEMM16_BN128, K_GEMM16_BN64, K_GEMM16_BN32, K_GEMM16_POOL,
    K_EDGE16_H32, K_EDGE16_H64, K_EDGE16_H128, K_EDGE16_H256, K_POINTCONV16, K_GEMM16_DMA,
    K_COSINE_KNN, K_FLOW_VOTE, K_JOINTS,
    K_GEMM16_DMAP, K_GEMM16_DMA128, K_EDGE16_H256_PP, K_EDGE16_H128_WS, K_EDGE16_PC, K_GEOGRAPH, K_EDGE16_X3, K_EDGE_X3, K_EDGE16_X3P, K_EDGE16_H128_RL,
    K_COUNT
};
static_assert(K_COUNT <= MORIG_PROF_KINDS, "raise MORIG_PROF_KINDS");

void set_hip_error(hipError_t e);
// RAII: records a start/stop event pair around the enclosed launches when profiling is on.
struct ProfScope {
    ProfScope(int kind, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    int slot; hipStream_t stream; ProfScope* outer;
};
// The launcher that finally picks the kernel re-labels the innermost open scope of this thread, so that every kind of the
// hot kernels maps to ONE kernel symbol (morig_prof_symbol): bench.py's roofline is then recomputable from a rocprofv3 trace.
void prof_retag(int kind);

inline int check_hip(hipError_t e) {
    if (e == hipSuccess) return MORIG_OK;
    set_hip_error(e);
    return MORIG_E_HIP;
}
#define MORIG_HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { ::morig::set_hip_error(_e); return MORIG_E_HIP; } } while (0)
#define MORIG_LAUNCH_CHECK() MORIG_HIP_TRY(hipGetLastError())

// parameter blocks of the specialised kernels (edge_pc.hip, gemm_dma.hip), filled by the launchers in tile_gemm.hip
struct EdgePcParams {
    int H;
    const float* W; int ldw;                     // split-fp16 image of W2 [H][ldw]
    const float* bias; const float* scale; const float* shift;
    const float* A; int lda; const float* B; int ldb;
    const int* rowptr; const int* srcS; const int* dstS; int n_nodes; int rep_in; int rep_out; int tiles_per_rep;
    int replicas;
    float* Y; int ldy;
    int y16;                                     // edge_rl.hip / edge_ws.hip: results leave in the split-fp16 activation layout
    int run;                                     // edge_rl.hip / edge_ws.hip: consecutive tiles per wave / workgroup (open segments are carried)
    int* ovf;
    int quad;                                    // CSR segments are 4-aligned (MORIG_CSR_PAD4)
    int min4;                                    // CSR segments hold >= 4 rows, not aligned (MORIG_CSR_MIN4): edge_ws.hip's mixed-quad form
    unsigned long long* trace;                   // -DMORIG_PP_TRACE builds only
    int dbg;                                     // ablation (tools/microbench.py): 1 no epilogue, 2 no MFMA, 4 no gathers, 8 no W loads
};
struct GemmDmaParams {
    int M, N, K;
    const float* X; int ldx;                     // split-fp16 layout
    const float* Xt; int ldt; int tail_rows; int tail_chunks;   // morig_gemm_args.X_tail: the last tail_chunks chunks of X from Xt[row % tail_rows]
    const float* W; int ldw;                     // split-fp16 layout, rows padded to the column tile
    const float* bias; const float* scale; const float* shift; int relu;
    const float* rowbias; int ld_rowbias; const int* seg;
    float* Y; int ldy; int y16;
    float* pool; int ld_pool;                    // column max per segment instead of a store (seg sorted)
    int tiles_n;
    int* ovf;
    int dbg;                                     // ablation: 1 = no epilogue
    unsigned long long* trace;                   // -DMORIG_DMA_TRACE measurement builds: s_memtime stamps of one mid-launch workgroup           // first wave of workgroups: start delay = phase * ticks of the 100 MHz clock (0 = none)
};
int launch_edge_pc(const EdgePcParams& p, int nblocks, hipStream_t s);        // edge_pc.hip
void set_reserved_cus(int n);
int reserved_cus();
int launch_edge_pp(const EdgePcParams& p, int nblocks, hipStream_t s);        // edge_pp.hip (persistent)
int launch_edge_ws(const EdgePcParams& p, int nblocks, hipStream_t s);        // edge_ws.hip (persistent, W2 resident in registers; 4-aligned CSR)
int launch_edge_rl(const EdgePcParams& p, int ntiles64, hipStream_t s);       // edge_rl.hip (persistent, H = 128, W2 resident in LDS, waves independent; 4-aligned CSR)
int launch_gemm16_dma(const GemmDmaParams& p, int tiles_m128, hipStream_t s); // gemm_dma.hip
struct EdgeX3Params {
    const float* X; int ldx;                       // [rows][ldx >= 4]: 3 input channels per vertex
    const float* W1a; const float* W1b; const float* b1;      // [32][4], [32][4], [32]
    const float* A; int lda; const float* B; int ldb;          // X == nullptr: the per-vertex first-layer terms [rows][>= 32] instead (morig_edgeconv, H = 32)
    const float* W2s; int ldw;                     // split-fp16 image of W2 [32][ldw]
    const float* bias; const float* scale; const float* shift;
    const int* rowptr; const int* srcS; const int* dstS; int n_nodes; int cap;
    int rep_in, rep_out, replicas;
    float* Y; int ldy;
    int* ovf;
};

int launch_edge_x3(const EdgeX3Params& p, int n_tiles_cap, hipStream_t s);      // edge_x3.hip (persistent 32-wide EdgeConv: 3-channel inputs, or gathered [A | B] rows)
int launch_gemm16_dmap(const GemmDmaParams& p, hipStream_t s);                // gemm_dmap.hip (persistent, 256 x 256 tiles)
// vertex_ops.hip: a dense layer on at most 128 rows (fp32 X, W, Y; bias / ReLU / column affine), weights distributed over the chip
bool few_rows_gemm_takes(int M, int N, int K, int ldx, int ldw);
int launch_few_rows_gemm(const float* X, int ldx, int M, const float* W, int ldw, int N, int K, const float* bias, const float* scale,
                         const float* shift, int relu, float* Y, int ldy, hipStream_t s);

// [ABI 3] every argument struct starts with struct_size: the caller's struct is copied into a zeroed struct of THIS build (members the
// caller does not know read as 0); a size below the struct's ABI-3 size -- a version-1 / -2 caller, or garbage -- is refused.
static_assert(sizeof(morig_gemm_args) >= MORIG_GEMM_ARGS_V3_SIZE && sizeof(morig_edgeconv_args) >= MORIG_EDGECONV_ARGS_V3_SIZE &&
              sizeof(morig_edgeconv_x3_args) >= MORIG_EDGECONV_X3_ARGS_V3_SIZE && sizeof(morig_segmax_args) >= MORIG_SEGMAX_ARGS_V3_SIZE &&
              sizeof(morig_pointconv_args) >= MORIG_POINTCONV_ARGS_V3_SIZE, "argument structs only grow");
static_assert(sizeof(morig_edgeconv_args) == MORIG_EDGECONV_ARGS_SIZE && sizeof(morig_edgeconv_x3_args) == MORIG_EDGECONV_X3_ARGS_SIZE,
              "include/morig_hip.h states the current sizes");
template <class T> static inline bool take_args(const T* a, T& mine, uint32_t v3_size) {
    if (!a) return false;
    const uint32_t n = a->struct_size;
    if (n < v3_size || n > 1024u || (n & 7u)) return false;      // (an old-layout struct spells M / H / N here: almost always refused too)
    memset(&mine, 0, sizeof(T));
    memcpy(&mine, a, n < sizeof(T) ? n : sizeof(T));
    mine.struct_size = (uint32_t)sizeof(T);
    return true;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -------------------------------------------------------------------
// float max via integer atomics; identity element = 0xFFFFFFFF (what hipMemset 0xFF leaves):
//   v >= 0 : signed max on the bit pattern (any negative float / the init pattern is a negative int)
//   v <  0 : unsigned min on the bit pattern (larger magnitude = larger unsigned; non-negative
//            floats are < 0x80000000 so they survive; the init 0xFFFFFFFF is the unsigned maximum)
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (v == 0.0f) v = 0.0f;                         // -0 -> +0 (INT_MIN would lose to the init)
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else           atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// bijective XCD-aware remap of a linear block id: blocks b, b+8, b+16 ... (same XCD under the
// observed round-robin dispatch) receive CONTIGUOUS work items, so neighbours share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---- two-pass column reductions in fp64 (train_ops.hip / train_bwd.hip) ----------------------------------------------------------
// First pass: grid (64-column groups, row slabs), 256 threads = (float4 column quads of the group) x (row lanes); every slab writes
// its [2][cols] partial sums (zeros past the live rows). Second pass: 4 columns x 64 slab lanes per block. Every order is fixed.
// Rows per slab: 256, more once that would give more than ~1024 workgroups (stats_slab_rows).
inline int stats_slab_rows(int rows_cap, int cols) {
    // ~1024 workgroups over (64-column groups x row slabs): wide matrices get fewer, longer slabs -- the second pass reads
    // slabs x 2 x cols doubles on cols / 16 workgroups, which at 1024 slabs x 256 columns was 4 MB per call and 10-17 us per launch
    const int groups = (cols + 63) / 64;
    int slabs = 1024 / (groups > 0 ? groups : 1);
    if (slabs < 64) slabs = 64;
    int r = ((rows_cap > 0 ? rows_cap : 1) + slabs - 1) / slabs;
    r = (r + 3) & ~3;
    return r < 256 ? 256 : r;
}

#ifdef __HIPCC__
// Two fp32 values split for the split-fp16 contraction: hi = the two fp16(v) (truncated: ONE v_cvt_pkrtz), lo = the two
// fp16(v - hi) rounded to nearest (one v_fma_mix each: f32 v * 1.0 - f16 hi, written into one half of the destination).
// v - hi is exact in fp32, so this is bit-identical to the cvt / sub / cvt sequence the plain C expression compiles to, at 3 VALU
// per pair instead of 7. Operands travel as float bit patterns (the host pass type-checks the asm constraints too).
__device__ inline void split_pair_f16(float v0, float v1, float& hi, float& lo) {
    typedef __fp16 split_h2 __attribute__((ext_vector_type(2)));
    const split_h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);
    hi = __builtin_bit_cast(float, h);
#ifdef MORIG_SPLIT_C_FORM                                  // measurement variant: the plain C expression (A/B of the conversion cost)
    split_h2 l;
    l[0] = (__fp16)(v0 - (float)h[0]); l[1] = (__fp16)(v1 - (float)h[1]);
    lo = __builtin_bit_cast(float, l);
#else
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(v0), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(hi));
#endif
}
#endif

// bf16 counterpart (gradient contractions: bf16 keeps float32's exponent range, no range guard): hi = bf16(v), lo = bf16(v - hi), both
// rounded to nearest even by v_cvt_pk_bf16_f32 (one instruction per pair); ~16 mantissa bits per operand
#ifdef __HIPCC__
__device__ inline void split_pair_bf16(float v0, float v1, float& hi, float& lo) {
    typedef __bf16 split_b2 __attribute__((ext_vector_type(2)));
    const split_b2 h = {(__bf16)v0, (__bf16)v1};
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    hi = __builtin_bit_cast(float, h);
    const float r0 = v0 - __uint_as_float(hb << 16), r1 = v1 - __uint_as_float(hb & 0xffff0000u);
    const split_b2 l = {(__bf16)r0, (__bf16)r1};
    lo = __builtin_bit_cast(float, l);
}
#endif

#ifdef __HIPCC__
// the same first limb together with the EXACT remainders r = v - hi (fp32: the difference of a float and its bf16 rounding is exact):
// the input of the next limb of a three-limb split (tile_gemm.hip PREC_BF16X6)
__device__ inline void split_pair_bf16_rem(float v0, float v1, float& hi, float& r0, float& r1) {
    typedef __bf16 split_b2 __attribute__((ext_vector_type(2)));
    const split_b2 h = {(__bf16)v0, (__bf16)v1};
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    hi = __builtin_bit_cast(float, h);
    r0 = v0 - __uint_as_float(hb << 16);
    r1 = v1 - __uint_as_float(hb & 0xffff0000u);
}
#endif

// V consecutive floats of a row (V = 4: one 16-byte access; V = 1: any alignment). vec4_ok(): the host-side predicate
template <int V> struct VecF { float v[V]; };
#ifdef __HIPCC__
template <int V> __device__ inline VecF<V> ldv(const float* p) {
    VecF<V> r;
    if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else r.v[0] = *p;
    return r;
}
template <int V> __device__ inline void stv(float* p, const VecF<V>& r) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    else *p = r.v[0];
}
#endif
static inline bool vec4_ptr(const void* p, long ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; }

#ifdef __HIPCC__
// the block's row-lane sums (acc[k][j]: statistic k of column c0 + 4 q + j) -> part[slab][k][cols], row lanes added in order
__device__ inline void stats_block_store(const double (&acc)[2][4], int quads, int RL, int q, int rl, int c0, int cols, int slab,
                                         double* __restrict__ part) {
    __shared__ double sh[2 * 1024];                                   // [k][column of the group][row lane]
    const int gc = quads * 4;
    if (rl < RL) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) sh[(k * gc + q * 4 + j) * RL + rl] = acc[k][j];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 2 * gc) {
        const int k = t / gc, col = t - k * gc;
        double s = 0.0;
        for (int i = 0; i < RL; ++i) s += sh[(k * gc + col) * RL + i];
        if (c0 + col < cols) part[((size_t)slab * 2 + k) * cols + c0 + col] = s;
    }
}

// second pass of one statistic pair: STATS_FC columns x 64 slab lanes per workgroup (the partials of a column are few and small:
// what a launch costs is the length of the longest lane's chain of additions -- 16 columns x 16 lanes was 64 dependent steps at 1024
// slabs), then a fixed tree over the lanes. Returns through s / q the sums of column c, valid on threadIdx.x < STATS_FC.
constexpr int STATS_FC = 4;
__device__ inline void stats_final_sums(const double* __restrict__ part, int slabs, int cols, int c, double& s, double& q) {
    const int sl = threadIdx.x / STATS_FC, l = threadIdx.x % STATS_FC;
    constexpr int LANES = 256 / STATS_FC;
    s = 0.0; q = 0.0;
    if (c < cols)
        for (int b = sl; b < slabs; b += LANES) { s += part[((size_t)b * 2 + 0) * cols + c]; q += part[((size_t)b * 2 + 1) * cols + c]; }
    __shared__ double sh[2][LANES][STATS_FC];
    sh[0][sl][l] = s; sh[1][sl][l] = q;
    __syncthreads();
    for (int h = LANES / 2; h > 0; h >>= 1) {
        if (sl < h) { sh[0][sl][l] += sh[0][sl + h][l]; sh[1][sl][l] += sh[1][sl + h][l]; }
        __syncthreads();
    }
    s = sh[0][0][l]; q = sh[1][0][l];
}
#endif

}  // namespace morig
