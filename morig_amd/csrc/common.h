// Shared internals of libmorig_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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
    K_COUNT
};
static_assert(K_COUNT <= MORIG_PROF_KINDS, "raise MORIG_PROF_KINDS");

void set_hip_error(hipError_t e);
// RAII: records a start/stop event pair around the enclosed launches when profiling is on.
struct ProfScope {
    ProfScope(int kind, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    int slot; hipStream_t stream;
};

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
    int* ovf;
    int quad;                                    // CSR segments are 4-aligned (MORIG_CSR_PAD4)
    unsigned long long* trace;                   // -DMORIG_PP_TRACE builds only
    int dbg;                                     // ablation (tools/microbench.py): 1 no epilogue, 2 no MFMA, 4 no gathers, 8 no W loads
};
struct GemmDmaParams {
    int M, N, K;
    const float* X; int ldx;                     // split-fp16 layout
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
int launch_gemm16_dma(const GemmDmaParams& p, int tiles_m128, hipStream_t s); // gemm_dma.hip
int launch_gemm16_dmap(const GemmDmaParams& p, hipStream_t s);                // gemm_dmap.hip (persistent, 256 x 256 tiles)

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

}  // namespace morig
