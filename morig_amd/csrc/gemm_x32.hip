// Dense GEMM with an fp32 X operand on the split-fp16 path, PERSISTENT and wave-specialised: Y = epilogue(X W^T).
//
// Where it sits: the vertex MLP of a GCU (models/basic_modules.py:170, 212: MLP([O (+ 2 D), O]) on the concatenated EdgeConv
// outputs). Its input comes out of the segmented-max epilogues as fp32 -- tile-straddling segments are finished by atomics, so the
// producers cannot emit the split activation layout (measured twice) -- and the generic tile engine
// (tile_gemm.hip, 128 x 128 tile) converts every X row to (hi, lo) again for each 128-column tile, in the same four waves that
// issue the MFMAs, between two barriers per chunk (MFMA utilisation 0.39).
// Here the roles are split as in edge_pp.hip, without the gather: waves 8-11 PRODUCE -- coalesced 16-byte loads of the X chunk
// (128 rows x 32 k) and of the W chunk (256 columns x 32 k, already a split image), fp32 -> fp16 (hi, lo) with v_cvt_pkrtz /
// v_fma_mix, 8-byte LDS writes -- and waves 0-7 CONSUME: 64 x 64 wave tiles, 24 MFMAs per chunk and wave, fragments by
// ds_read_b128 from the padded rows; two consumer waves per SIMD, so that one's LDS round trips run under the other's MFMAs
// (a first version with four 64 x 128 consumers, one per SIMD, measured 15 % SLOWER than the tile engine: nothing covered the
// fragment latency after each barrier). X is converted once per 256 output columns. One s_barrier per chunk hands ring stage g & 1
// over; the producers load chunk g + 2 into registers while chunk g + 1 is being written, across tile boundaries (persistent
// workgroups: the next tile's first chunks are staged under the epilogue of the current one, which transposes through its own
// LDS scratch: epilogue_store.h, 16-row slabs).
//
// Takes: N a multiple of 256, K a multiple of 32, no row bias, no pooling (those stay on tile_gemm.hip / gemm_dma.hip).
#include "common.h"
#include "epilogue_store.h"
#include <atomic>
#include <stdlib.h>

namespace morig {

typedef float xp_f32x16 __attribute__((ext_vector_type(16)));
typedef float xp_f32x4 __attribute__((ext_vector_type(4)));
typedef float xp_b32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xp_f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 xp_h2 __attribute__((ext_vector_type(2)));

constexpr int XP_BM = 128, XP_BN = 256, XP_KC = 32, XP_LDB = 144;     // LDB: bytes per LDS row = [32 hi | 32 lo | 16 pad]
constexpr int XP_STAGE = (XP_BM + XP_BN) * XP_LDB;
constexpr int XP_NCW = 8, XP_NT = 2;                                  // consumer waves; 32-column MFMA tiles per consumer wave
using XpEpi = EpilogueTile<XP_NT, 16, 4>;

__device__ __forceinline__ void xp_barrier() {           // drains this wave's LDS traffic only: prefetch loads stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(768) void gemm_x32_pc_kernel(const GemmDmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * XP_STAGE + XP_NCW * XpEpi::FLOATS * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= XP_NCW;
    const int tiles_n = p.N / XP_BN, tiles_m = (p.M + XP_BM - 1) / XP_BM;
    const int T = tiles_m * tiles_n, NCH = p.K / XP_KC;
    // XCD x (= blockIdx & 7 under round-robin dispatch) owns a contiguous range of tiles; the column tiles of a row tile are adjacent
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int t_lo = (int)((long long)T * xcd / 8), t_hi = (int)((long long)T * (xcd + 1) / 8);
    const int n_my = (t_hi - t_lo - bi + nbx - 1) / nbx;
    if (n_my <= 0) return;                                                   // block-uniform
    const int G = n_my * NCH;                                                // chunks this workgroup walks
    auto tile_of = [&](int j) __attribute__((always_inline)) { return t_lo + bi + j * nbx; };

    if (producer) {
        const int pt = tid - 64 * XP_NCW;
        // X: 4 pieces of 16 B per thread and chunk (row = q >> 3, piece = q & 7); W: 8 pieces (column = q >> 3)
        xp_f32x4 xr[4], wr[8];
        const char* Xb = reinterpret_cast<const char*>(p.X);
        const char* Wb = reinterpret_cast<const char*>(p.W);
        float amax = 0.f;
        auto issue = [&](int g) __attribute__((always_inline)) {
            const int j = g / NCH, c = g - j * NCH;
            const int t = tile_of(j), tm = t / tiles_n, tn = t - tm * tiles_n;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = pt + 256 * i;
                const int row = min(tm * XP_BM + (q >> 3), p.M - 1);          // rows past the end re-read the last one (never stored)
                xr[i] = *reinterpret_cast<const xp_f32x4*>(Xb + ((size_t)row * p.ldx + c * XP_KC + (q & 7) * 4) * 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = pt + 256 * i;
                const int n = tn * XP_BN + (q >> 3);
                wr[i] = *reinterpret_cast<const xp_f32x4*>(Wb + (size_t)n * p.ldw * 4 + c * 128 + (q & 7) * 16);
            }
        };
        auto stage = [&](int g) __attribute__((always_inline)) {
            char* st = smem + (g & 1) * XP_STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = pt + 256 * i;
                xp_b32x2 hw, lw;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const float v0 = xr[i][e], v1 = xr[i][e + 1];
                    const xp_h2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);       // hi = fp16(v) truncated; lo = fp16(v - hi): hi + lo == v to 2^-22
                    const float hb = __builtin_bit_cast(float, h);
                    float lb;
                    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(v0), "v"(hb));
                    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb) : "v"(v1), "v"(hb));
                    hw[e >> 1] = hb; lw[e >> 1] = lb;
                    amax = fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1)));
                }
                char* rowp = st + (q >> 3) * XP_LDB + (q & 7) * 8;
                *reinterpret_cast<xp_b32x2*>(rowp) = hw;
                *reinterpret_cast<xp_b32x2*>(rowp + 64) = lw;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = pt + 256 * i;
                *reinterpret_cast<xp_f32x4*>(st + (XP_BM + (q >> 3)) * XP_LDB + (q & 7) * 16) = wr[i];
            }
        };
        issue(0);
        stage(0);
        if (G > 1) issue(1);
        for (int g = 0; g < G; ++g) {
            xp_barrier();                                                    // B_g: chunk g visible, stage (g + 1) & 1 free
            if (g + 1 < G) {
                stage(g + 1);
                if (g + 2 < G) issue(g + 2);
            }
        }
        if (!(amax < 65000.f)) *p.ovf = 1;
        return;
    }

    // ---- consumers: wave (wm, wn) owns rows [64 wm, 64 wm + 64) x columns [64 wn, 64 wn + 64) of the tile ----
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    float* T_scratch = reinterpret_cast<float*>(smem + 2 * XP_STAGE) + wave * XpEpi::FLOATS;
    xp_f32x16 acc[2][XP_NT];
    int g = 0;
    for (int j = 0; j < n_my; ++j) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < XP_NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        for (int c = 0; c < NCH; ++c, ++g) {
            xp_barrier();                                                    // B_g
            const char* st = smem + (g & 1) * XP_STAGE;
            const char* za = st + (wm * 64 + l31) * XP_LDB + 16 * hi;
            const char* wb = st + (XP_BM + wn * 32 * XP_NT + l31) * XP_LDB + 16 * hi;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                xp_f16x8 ah[2], al[2], bh[XP_NT], bl[XP_NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    ah[mt] = *reinterpret_cast<const xp_f16x8*>(za + mt * 32 * XP_LDB + 32 * ks);
                    al[mt] = *reinterpret_cast<const xp_f16x8*>(za + mt * 32 * XP_LDB + 32 * ks + 64);
                }
#pragma unroll
                for (int nt = 0; nt < XP_NT; ++nt) {
                    bh[nt] = *reinterpret_cast<const xp_f16x8*>(wb + nt * 32 * XP_LDB + 32 * ks);
                    bl[nt] = *reinterpret_cast<const xp_f16x8*>(wb + nt * 32 * XP_LDB + 32 * ks + 64);
                }
#pragma unroll
                for (int term = 0; term < 3; ++term)                      // the four accumulators between two uses of the same one
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < XP_NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[mt] : ah[mt], term == 1 ? bl[nt] : bh[nt],
                                                                                 acc[mt][nt], 0, 0, 0);
            }
        }
        const int t = tile_of(j), tm = t / tiles_n, tn = t - tm * tiles_n;
        store_tile_transposed<2, XP_NT, true, GemmDmaParams, 16, 4>(p, acc, T_scratch, nullptr, wm * 64, tm * XP_BM, p.M,
                                                                     tn * XP_BN + wn * 32 * XP_NT, lane);
    }
}

static int xp_cu_count() {
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

bool gemm_x32_pc_takes(int M, int N, int K, bool rowbias) {
    static const int off = [] { const char* e = getenv("MORIG_GEMM_X32PC"); return e && atoi(e) == 0 ? 1 : 0; }();
    return !off && !rowbias && N % XP_BN == 0 && K % XP_KC == 0 && K >= 2 * XP_KC && M >= 8 * XP_BM;
}

int launch_gemm_x32_pc(const GemmDmaParams& p, hipStream_t s) {
    int ncu = xp_cu_count();
    ncu = ncu > 8 ? (ncu / 8) * 8 : 8;
    int avail = ncu - ((reserved_cus() + 7) / 8) * 8;
    if (avail < 8) avail = 8;
    const int T = cdiv(p.M, XP_BM) * (p.N / XP_BN);
    const int grid = T < avail ? ((T + 7) / 8) * 8 : avail;                  // one persistent workgroup per CU, a multiple of 8 (XCDs)
    hipLaunchKernelGGL(gemm_x32_pc_kernel, dim3(grid), dim3(768), 0, s, p);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig
