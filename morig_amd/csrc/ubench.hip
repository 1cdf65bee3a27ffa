// In-library probe of the matrix pipes' POWER-CAPPED rate (bench.py `roofline.mfma_power_capped_tflops`): the dense f16 MFMA peak of
// the data sheet (2.5 PFLOP/s at 2.4 GHz) assumes a clock the socket power cap does not grant once the operands toggle -- measured on
// MI355X: ~1.65 PFLOP/s at 1.66-1.68 GHz with random operands, 2.43 PFLOP/s at 2.4 GHz with zeros (profiles/r05g_mfma_power_ceiling.txt).
// One persistent launch: one workgroup of 512 threads per CU (2 waves per SIMD), every wave issues iters x 24 v_mfma_f32_32x32x16_f16
// on 8 independent accumulators; operands never leave the registers. mode 0: random fp16 operands; mode 1: zeros.
#include "common.h"

namespace morig {

typedef float ub_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ub_f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void ubench_mfma_kernel(int iters, float* out, unsigned seed) {
    unsigned s = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 1u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    ub_f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        union { unsigned u[4]; ub_f16x8 v; } ua, ub;
        for (int j = 0; j < 4; ++j) {
            // fp16 pairs with exponents around 1 (no inf / nan): random sign and mantissa
            const unsigned r = rnd(), q = rnd();
            ua.u[j] = MODE == 1 ? 0u : ((r & 0x83ff83ffu) | 0x34003400u | ((r >> 3) & 0x0c000c00u));
            ub.u[j] = MODE == 1 ? 0u : ((q & 0x83ff83ffu) | 0x34003400u | ((q >> 3) & 0x0c000c00u));
        }
        a[i] = ua.v; b[i] = ub.v;
    }
    ub_f32x16 c[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + g) & 3], b[i & 3], c[i], 0, 0, 0);
        if ((it & 63) == 63) for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] *= 1e-3f;     // bounded accumulators
    }
    float acc = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc += c[i][r];
    if (acc == 12345.678f) out[0] = acc;
}

}  // namespace morig

using namespace morig;

// flops of one call = 2 * 32 * 32 * 16 * 24 * iters * 8 waves * workgroups * launches (returned through *flops when not NULL)
extern "C" int morig_ubench_mfma(int mode, int iters, int launches, float* scratch, double* flops, void* stream) {
    if (iters <= 0 || launches <= 0 || !scratch) return MORIG_E_INVALID;
    int dev = 0, ncu = 256;
    MORIG_HIP_TRY(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int l = 0; l < launches; ++l) {
        if (mode == 1) hipLaunchKernelGGL(ubench_mfma_kernel<1>, dim3(ncu), dim3(512), 0, s, iters, scratch, 17u + l);
        else           hipLaunchKernelGGL(ubench_mfma_kernel<0>, dim3(ncu), dim3(512), 0, s, iters, scratch, 17u + l);
    }
    MORIG_LAUNCH_CHECK();
    if (flops) *flops = 2.0 * 32 * 32 * 16 * 24.0 * iters * 8.0 * ncu * launches;
    return MORIG_OK;
}
