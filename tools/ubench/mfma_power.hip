// What does the matrix pipe cost in POWER? One persistent launch per call: 256 workgroups x 512 threads (2 waves per SIMD), every wave
// issues `iters` x 24 v_mfma_f32_32x32x16_f16 on 8 independent accumulators with operands that never leave the registers.
//   mode 0: operands = random fp16 bit patterns (hi-like: full-range mantissas)     mode 1: operands = zeros
//   mode 2: random operands + 12 ds_read_b128 per 24 MFMAs (the fragment traffic of the GEMM kernels; results ignored)
//   mode 3: random operands + 32 VALU (v_fma_f32) per 24 MFMAs
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/ubench/mfma_power.hip -o tools/ubench/libmfma_power.so ; driver: tools/mfma_power.py
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_power_kernel(int iters, float* out, unsigned seed) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x;
    for (int i = tid; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = (i * 2654435761u) ^ seed;
    __syncthreads();
    unsigned s = seed ^ (blockIdx.x * 9781u + tid * 6271u + 1u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        union { unsigned u[4]; f16x8 v; } ua, ub;
        for (int j = 0; j < 4; ++j) {
            // fp16 pairs with exponents around 1.0 (no inf / nan): sign random, exponent 13..16, mantissa random
            unsigned r = rnd(), q = rnd();
            ua.u[j] = MODE == 1 ? 0u : ((r & 0x83ff83ffu) | 0x34003400u | ((r >> 3) & 0x0c000c00u));
            ub.u[j] = MODE == 1 ? 0u : ((q & 0x83ff83ffu) | 0x34003400u | ((q >> 3) & 0x0c000c00u));
        }
        a[i] = ua.v; b[i] = ub.v;
    }
    f32x16 c[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    const char* lp = lds + (tid & 63) * 16;
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    float v0 = 1.0f + tid * 1e-6f, v1 = 0.5f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + g) & 3], b[i & 3], c[i], 0, 0, 0);
                if constexpr (MODE == 2) {
                    if ((i & 1) == 0) { const f32x4 t = *reinterpret_cast<const f32x4*>(lp + ((g * 8 + i) * 1024 & 0xffff)); sink += t; }
                }
                if constexpr (MODE == 3) {
                    if (i < 4) {   // 4 x 3 x ... ~ 1.3 VALU per MFMA
                        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
                        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
                        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
                    }
                }
            }
        }
        // keep the accumulators bounded (and the loop honest): every 64 iterations fold them
        if ((it & 63) == 63) for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] *= 1e-3f;
    }
    float acc = sink[0] + sink[1] + sink[2] + sink[3] + v0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc += c[i][r];
    if (acc == 12345.678f) out[0] = acc;
}

extern "C" int mfma_power_run(int mode, int iters, int launches, float* out) {
    for (int l = 0; l < launches; ++l) {
        switch (mode) {
            case 0: hipLaunchKernelGGL(mfma_power_kernel<0>, dim3(256), dim3(512), 0, 0, iters, out, 17u + l); break;
            case 1: hipLaunchKernelGGL(mfma_power_kernel<1>, dim3(256), dim3(512), 0, 0, iters, out, 17u + l); break;
            case 2: hipLaunchKernelGGL(mfma_power_kernel<2>, dim3(256), dim3(512), 0, 0, iters, out, 17u + l); break;
            default: hipLaunchKernelGGL(mfma_power_kernel<3>, dim3(256), dim3(512), 0, 0, iters, out, 17u + l); break;
        }
    }
    return (int)hipDeviceSynchronize();
}
