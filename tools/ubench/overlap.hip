// Micro-benchmark: do MFMA (one wave per SIMD) and VALU / LDS / VMEM work (another wave on the same SIMD) overlap on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// mode bits: 1 = waves 0-3 run MFMA loop, 2 = waves 4-7 run VALU loop, 4 = waves 4-7 run LDS-write loop,
//            8 = waves 4-7 run global-load loop, 16 = barrier every 48 MFMAs / per producer iteration
template <bool AG>
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, const float* src, float* out, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sink = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if ((mode & 32) && wave >= 4) __builtin_amdgcn_s_setprio(3);
    if ((mode & 64) && wave < 4) __builtin_amdgcn_s_setprio(3);
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc[8];
            for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            f16x8 a, b;
            for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(lane * 0.001f + q); b[q] = (_Float16)(lane * 0.002f - q); }
            const float* cgp = src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
            f32x4 cg[12];
            for (int u = 0; u < 12; ++u) cg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < iters; ++it) {
                if (mode & 512) { for (int u = 0; u < 12; ++u) a[u & 7] += (_Float16)cg[u][0]; }
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if ((i & 3) == 0) {
                            const int q = u * 2 + (i >> 2);
                            const float* ga = cgp + (size_t)((it * 12 + q) & 1023) * 1024 * 64;
                            if (mode & 512) cg[q] = *reinterpret_cast<const f32x4*>(ga);
                            if (mode & 1024) __builtin_amdgcn_global_load_lds((glb_void_t*)ga, (lds_void_t*)(lds + wave * 12288 + q * 1024), 16, 0, 0);
                        }
                        if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                    }
                if (mode & 16) __syncthreads();
            }
            for (int i = 0; i < 8; ++i) sink += acc[i][0];
        } else if (mode & 16) {
            for (int it = 0; it < iters; ++it) __syncthreads();
        }
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = lane * 0.5f + i;
        const float* gp = src + (size_t)(blockIdx.x * 256 + (threadIdx.x - 256)) * 4;
        for (int it = 0; it < iters; ++it) {
            if (mode & 2) {
#pragma unroll
                for (int u = 0; u < 12; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i] * 1.0001f + 0.5f, v[(i + 1) & 15]);   // 2 VALU each -> 384 VALU
            }
            if (mode & 4) {
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    f32x4 w = {v[0], v[1], v[2], v[3] + u};
                    *reinterpret_cast<f32x4*>(lds + ((threadIdx.x - 256) * 16 + u * 4096)) = w;
                }
            }
            if (mode & 256) {
#pragma unroll
                for (int u = 0; u < 12; ++u)
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(gp + (size_t)((it * 12 + u) & 1023) * 1024 * 64), (lds_void_t*)(lds + (wave - 4) * 12288 + u * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (mode & 8) {
                f32x4 g[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) g[u] = *reinterpret_cast<const f32x4*>(gp + (size_t)((it * 12 + u) & 1023) * 1024 * 64);
#pragma unroll
                for (int u = 0; u < 12; ++u) v[u] += g[u][0];
            }
            if (mode & 16) __syncthreads();
        }
        for (int i = 0; i < 16; ++i) sink += v[i];
        if (mode & (4 | 256 | 1024)) sink += lds[threadIdx.x];
    }
    if (sink == 12345.f) out[threadIdx.x] = sink;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) {
        clk[2 * (threadIdx.x >> 8)] = __builtin_readcyclecounter() - t0;
        clk[2 * (threadIdx.x >> 8) + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

int main() {
    float *src, *out; unsigned long long* clk; hipMalloc(&clk, 64); unsigned long long hclk[4];
    hipMalloc(&src, (size_t)1024 * 1024 * 64 * 4 + (1 << 24)); hipMalloc(&out, 4096);
    hipMemset(src, 0, (size_t)1024 * 1024 * 64 * 4 + (1 << 24));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int modes[] = {1, 8, 9, 256, 257, 513, 1025, 2, 515, 1027};
    const int iters = 2000;
    for (int m : modes) {
        if (m & 128) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), 0, 0, m, 10, src, out, clk);
        else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 0, 0, m, 10, src, out, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (m & 128) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), 0, 0, m, iters, src, out, clk);
        else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 0, 0, m, iters, src, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hclk, clk, 32, hipMemcpyDeviceToHost);
        printf("[cons %llu cyc %.2f GHz | prod %llu cyc %.2f GHz] ", hclk[0], hclk[0] / (hclk[1] * 10.0), hclk[2], hclk[2] / (hclk[3] * 10.0));
        printf("mode %2d [%s%s%s%s%s]: %.3f ms  -> %.0f cycles/iter @2.4GHz\n", m, (m & 1) ? "MFMA " : "", (m & 2) ? "VALU " : "",
               (m & 4) ? "LDSW " : "", (m & 8) ? "VMEM " : (m & 256) ? "pDMA " : (m & 512) ? "cLOAD " : (m & 1024) ? "cDMA " : "", (m & 16) ? "BAR " : "", ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
