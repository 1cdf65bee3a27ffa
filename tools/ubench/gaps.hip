// Micro-benchmark: how large a bubble in a dense MFMA stream lets a sibling wave on the same SIMD issue VMEM?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int G, int NOP, int KIND>   // bubble after every G MFMAs; KIND 0: s_nop NOP, 1: one v_mov (VALU), 2: s_sleep 0? 3: setprio toggle
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, const float* src, float* out, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sink = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc[8];
            for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            f16x8 a, b;
            for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(lane * 0.001f + q); b[q] = (_Float16)(lane * 0.002f - q); }
            float dummy = lane;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 48; ++u) {
                    acc[u & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 7], 0, 0, 0);
                    if (G > 0 && (u % G) == G - 1) {
                        if (KIND == 0) asm volatile("s_nop %0" ::"n"(NOP));
                        else if (KIND == 1) asm volatile("v_mov_b32 %0, %0" : "+v"(dummy));
                        else if (KIND == 2) __builtin_amdgcn_s_sleep(NOP);
                        else if (KIND == 3) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(1); }
                    }
                }
            }
            for (int i = 0; i < 8; ++i) sink += acc[i][0];
            sink += dummy;
        }
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = lane * 0.5f + i;
        const float* gp = src + (size_t)(blockIdx.x * 256 + (threadIdx.x - 256)) * 4;
        for (int it = 0; it < iters; ++it) {
            if (mode & 2) {
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i] * 1.0001f + 0.5f, v[(i + 1) & 15]);
            }
            if (mode & 8) {
                f32x4 g[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) g[u] = *reinterpret_cast<const f32x4*>(gp + (size_t)((it * 12 + u) & 1023) * 1024 * 64);
#pragma unroll
                for (int u = 0; u < 12; ++u) v[u] += g[u][0];
            }
        }
        for (int i = 0; i < 16; ++i) sink += v[i];
    }
    if (sink == 12345.f) out[threadIdx.x] = sink;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) clk[threadIdx.x >> 8] = __builtin_readcyclecounter() - t0;
}

template <int G, int NOP, int KIND>
void run(const char* name, const float* src, float* out, unsigned long long* clk) {
    unsigned long long h[2], h1[2], h8[2];
    const int iters = 2000;
    auto go = [&](int mode, unsigned long long* r) {
        hipLaunchKernelGGL((k<G, NOP, KIND>), dim3(256), dim3(512), 0, 0, mode, 10, src, out, clk);
        hipLaunchKernelGGL((k<G, NOP, KIND>), dim3(256), dim3(512), 0, 0, mode, iters, src, out, clk);
        hipDeviceSynchronize();
        hipMemcpy(r, clk, 16, hipMemcpyDeviceToHost);
    };
    go(1, h1); go(8, h8); go(9, h);
    unsigned long long j[2], j2[2];
    go(10, j2); go(11, j);
    printf("%-22s MFMA alone %5.1f cyc/mfma | VMEM alone %5llu cyc/it | both: MFMA %5.1f, VMEM %5llu cyc/it | VALU+VMEM alone %5llu, with MFMA %5llu (MFMA %5.1f)\n", name,
           h1[0] / (48.0 * iters), h8[1] / iters, h[0] / (48.0 * iters), h[1] / iters, j2[1] / iters, j[1] / iters, j[0] / (48.0 * iters));
}

int main() {
    float *src, *out; unsigned long long* clk;
    hipMalloc(&clk, 64); hipMalloc(&src, (size_t)1024 * 1024 * 64 * 4 + (1 << 24)); hipMalloc(&out, 4096);
    hipMemset(src, 0, (size_t)1024 * 1024 * 64 * 4 + (1 << 24));
    run<0, 0, 0>("dense", src, out, clk);
    run<1, 0, 0>("nop0 every 1", src, out, clk);
    run<3, 0, 0>("nop0 every 3", src, out, clk);
    run<3, 3, 0>("nop3 every 3", src, out, clk);
    run<3, 7, 0>("nop7 every 3", src, out, clk);
    run<6, 7, 0>("nop7 every 6", src, out, clk);
    run<6, 15, 0>("nop15 every 6", src, out, clk);
    run<12, 15, 0>("nop15 every 12", src, out, clk);
    run<3, 0, 1>("v_mov every 3", src, out, clk);
    run<1, 0, 1>("v_mov every 1", src, out, clk);
    run<3, 0, 2>("sleep0 every 3", src, out, clk);
    run<6, 1, 2>("sleep1 every 6", src, out, clk);
    run<3, 0, 3>("setprio every 3", src, out, clk);
    return 0;
}
