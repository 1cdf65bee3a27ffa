// Micro-benchmark: cycles per v_mfma_f32_32x32x16_f16 on one SIMD as a function of the DEPENDENCY DISTANCE between MFMAs that
// accumulate into the same registers, with one and with two waves per SIMD (the two-wave case is what the 512-thread kernels run).
// Patterns (48 MFMAs per loop trip, NACC accumulators):
//   rr<D>   : round robin over D accumulators (distance D)                      -- D = 1, 2, 3, 4, 8
//   tri<D>  : three in a row on one accumulator, then the next of D accumulators -- the split-fp16 product as written
//             (lo*hi, hi*lo, hi*hi into acc[mt][nt])
//   pair    : A B A B A B over 2 accumulators, then the next pair (mma_pair of gemm_dmap.hip, mma of edge_ws.hip)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_dep mfma_dep.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int PAT, int D>
__global__ __launch_bounds__(512) void k(int iters, float* out, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a[4], b[4];
    for (int f = 0; f < 4; ++f)
        for (int q = 0; q < 8; ++q) { a[f][q] = (_Float16)(lane * 0.001f + q + f); b[f][q] = (_Float16)(lane * 0.002f - q - f); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 48; ++u) {
            int i;
            if (PAT == 0) i = u % D;                        // round robin
            else if (PAT == 1) i = (u / 3) % D;             // three in a row
            else i = 2 * ((u / 6) % 4) + (u & 1);           // pairs
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u >> 2) & 3], acc[i], 0, 0, 0);
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sink = 0.f;
    for (int i = 0; i < 8; ++i) sink += acc[i][0];
    if (sink == 12345.f) out[threadIdx.x] = sink;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int PAT, int D>
void run(const char* name, float* out, unsigned long long* clk) {
    const int iters = 4000;
    double r[2];
    for (int w = 0; w < 2; ++w) {                           // 256 threads = one wave per SIMD, 512 = two
        const int threads = w ? 512 : 256;
        hipLaunchKernelGGL((k<PAT, D>), dim3(256), dim3(threads), 0, 0, 50, out, clk);
        hipLaunchKernelGGL((k<PAT, D>), dim3(256), dim3(threads), 0, 0, iters, out, clk);
        hipDeviceSynchronize();
        unsigned long long h = 0;
        hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        r[w] = (double)h / (48.0 * iters * (w ? 2 : 1));    // cycles per MFMA issued on the SIMD
    }
    printf("%-10s 1 wave/SIMD %6.1f cyc per MFMA | 2 waves/SIMD %6.1f cyc per MFMA (SIMD-level; 32.0 = the pipe's rate)\n", name, r[0], r[1]);
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&clk, 64); hipMalloc(&out, 4096);
    run<0, 1>("rr<1>", out, clk);
    run<0, 2>("rr<2>", out, clk);
    run<0, 3>("rr<3>", out, clk);
    run<0, 4>("rr<4>", out, clk);
    run<0, 8>("rr<8>", out, clk);
    run<1, 2>("tri<2>", out, clk);
    run<1, 4>("tri<4>", out, clk);
    run<1, 8>("tri<8>", out, clk);
    run<2, 8>("pair", out, clk);
    return 0;
}
