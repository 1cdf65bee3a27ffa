// Micro-benchmark: per-CU store throughput of the GEMM copy-out patterns (8 waves per CU, 16-byte stores).
//   A: split layout as written today -- per row, 64-byte hi pieces by one instruction, the 64-byte lo pieces by the next
//   B: same bytes, every instruction writes full 128-byte lines
//   C: fp32 layout (one instruction = 2 rows x 512 B contiguous)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(float* Y, int ldy /*floats*/, int tiles_n, int reps, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    char* base = reinterpret_cast<char*>(Y + (size_t)(tm * 256 + wm * 64) * ldy + tn * 256 + wn * 128);
    const size_t ldb = (size_t)ldy * 4;
    f32x4 v = {lane * 1.f, 2.f, 3.f, wave * 1.f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep)
        for (int slab = 0; slab < 2; ++slab)
            for (int pass = 0; pass < 8; ++pass) {
                const int r0 = slab * 32 + pass * 4;
                if (PAT == 0) {
                    char* a = base + (size_t)(r0 + (lane >> 4)) * ldb + ((lane & 15) >> 2) * 128 + (lane & 3) * 16;
                    *reinterpret_cast<f32x4*>(a) = v;
                    *reinterpret_cast<f32x4*>(a + 64) = v;
                } else {
                    char* a = base + (size_t)(r0 + (lane >> 5)) * ldb + (lane & 31) * 16;
                    *reinterpret_cast<f32x4*>(a) = v;
                    *reinterpret_cast<f32x4*>(a + 2 * ldb) = v;
                }
                v[1] += 1.f;
            }
    if (blockIdx.x == 40 && threadIdx.x == 0) clk[0] = __builtin_readcyclecounter() - t0;
}

int main(int argc, char** argv) {
    const int M = 327680, N = 1024;
    float* Y; unsigned long long* clk;
    hipMalloc(&Y, (size_t)M * N * 4); hipMalloc(&clk, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nblk : {256, 5120})
        for (int pat = 0; pat < 2; ++pat) {
            float best = 1e9f; unsigned long long c = 0;
            for (int t = 0; t < 5; ++t) {
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(512), 0, 0, Y, N, 4, 1, clk);
                else hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(512), 0, 0, Y, N, 4, 1, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
            }
            const double bytes = (double)nblk * 256 * 256 * 4;
            printf("blocks %5d pattern %c: %.1f us  %.2f TB/s  block-40 cycles %llu (%.0f per pass)\n", nblk, "AB"[pat], best * 1e3, bytes / best / 1e9, c, c / 16.0);
        }
    return 0;
}
