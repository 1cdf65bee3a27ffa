// Ground truth of v_mfma_f32_32x32x16_f16 operand / result layouts: A[i][k] = (i == k), B[k][j] = 100 k + j  ->  D[i][j] = 100 i + j (i < 16)
// Prints, for a few lanes, the 16 result registers. Build: hipcc --offload-arch=gfx950 mfma_layout.hip -o mfma_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, int swap) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) {
        const int kk = 8 * hi + q;                       // this lane's k-slice
        a[q] = (_Float16)(l31 == kk ? 1.f : 0.f);        // A[i = l31][kk] = delta
        b[q] = (_Float16)(100.f * kk + l31);             // B[kk][j = l31]  (exact in fp16 up to 2048)
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (swap) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0);
    else      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 16 * 4);
    float h[64 * 16];
    for (int swap = 0; swap < 2; ++swap) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, swap);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("operands %s\n", swap ? "(b, a): D' = B^T-ish" : "(a, b)");
        for (int lane : {0, 1, 5, 32, 33}) {
            printf(" lane %2d:", lane);
            for (int r = 0; r < 16; ++r) printf(" %6.0f", h[lane * 16 + r]);
            printf("\n");
        }
    }
    return 0;
}
