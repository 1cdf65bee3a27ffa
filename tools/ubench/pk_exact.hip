// Are v_pk_add_f32 / v_pk_mul_f32 bit-identical to the scalar ops? (sqdist3 on random inputs, scalar vs packed)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sq(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
__global__ void k(const float* a, const float* b, int n, unsigned* out, int* bad) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n) return;
    const float tx = b[0], ty = b[1], tz = b[2];
    const float s0 = sq(a[3 * i], a[3 * i + 1], a[3 * i + 2], tx, ty, tz), s1 = sq(a[3 * i + 3], a[3 * i + 4], a[3 * i + 5], tx, ty, tz);
    f2 d;
    {
#pragma clang fp contract(off)
        const f2 X = {a[3 * i], a[3 * i + 3]}, Y = {a[3 * i + 1], a[3 * i + 4]}, Z = {a[3 * i + 2], a[3 * i + 5]};
        const f2 T0 = {tx, tx}, T1 = {ty, ty}, T2 = {tz, tz};
        const f2 dx = X - T0, dy = Y - T1, dz = Z - T2;
        d = (dx * dx + dy * dy) + dz * dz;
    }
    if (__float_as_uint(d[0]) != __float_as_uint(s0) || __float_as_uint(d[1]) != __float_as_uint(s1)) {
        const int slot = atomicAdd(bad, 1);
        if (slot < 4) { out[4 * slot] = __float_as_uint(d[0]); out[4 * slot + 1] = __float_as_uint(s0); out[4 * slot + 2] = __float_as_uint(d[1]); out[4 * slot + 3] = __float_as_uint(s1); }
    }
}
int main() {
    const int n = 1 << 22;
    float* h = (float*)malloc(n * 3 * sizeof(float));
    srand(1);
    for (int i = 0; i < 3 * n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *a, *b; unsigned* out; int* bad;
    hipMalloc(&a, n * 3 * 4); hipMalloc(&b, 16); hipMalloc(&out, 64); hipMalloc(&bad, 4);
    hipMemcpy(a, h, n * 3 * 4, hipMemcpyHostToDevice); hipMemcpy(b, h + 7, 12, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
    k<<<n / 2 / 256, 256>>>(a, b, n, out, bad);
    int nb; unsigned o[16];
    hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(o, out, 64, hipMemcpyDeviceToHost);
    printf("pairs with a mismatch: %d of %d\n", nb, n / 2);
    for (int s = 0; s < (nb < 4 ? nb : 4); ++s) printf("  pk %08x scalar %08x | pk %08x scalar %08x\n", o[4 * s], o[4 * s + 1], o[4 * s + 2], o[4 * s + 3]);
    return 0;
}
