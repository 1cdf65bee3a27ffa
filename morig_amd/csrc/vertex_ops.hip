// Small vertex-parallel operators of the forward path: HBM-bound, one pass each.
#include "common.h"

namespace morig {

// one wave per row -- two rows per wave when a row has at most 32 columns (each half-wave reduces on its own: the same partial
// sums in the same order as a whole wave whose upper half holds zeros); y = x / max(||x||, 1e-12)
// (torch.nn.functional.normalize, p=2, eps=1e-12)
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, int ldx, int rows_per_rep, int reps, int cols,
                                                      float* __restrict__ y, int ld_row, int ld_rep) {
    const int wave = threadIdx.x >> 6, l64 = threadIdx.x & 63;
    const int two = cols <= 32 ? 1 : 0;
    const int lane = two ? (l64 & 31) : l64, half = two ? (l64 >> 5) : 0, step = two ? 32 : 64, rpw = two ? 2 : 1;
    const int64_t total = (int64_t)rows_per_rep * reps;
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6) * rpw;
    for (int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * rpw; base < total; base += wstride) {   // wave-uniform
        const int64_t m = base + half;
        const bool live = m < total;
        const float* xr = x + (live ? m : 0) * ldx;
        float ss = 0.f;
        if (live)
            for (int c = lane; c < cols; c += step) { const float v = xr[c]; ss += v * v; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            if (!(two && o == 32)) ss += __shfl_xor(ss, o, 64);
        if (!live) continue;
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        const int r = (int)(m / rows_per_rep);
        const int64_t v = m - (int64_t)r * rows_per_rep;
        float* yr = y + v * ld_row + (int64_t)r * ld_rep;
        for (int c = lane; c < cols; c += step) yr[c] = xr[c] * inv;
    }
}

// The same normalisation for 16-byte aligned rows of cols in {32, 64}: a lane owns 4 adjacent columns (one 16-byte load and store),
// cols / 4 lanes share a row, so a wave moves 8 (4) rows per instruction with two row groups in flight -- the one-element-per-lane
// form above keeps a single 256-byte request per wave outstanding and reached 26 % of the HBM rate on the 1.3 M x 32 launch.
template <int COLS>
__global__ __launch_bounds__(256) void rownorm4_kernel(const float* __restrict__ x, int ldx, int rows_per_rep, int reps,
                                                       float* __restrict__ y, int ld_row, int ld_rep) {
    constexpr int LPR = COLS / 4, RPW = 64 / LPR;                  // lanes per row, rows per wave instruction
    const int wave = threadIdx.x >> 6, l64 = threadIdx.x & 63;
    const int sub = l64 / LPR, c4 = (l64 % LPR) * 4;
    const int64_t total = (int64_t)rows_per_rep * reps;
    const int64_t wstride = (int64_t)gridDim.x * (blockDim.x >> 6) * RPW * 2;
    for (int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * RPW * 2; base < total; base += wstride) {   // wave-uniform
        float4 v[2]; bool live[2]; float ss[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t m = base + u * RPW + sub;
            live[u] = m < total;
            v[u] = live[u] ? *reinterpret_cast<const float4*>(x + m * ldx + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ss[u] = (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) ss[u] += __shfl_xor(ss[u], o, 64);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live[u]) continue;
            const int64_t m = base + u * RPW + sub;
            const float inv = 1.0f / fmaxf(sqrtf(ss[u]), 1e-12f);
            const int r = (int)(m / rows_per_rep);
            const int64_t vtx = m - (int64_t)r * rows_per_rep;
            *reinterpret_cast<float4*>(y + vtx * ld_row + (int64_t)r * ld_rep + c4) = make_float4(v[u].x * inv, v[u].y * inv, v[u].z * inv, v[u].w * inv);
        }
    }
}

// CLS-only temporal attention (see include/morig_hip.h). A 256-thread block owns 64 vertices: their frames are
// loaded coalesced into LDS (row stride T*C+1 -> conflict-free per-vertex reads), thread (vertex, head) does the
// (T+1)-way softmax and the weighted token sum, results leave through LDS so the stores are coalesced too.
constexpr int ATT_TMAX = 8;
constexpr int ATT_VB = 64;
__global__ __launch_bounds__(256) void cls_attention_kernel(const float* __restrict__ x, int n, int T, int C, int heads,
                                                            const float* __restrict__ g, const float* __restrict__ cls,
                                                            float* __restrict__ y, int ldy) {
    extern __shared__ float sh[];               // g [heads*C] | cls [C] | frames [VB][T*C+1] | out [VB][heads*C+1]
    const int TC = T * C, HC = heads * C;
    float* sg = sh;
    float* sc = sg + HC;
    float* sx = sc + C;
    float* so = sx + ATT_VB * (TC + 1);
    const int v0 = blockIdx.x * ATT_VB;
    const int nv = min(ATT_VB, n - v0);
    for (int i = threadIdx.x; i < HC; i += blockDim.x) sg[i] = g[i];
    for (int i = threadIdx.x; i < C; i += blockDim.x) sc[i] = cls[i];
    const float* xb = x + (size_t)v0 * TC;
    for (int i = threadIdx.x; i < nv * TC; i += blockDim.x) { const int v = i / TC, c = i - v * TC; sx[v * (TC + 1) + c] = xb[i]; }
    __syncthreads();
    for (int w = threadIdx.x; w < nv * heads; w += blockDim.x) {
        const int v = w % nv, h = w / nv;       // consecutive lanes -> consecutive vertices
        const float* gh = sg + h * C;
        const float* xv = sx + v * (TC + 1);
        float s[ATT_TMAX + 1];
        float s0 = 0.f;
        for (int c = 0; c < C; ++c) s0 += sc[c] * gh[c];
        s[0] = s0;
        float mx = s0;
#pragma unroll
        for (int t = 0; t < ATT_TMAX; ++t) {
            float d = -INFINITY;
            if (t < T) {
                d = 0.f;
                for (int c = 0; c < C; ++c) d += xv[t * C + c] * gh[c];
            }
            s[t + 1] = d;
            mx = fmaxf(mx, d);
        }
        float den = 0.f;
#pragma unroll
        for (int t = 0; t <= ATT_TMAX; ++t) { s[t] = (t <= T) ? __expf(s[t] - mx) : 0.f; den += s[t]; }
        const float inv = 1.0f / den;
        float* yo = so + v * (HC + 1) + h * C;
        for (int c = 0; c < C; ++c) {
            float a = s[0] * sc[c];
#pragma unroll
            for (int t = 0; t < ATT_TMAX; ++t) if (t < T) a += s[t + 1] * xv[t * C + c];
            yo[c] = a * inv;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nv * HC; i += blockDim.x) { const int v = i / HC, c = i - v * HC; y[(size_t)(v0 + v) * ldy + c] = so[v * (HC + 1) + c]; }
}

// The same operator for C = 32 channels (the networks' motion feature) without LDS [r06]: 8 lanes share a vertex, a lane owns 4 adjacent
// channels of every frame (16-byte loads: a wave instruction moves 8 whole 128-byte rows), the T + 1 scores per head are 4-channel partial
// dot products reduced over the 8 lanes (3 exchange steps), softmax and the weighted token sum stay in registers, 16-byte stores. The
// LDS form above spends its time in index arithmetic (a division per staged element) and in (vertex, head) threads that leave half of
// the block idle at 2 heads: 260 us for the headline's 262 144 vertices, 0.9 TB/s of the operator's 235 MB.
template <int TMAX, int HMAX>
__global__ __launch_bounds__(256) void cls_attention_c32_kernel(const float* __restrict__ x, int n, int T, int heads,
                                                                const float* __restrict__ g, const float* __restrict__ cls,
                                                                float* __restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63, q = lane & 7, vs = lane >> 3;
    auto dot4 = [](const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); };
    auto sum8 = [](float v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); return v; };
    const float4 c4 = *reinterpret_cast<const float4*>(cls + 4 * q);
    float4 g4[HMAX];
    float s0[HMAX];
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        g4[h] = h < heads ? *reinterpret_cast<const float4*>(g + h * 32 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        s0[h] = sum8(dot4(c4, g4[h]));
    }
    const int64_t waves = (int64_t)gridDim.x * 4;
    for (int64_t vb = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; vb < n; vb += waves * 8) {        // wave-uniform
        const int64_t v = vb + vs;
        const bool live = v < n;
        const float* xv = x + (live ? v : 0) * T * 32 + 4 * q;
        float4 xt[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) xt[t] = t < T ? *reinterpret_cast<const float4*>(xv + t * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            if (h >= heads) break;                                    // uniform
            float sc[TMAX];
            float mx = s0[h];
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t >= T) break;
                sc[t] = sum8(dot4(xt[t], g4[h]));
                mx = fmaxf(mx, sc[t]);
            }
            const float p0 = __expf(s0[h] - mx);
            float den = p0;
            float4 a = make_float4(p0 * c4.x, p0 * c4.y, p0 * c4.z, p0 * c4.w);
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t >= T) break;
                const float pt = __expf(sc[t] - mx);
                den += pt;
                a.x += pt * xt[t].x; a.y += pt * xt[t].y; a.z += pt * xt[t].z; a.w += pt * xt[t].w;
            }
            const float inv = 1.0f / den;
            if (live) *reinterpret_cast<float4*>(y + v * ldy + h * 32 + 4 * q) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
        }
    }
}

__global__ void frame_reduce_kernel(const float* __restrict__ x, int n, int T, int C, int mode, float* __restrict__ y, int ldy) {
    const int64_t total = (int64_t)n * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t v = i / C; const int c = (int)(i - v * C);
        const float* xv = x + v * T * C + c;
        float a = xv[0];
        for (int t = 1; t < T; ++t) { const float b = xv[t * C]; a = mode ? fmaxf(a, b) : a + b; }
        y[v * ldy + c] = mode ? a : a / (float)T;
    }
}

// Dense layer on a FEW rows (M <= 128: one row per mesh and keyframe -- the Linear behind the pooled global feature that produces the
// row bias of `mlp_transform`, models/rignet.py:60-63 / models/corrnet.py:66-68; K <= 1024). The 128-row tile engine runs such a launch as
// 8 ... 24 workgroups walking 32 K-chunks each, a chain of dependent load -> LDS -> barrier -> MFMA steps: 50 us whether M is 2 or 320
// (profiles/r06z_timeline_B1.txt: two of them are 5 % of the one-mesh forward). Here the WEIGHTS are what is distributed: a workgroup
// owns NC = 4 output columns, every lane keeps its K-slice of those 4 rows of W in registers (lane l: columns 4 l .. 4 l + 3 of every
// 256-column block), and a wave takes rows m = wave, wave + 4, ... eight at a time (32 independent 16-byte loads in flight), multiplies in
// fp32 FMAs and reduces over its lanes in a fixed order (deterministic). N / 4 workgroups: 256 for the 1024-wide layers. Plain float32
// arithmetic -- no fp16 conversion, hence no range guard to report to. Measured (profiles/r07b_*): M = 2 ... 64 rows 7 ... 27 us against the
// tile engine's 50; M = 320 115 us (a wave walks 80 rows, six exchange steps per row and column) -- hence the 128-row limit.
template <int NC, int KJ, int RB>
__global__ __launch_bounds__(256) void few_rows_gemm_kernel(const float* __restrict__ X, int ldx, int M, const float* __restrict__ W, int ldw,
                                                            int N, int K, const float* __restrict__ bias, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu, float* __restrict__ Y, int ldy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * NC;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w[NC][KJ];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const int k = 4 * lane + 256 * j;
            w[c][j] = (n0 + c < N && k < K) ? *reinterpret_cast<const float4*>(W + (size_t)(n0 + c) * ldw + k) : zero4;
        }
    // lane c < NC finishes column n0 + c
    const int nc = n0 + (lane < NC ? lane : 0);
    const bool owner = lane < NC && nc < N;
    const float b = (owner && bias) ? bias[nc] : 0.f, sc = (owner && scale) ? scale[nc] : 1.f, sh = (owner && shift) ? shift[nc] : 0.f;
    for (int m0 = wave; m0 < M; m0 += 4 * RB) {                       // wave-uniform
        float4 x[RB][KJ];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int m = m0 + 4 * u;
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int k = 4 * lane + 256 * j;
                x[u][j] = (m < M && k < K) ? *reinterpret_cast<const float4*>(X + (size_t)m * ldx + k) : zero4;
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int m = m0 + 4 * u;
            if (m >= M) break;                                        // wave-uniform
            float mine = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    a = fmaf(x[u][j].x, w[c][j].x, a); a = fmaf(x[u][j].y, w[c][j].y, a);
                    a = fmaf(x[u][j].z, w[c][j].z, a); a = fmaf(x[u][j].w, w[c][j].w, a);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                if (lane == c) mine = a;
            }
            if (owner) {
                float v = mine + b;
                if (relu) v = v > 0.f ? v : 0.f;
                Y[(size_t)m * ldy + nc] = v * sc + sh;
            }
        }
    }
}

bool few_rows_gemm_takes(int M, int N, int K, int ldx, int ldw) {
    static const bool off = getenv("MORIG_NO_FEW_ROWS") != nullptr;
    return !off && M <= 128 && N >= 128 && K >= 128 && K <= 1024 && (K & 3) == 0 && (ldx & 3) == 0 && (ldw & 3) == 0;
}

int launch_few_rows_gemm(const float* X, int ldx, int M, const float* W, int ldw, int N, int K, const float* bias, const float* scale,
                         const float* shift, int relu, float* Y, int ldy, hipStream_t s) {
    constexpr int NC = 4;
    hipLaunchKernelGGL((few_rows_gemm_kernel<NC, 4, 8>), dim3(cdiv(N, NC)), dim3(256), 0, s, X, ldx, M, W, ldw, N, K, bias, scale, shift, relu,
                       Y, ldy);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

}  // namespace morig

using namespace morig;

extern "C" int morig_rownorm(const float* x, int32_t ldx, int32_t rows_per_rep, int32_t replicas, int32_t cols,
                             float* y, int32_t ld_row, int32_t ld_rep, void* stream) {
    if (!x || !y || rows_per_rep <= 0 || replicas <= 0 || cols <= 0 || ldx < cols) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)rows_per_rep * replicas;
    int64_t blocks = (rows + 3) / 4;
    if (cols <= 32) blocks = (rows + 7) / 8;                         // two rows per wave
    if (blocks > 256 * 32) blocks = 256 * 32;
    ProfScope ps(K_ROWNORM, s, 3.0 * rows * cols, 8.0 * rows * cols);
    const bool v4 = (cols == 32 || cols == 64) && (ldx & 3) == 0 && (ld_row & 3) == 0 && (ld_rep & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    if (v4) {
        const int rpb = 4 * (64 / (cols / 4)) * 2;                   // rows per block and iteration
        int64_t b4 = (rows + rpb - 1) / rpb;
        if (b4 > 256 * 16) b4 = 256 * 16;
        if (cols == 32) hipLaunchKernelGGL(rownorm4_kernel<32>, dim3((int)b4), dim3(256), 0, s, x, ldx, rows_per_rep, replicas, y, ld_row, ld_rep);
        else            hipLaunchKernelGGL(rownorm4_kernel<64>, dim3((int)b4), dim3(256), 0, s, x, ldx, rows_per_rep, replicas, y, ld_row, ld_rep);
        MORIG_LAUNCH_CHECK();
        return MORIG_OK;
    }
    hipLaunchKernelGGL(rownorm_kernel, dim3((int)blocks), dim3(256), 0, s, x, ldx, rows_per_rep, replicas, cols, y, ld_row, ld_rep);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_cls_attention(const float* x, int32_t n, int32_t T, int32_t C, int32_t heads,
                                   const float* g, const float* cls, float* y, int32_t ldy, void* stream) {
    if (!x || !g || !cls || !y || n <= 0 || C <= 0 || heads <= 0 || ldy < heads * C) return MORIG_E_INVALID;
    if (T < 1 || T > ATT_TMAX) return MORIG_E_UNSUPPORTED;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope ps(K_ATTN, s, 4.0 * n * heads * (T + 1) * C, 4.0 * n * (T * C + heads * C));
    static const bool no_c32 = getenv("MORIG_ATTN_LDS") != nullptr;
    if (C == 32 && heads <= 4 && !no_c32 && (ldy & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(cls)) & 15) == 0) {
        int blocks = cdiv(n, 32);
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL((cls_attention_c32_kernel<ATT_TMAX, 4>), dim3(blocks), dim3(256), 0, s, x, n, T, heads, g, cls, y, ldy);
        MORIG_LAUNCH_CHECK();
        return MORIG_OK;
    }
    const size_t lds = ((size_t)(heads + 1) * C + (size_t)ATT_VB * (T * C + 1) + (size_t)ATT_VB * (heads * C + 1)) * sizeof(float);
    if (lds > 64 * 1024) return MORIG_E_UNSUPPORTED;
    hipLaunchKernelGGL(cls_attention_kernel, dim3(cdiv(n, ATT_VB)), dim3(256), lds, s, x, n, T, C, heads, g, cls, y, ldy);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}

extern "C" int morig_frame_reduce(const float* x, int32_t n, int32_t T, int32_t C, int32_t mode, float* y, int32_t ldy, void* stream) {
    if (!x || !y || n <= 0 || T <= 0 || C <= 0 || ldy < C || (mode != 0 && mode != 1)) return MORIG_E_INVALID;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = ((int64_t)n * C + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ProfScope ps(K_MISC, s, (double)n * T * C, 4.0 * n * C * (T + 1));
    hipLaunchKernelGGL(frame_reduce_kernel, dim3((int)blocks), dim3(256), 0, s, x, n, T, C, mode, y, ldy);
    MORIG_LAUNCH_CHECK();
    return MORIG_OK;
}
