/*
 * morig_hip.h -- C ABI of libmorig_hip.so: the MI355X (gfx950) kernels behind MoRig's
 * geometric-network forward path.
 *
 * Boundary contract:
 *   - plain `extern "C"` functions, plain pointers and sizes, no C++/torch types;
 *   - every buffer is DEVICE memory owned by the caller (PyTorch-ROCm allocates it); the library
 *     borrows pointers for the duration of the call and keeps nothing;
 *   - every call enqueues asynchronously on the given hipStream_t (passed as void*); no hidden
 *     device synchronisation (the profiling collector is the one documented exception);
 *   - every call returns 0 or a negative MORIG_E_* code; no exceptions, no abort().
 *
 * The reference has no FFI of its own: its native work is delegated to three un-vendored wheels.
 * Each entry point therefore names the reference CALL SITE (file:line under /root/reference) whose
 * third-party operator it replaces; INTEGRATION.md shows the Python binding a maintainer would add.
 *
 * All matrices are row-major fp32. "ld" = row stride in floats. Unless noted, ld and the column
 * offset of a matrix handed to a GEMM-type call must be multiples of 4 floats (16-byte rows).
 */
#ifndef MORIG_HIP_H
#define MORIG_HIP_H

#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): morig_gemm_args.w_split_format and morig_edgeconv_args.out_split appended to their structs; morig_gemm_tn_shift,
 * morig_edgeconv_can_split_out, morig_ubench_mfma added.
 * 3 (round 6): every argument struct STARTS with `uint32_t struct_size` (see below): appending a member no longer lets the library read
 * past the end of an older caller's struct. morig_gemm_args gained X_tail / ld_tail / tail_rows / tail_cols; morig_pack_tails added. */
#define MORIG_ABI_VERSION 3

/* Argument structs (morig_gemm_args, morig_edgeconv_args, morig_edgeconv_x3_args, morig_segmax_args, morig_pointconv_args):
 * the first member, struct_size, is sizeof() of the struct AS THE CALLER WAS COMPILED (MORIG_INIT_ARGS sets it and zeroes the rest).
 * Members are only ever appended. The library copies min(struct_size, its own sizeof) bytes into a zeroed struct of its own, so members
 * the caller does not know read as 0 / NULL; a struct_size below the ABI-3 size of the struct (MORIG_*_ARGS_V3_SIZE: what a version-1 / -2
 * caller's memory would spell there, or garbage) is MORIG_E_INVALID. */
#define MORIG_INIT_ARGS(type, var) type var; memset(&(var), 0, sizeof(var)); (var).struct_size = (uint32_t)sizeof(var)
#define MORIG_GEMM_ARGS_V3_SIZE        192u
#define MORIG_EDGECONV_ARGS_V3_SIZE    192u
#define MORIG_EDGECONV_X3_ARGS_V3_SIZE 168u
#define MORIG_SEGMAX_ARGS_V3_SIZE      144u
#define MORIG_POINTCONV_ARGS_V3_SIZE   176u
/* members appended since (a struct only grows at its end; a caller built against the smaller struct passes its own size and the library reads
 * the new members as 0): round 6 -- init_with / split_with / skip_init / skip_split of morig_edgeconv_args, init_with / skip_init of
 * morig_edgeconv_x3_args. sizeof of the CURRENT structs: */
#define MORIG_EDGECONV_ARGS_SIZE       216u
#define MORIG_EDGECONV_X3_ARGS_SIZE    184u

/* status codes */
#define MORIG_OK              0
#define MORIG_E_INVALID      -1   /* null pointer, negative size, misaligned ld ...            */
#define MORIG_E_UNSUPPORTED  -2   /* width / shape the kernels are not instantiated for        */
#define MORIG_E_HIP          -3   /* a HIP runtime call failed; see morig_last_hip_error()      */
#define MORIG_E_NODEVICE     -4   /* no gfx950 device visible                                   */

int         morig_abi_version(void);
/* Scheduling hint, process-wide: leave n CUs (rounded up to 8) free of the persistent EdgeConv workgroups launched from now
 * on. The host sets it while single-CU-per-cloud kernels (FPS) run on a second stream, and back to 0 afterwards. */
int         morig_reserve_cus(int n);
const char* morig_strerror(int status);
int         morig_last_hip_error(void);            /* raw hipError_t of the last MORIG_E_HIP   */
/* device facts used by the host side for roofline arithmetic; arch must start with "gfx950" */
int         morig_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz, char* arch, int arch_len);

/* --------------------------------------------------------------------------------------------
 * Graph preparation: COO -> self-loop-normalised CSR by destination.
 * Replaces, once per forward instead of once per conv:
 *   remove_self_loops + add_self_loops            models/basic_modules.py:149-150, 188-189
 *   the implicit sort-free scatter of MessagePassing.propagate(aggr='max')  (:151, :190)
 *
 * edge_index : int64 [2, n_edges] contiguous (row 0 = source j, row 1 = target i), as PyG batches it
 * rowptr     : int32 [n_nodes + 1]   out; rowptr[n_nodes] = E' = #(non-loop edges) + n_nodes
 * src_sorted : int32 [n_edges + n_nodes] out (capacity; entries beyond E' untouched)
 * dst_sorted : int32 [n_edges + n_nodes] out (target of every sorted edge, i.e. expanded rowptr)
 * cursor     : int32 [n_nodes + 1]   scratch
 * status     : int32 [1] out; set non-zero when an index is outside [0, n_nodes)
 * Order inside a target's segment is unspecified (max-aggregation is order-free => results are
 * bit-identical for any order).
 */
int morig_csr_build(const int64_t* edge_index, int64_t n_edges, int32_t n_nodes,
                    int32_t* rowptr, int32_t* src_sorted, int32_t* dst_sorted,
                    int32_t* cursor, int32_t* status, void* stream);
/* bipartite form used by PointConv (PyG applies remove_self_loops/add_self_loops(num_nodes =
 * min(N_src, N_dst)) to the raw (source, target) index pairs even though they index different sets):
 * sources in [0, n_src_nodes), targets in [0, n_nodes), n_src_nodes >= n_nodes; pairs with a negative
 * index (unused ball-query slots) are skipped with MORIG_CSR_SKIP_NEGATIVE. */
#define MORIG_CSR_SKIP_NEGATIVE 1   /* pairs with a negative index are skipped instead of flagged            */
#define MORIG_CSR_PAD4          2   /* every target's segment is padded to a multiple of 4 entries by repeating
                                       its self loop (max-aggregation is idempotent); capacity n_edges + 4*n_nodes.
                                       morig_edgeconv's `quad_aligned` fast epilogue requires it.              */
#define MORIG_CSR_MIN4          4   /* morig_csr_build_bipartite (not with PAD4) / morig_csr_build_dual: every segment of the PLAIN CSR holds at least 4 rows (a shorter one is filled
                                       up with copies of its self loop; capacity n_edges + 4*n_nodes). A quad of 4 consecutive rows then
                                       never holds more than two segments, which is what morig_edgeconv's `seg_min4` form needs: the
                                       H = 256 kernel reduces a quad that straddles two segments into two values instead of requiring
                                       4-ALIGNED segments (MORIG_CSR_PAD4 costs 9-14 % extra rows on the rig graphs, this ~0).           */
int morig_csr_build_bipartite(const int64_t* edge_index, int64_t n_edges, int32_t n_src_nodes, int32_t n_nodes,
                              int32_t flags, int32_t* rowptr, int32_t* src_sorted, int32_t* dst_sorted,
                              int32_t* cursor, int32_t* status, void* stream);

/* --------------------------------------------------------------------------------------------
 * Arithmetic. Default: fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 products, k-ordered fma).
 * Split-fp16 fast path (selected per call by passing W_split): each fp32 operand x is split on the fly
 * into hi = fp16(x), lo = fp16(x - hi) and the product evaluated as hi*hi + hi*lo + lo*hi on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation -- fp16 products are exact in fp32 and the omitted
 * lo*lo term is 2^-22 relative, so the result is fp32-class (measured ~1e-6 of scale) at 5.3x the
 * fp32-MFMA rate. Weights are pre-split by the host: every 32-float chunk of a packed weight row is
 * replaced by [32 halves hi | 32 halves lo] (same bytes, same strides). Operands must stay inside the
 * fp16 range (|x| < 65000); otherwise the kernel raises *overflow and the caller must redo the layer
 * on the fp32 path (morig_amd.models does this for the whole forward).
 *
 * Fused dense layer:  Y = scale * act(X * W^T + bias + rowbias[seg[row]]) + shift
 *                     and/or  P[seg[row]] = max over rows (column-wise)            (pool != NULL)
 * Replaces nn.Linear -> ReLU -> BatchNorm1d(eval) chains of MLP()   models/basic_modules.py:31-36
 * and, with `pool`, scatter_max(x, batch) + repeat_interleave       models/rignet.py:62-64, 175-179,
 *                                                                   models/corrnet.py:43-45
 * (the broadcast is never materialised: the pooled vector enters the next layer as `rowbias`).
 */
#define MORIG_SPLIT_F16 0
#define MORIG_SPLIT_BF16 1
#define MORIG_SPLIT_BF16X6 2   /* [ABI 3] W_split == NULL: plain fp32 X and W, both split into three bf16 limbs IN the kernel, six MFMAs per
                                  product: float32-class results (dropped terms <= 2^-24 relative), float32's exponent range (no range word),
                                  2.7x the fp32-MFMA rate. What the train-mode forward and the exact re-run behind the range guard take
                                  instead of v_mfma_f32_32x32x2_f32. Plain stores and pooled launches. */
typedef struct morig_gemm_args {
    uint32_t struct_size;     /* sizeof(morig_gemm_args) of the caller's build (ABI 3)                */
    int32_t M, N, K;          /* logical sizes: Y is M x N, X is M x K                            */
    const float* X; int32_t ldx;
    const float* W; int32_t ldw;     /* packed weights [Npad][ldw]: row n = output channel n, ldw >= roundup(K,32),
                                        zero padded; Npad = N rounded up to the column tile (32/64/128) */
    const float* bias;        /* [Npad] or NULL                                                   */
    const float* scale;       /* [Npad] or NULL (=1): BatchNorm eval scale  gamma/sqrt(var+eps)   */
    const float* shift;       /* [Npad] or NULL (=0): BatchNorm eval shift  beta - mean*scale     */
    int32_t relu;             /* 1: act = ReLU, 0: identity                                       */
    const float* rowbias; int32_t ld_rowbias;   /* optional [n_seg][ld_rowbias] added before act  */
    const int32_t* seg;       /* [M] segment (mesh) id of every row; required with rowbias / pool */
    float* Y; int32_t ldy;    /* optional output                                                  */
    float* pool; int32_t ld_pool; int32_t n_seg;  /* optional [n_seg][ld_pool] column max per segment; a segment without rows
                                                   * receives 0 (torch_scatter's fill) */
    /* optional fast path (see "split-fp16" below): W_split has the shape/stride of W; overflow is an int32
     * on the device that the kernel sets to 1 if an operand left the fp16 range (result then invalid).
     * w_split_format = MORIG_SPLIT_BF16 selects the bf16 split instead: W_split then holds bf16 halves
     * (hi = bf16(w), lo = bf16(w - hi), same layout), X is split the same way in the kernel and the three products
     * run on v_mfma_f32_32x32x16_bf16 -- bf16 keeps float32's exponent range, so there is no range condition to
     * report (overflow may be NULL); ~16 mantissa bits per operand. Plain stores only (no pool, no split
     * activations): the gradient contractions of the training path (dX = dU W). With MORIG_SPLIT_F16 (= 0) a
     * missing overflow word is MORIG_E_INVALID: the format is never inferred from a NULL pointer. */
    const void* W_split; int32_t* overflow;
    /* split-fp16 ACTIVATIONS (only with W_split): a matrix window whose first column and row stride are
     * multiples of 32 floats and whose every aligned 32-column chunk holds [32 halves hi | 32 halves lo].
     * x_split: X is in that layout (loader becomes a plain copy: the hi/lo split was done once by the
     * producer instead of once per column tile); y_split: write Y in that layout. */
    int32_t x_split, y_split;
    int32_t w_split_format;   /* MORIG_SPLIT_F16 (0) or MORIG_SPLIT_BF16 (1): what W_split holds */
    /* [ABI 3] optional K TAIL (x_split launches only): the last tail_cols columns of X -- columns [K - tail_cols, K); tail_cols % 32 == 0,
     * K % 32 == 0, K > tail_cols -- are read from X_tail[row % tail_rows] instead of X[row]: 128-byte aligned rows of ld_tail floats
     * (ld_tail % 32 == 0, >= tail_cols) in the split-fp16 chunk layout. This is how a replica-invariant block of a concatenated input
     * enters without being copied into every replica's row: the position-branch features [pos_tpl | pos_geo] of a GCUMotion unit under
     * the keyframe loop (models/basic_modules.py:216-217 inside models/rignet.py:85-86; rows r * n + v of the 5 replicas all read tail
     * row v). X then needs only ldx >= K - tail_cols. MORIG_E_UNSUPPORTED where the launch would not take the LDS-DMA store kernel
     * (N % 256 != 0, pool, misaligned Y): the caller copies the block into X instead (morig_copy2d_pad_rep). */
    const float* X_tail; int32_t ld_tail, tail_rows, tail_cols;
} morig_gemm_args;
int morig_gemm(const morig_gemm_args* a, void* stream);

/* --------------------------------------------------------------------------------------------
 * Fused EdgeConv: gather -> 2-layer edge MLP -> segmented max, one pass, no per-edge tensor in HBM.
 * Replaces EdgeConv / EdgeConvMotion .propagate()           models/basic_modules.py:151-155, 190-195
 *
 * For every sorted edge e = (j -> i) and replica r (keyframe), row(i) = r*rep_stride + i
 * (separate strides for the inputs and the output so a replica-invariant branch, e.g. the position
 * branch of EdgeConvMotion under the keyframe loop models/rignet.py:85-86, is gathered from ONE copy):
 *     h1 = s1 * relu(A[row(i)] + B[row(j)]) + t1                  (layer 1 split per vertex:
 *                                                                  W1 [x_i ; x_j - x_i] = (W1a-W1b) x_i + W1b x_j)
 *     z  = s2 * relu(W2 h1 + b2) + t2
 *     out[row(i)] = max over incoming edges of z
 * H = hidden = output width; supported H: 16, 32, 64, 128, 256.
 */
typedef struct morig_edgeconv_args {
    uint32_t struct_size;              /* sizeof(morig_edgeconv_args) of the caller's build (ABI 3) */
    int32_t H;
    int32_t n_nodes, replicas;
    int32_t in_rep_stride;             /* row offset of replica r in A/B: r*in_rep_stride (0 = shared by all replicas) */
    int32_t out_rep_stride;            /* row offset of replica r in out                                              */
    const float* A; int32_t lda;       /* per-vertex target term  (incl. layer-1 bias) */
    const float* B; int32_t ldb;       /* per-vertex source term                         */
    const int32_t* rowptr; const int32_t* src_sorted; const int32_t* dst_sorted;
    int32_t edge_capacity;             /* upper bound of E' used to size the launch      */
    int32_t edge_count;                /* exact E' if the host knows it (0 = unknown); only used for
                                          the algorithmic-FLOP accounting of morig_prof_collect      */
    const float* s1; const float* t1;  /* [Hpad] BatchNorm-1 eval affine                  */
    const float* W2; int32_t ldw;      /* packed [Hpad][ldw], ldw >= roundup(H,32)... see morig_gemm_args.W */
    const float* b2; const float* s2; const float* t2;   /* [Hpad]                        */
    float* out; int32_t ldo;           /* out[row][0..H)                                  */
    const void* W2_split; int32_t* overflow;   /* optional split-fp16 fast path (H >= 32), as in morig_gemm_args */
    int32_t quad_aligned;              /* the CSR was built with MORIG_CSR_PAD4: segments are 4-aligned */
    int32_t out_split;                 /* store `out` in the split-fp16 activation layout (morig_gemm_args.x_split: per 32-column chunk
                                          32 hi halves, then 32 lo halves) so that the unit's MLP (models/basic_modules.py:216-217,
                                          `self.mlp(torch.cat(...))`) reads it through the LDS-DMA GEMM without an fp32 -> fp16 pass.
                                          Needs W2_split + overflow, `out` 128-byte aligned, ldo % 32 == 0 and the launch on one of the
                                          4-aligned-CSR kernels (H = 128 / 256): ask morig_edgeconv_can_split_out first;
                                          MORIG_E_UNSUPPORTED otherwise. *overflow is raised when a result leaves the fp16 range. */
    int32_t exact_arith;               /* [ABI 3] arithmetic of the EXACT path (W2_split == NULL): 0 = v_mfma_f32_32x32x2_f32, 1 = the bf16 x 6
                                          split of MORIG_SPLIT_BF16X6 (H >= 32; morig_edgeconv, morig_edge_hidden) */
    int32_t seg_min4;                  /* [ABI 3] the CSR was built with MORIG_CSR_MIN4 (segments of >= 4 rows, NOT 4-aligned; quad_aligned = 0):
                                          H = 256 split-fp16 launches with out_split then take the mixed-quad form of the W2-stationary
                                          kernel (morig_edgeconv_can_split_out answers for it); everywhere else the flag is ignored -- a
                                          MIN4 CSR is a valid plain CSR for every other kernel (max over a repeated row is the same max). */
    /* [ABI 3, appended in round 6: callers built against the 192-byte struct leave these zero] the boundary passes of TWO launches in one
     * launch each. A launch whose target segments straddle tile boundaries is bracketed by two small passes over exactly those rows of
     * `out` (identity pattern in front of the kernel's atomics; with out_split the conversion of those rows behind it). The two EdgeConvs
     * of a unit -- template graph and geodesic graph, models/basic_modules.py:165-177, 205-219 -- write disjoint column blocks and do not
     * read each other's results, so their passes can share launches: call the FIRST launch with init_with = &second (its pass also
     * prepares the second's rows) and skip_split = 1, then the SECOND with skip_init = 1 and split_with = &first (its conversion pass
     * also converts the first's rows; both must be out_split launches for that half). Both structs must stay valid for both calls, the
     * calls go to the same stream in that order, and nothing may read `out` of the first before the second has been enqueued. */
    const struct morig_edgeconv_args* init_with;
    const struct morig_edgeconv_args* split_with;
    int32_t skip_init, skip_split;
} morig_edgeconv_args;
int morig_edgeconv(const morig_edgeconv_args* a, void* stream);
/* 1 when morig_edgeconv would honour out_split for these arguments (pointers, widths, CSR form and the library's environment
 * switches are looked at; out_split itself is ignored), 0 otherwise. Launches nothing. */
int morig_edgeconv_can_split_out(const morig_edgeconv_args* a);

/* The same EdgeConv for a vertex input of 3 channels (the position branches, models/basic_modules.py:193-195 `nn_pos([pos_i, pos_j - pos_i])`,
 * and motionNet's first unit, whose feature is the 3-channel keyframe flow, models/rignet.py:86): the first Linear is evaluated inside the
 * kernel from the two gathered endpoints -- z = relu(W1a x_i + W1b x_j + b1), W1a = W_a - W_b, W1b = W_b of the reference's
 * Linear(6 -> H) on [x_i, x_j - x_i] -- so no per-vertex [A | B] table is written or gathered (2 x 16 bytes per edge row instead of
 * 2 x 4 H). H == 32 (two 16-wide layers paired, or one 32-wide layer); BatchNorm-1 must be folded into W2 / b2 (packing.fold_hidden_affine).
 *   X [rows][ldx] (ldx >= 4, 16-byte aligned rows; columns 0..2 used), replica r reads rows r * in_rep_stride + vertex;
 *   W1a, W1b [32][4] (column 3 ignored), b1 [32]; everything else as morig_edgeconv_args. */
typedef struct morig_edgeconv_x3_args {
    uint32_t struct_size;              /* sizeof(morig_edgeconv_x3_args) of the caller's build (ABI 3) */
    int32_t H;
    int32_t n_nodes, replicas;
    int32_t in_rep_stride, out_rep_stride;
    const float* X; int32_t ldx;
    const float* W1a; const float* W1b; const float* b1;
    const int32_t* rowptr; const int32_t* src_sorted; const int32_t* dst_sorted;
    int32_t edge_capacity; int32_t edge_count;
    const float* W2; int32_t ldw;
    const float* b2; const float* s2; const float* t2;
    float* out; int32_t ldo;
    const void* W2_split; int32_t* overflow;
    /* [ABI 3, appended in round 6] as morig_edgeconv_args.init_with / skip_init: the first launch of a pair prepares the shared rows of
     * both, the second skips its pass (these launches never store split rows, so there is no conversion pass to share) */
    const struct morig_edgeconv_x3_args* init_with;
    int32_t skip_init, reserved0;
} morig_edgeconv_x3_args;
int morig_edgeconv_x3(const morig_edgeconv_x3_args* a, void* stream);

/* PointConv (3-layer local_nn, models/basic_modules.py:72,82-84; PyG PointConv.message) in two passes:
 *   morig_edge_hidden : Z[e] = s2*relu(W2 (s1*relu(A_i + B_j) + t1) + b2) + t2 for every sorted edge e (< E');
 *                       same arguments as morig_edgeconv, `out` = Z [edge_capacity][ldo], replicas = 1
 *   morig_segmax_gemm : out[i] = max_{e -> i} scale*act(W X[e] + bias) + shift   (layer 3 + aggr='max')
 * With x = None the message is the offset alone; with features it is [x_j ‖ pos_j - pos_i], whose first
 * Linear splits per vertex exactly like EdgeConv's: B_j = W1 [x_j ‖ pos_j] + b1, A_i = -W1p pos_i.        */
int morig_edge_hidden(const morig_edgeconv_args* a, void* stream);
typedef struct morig_segmax_args {
    uint32_t struct_size;                /* sizeof(morig_segmax_args) of the caller's build (ABI 3) */
    int32_t N, K;
    const float* X; int32_t ldx;         /* per-edge rows [edge_capacity][ldx]                       */
    const float* W; int32_t ldw;         /* packed like morig_gemm_args.W (rows padded to 32/64/128/256) */
    const float* bias; const float* scale; const float* shift; int32_t relu;
    const int32_t* rowptr; const int32_t* dst_sorted; int32_t n_nodes;
    int32_t edge_capacity; int32_t edge_count;
    float* out; int32_t ldo;             /* [n_nodes][ldo]                                          */
    const void* W_split; int32_t* overflow;
} morig_segmax_args;
int morig_segmax_gemm(const morig_segmax_args* a, void* stream);

/* The same PointConv in ONE launch pair, straight from the slot table of morig_ball_query (csrc/pointconv_fused.hip;
 * split-fp16 arithmetic only; (H, H3) in {(32, 64), (64, 128)}, max_nbrs == 64): per centre c the kept slots
 * (0 <= source < n_src, source != c) plus the self loop (c, c) -- the edge set morig_csr_from_slots builds, i.e.
 * PointConv's remove_self_loops / add_self_loops (basic_modules.py:82-84) -- without a CSR, per-edge rows in HBM or atomics.
 *   A [n_centres][lda] = -W1p pos_c, B [n_src][ldb] = W1 [x_j | pos_j] + b1 (as for morig_edge_hidden);
 *   W2_split [H][ldw2], b2 [H]: Linear2 with BN1 folded in (packing.fold_hidden_affine);
 *   W3_split [H3][ldw3], b3 [H3]: Linear3 with BN2 folded in the same way; s3 / t3: BN3 (NULL: identity); relu3.
 * *status is set to 1 when a slot names a source >= n_src (out is unspecified then); *overflow as in morig_gemm.
 * MORIG_E_UNSUPPORTED for other widths: the caller falls back to morig_edge_hidden + morig_segmax_gemm. */
typedef struct morig_pointconv_args {
    uint32_t struct_size;                /* sizeof(morig_pointconv_args) of the caller's build (ABI 3) */
    const float* A; int32_t lda;
    const float* B; int32_t ldb;
    const int64_t* slots; int32_t max_nbrs;
    int32_t n_centres, n_src;
    int32_t H, H3;
    const void* W2_split; int32_t ldw2; const float* b2;
    const void* W3_split; int32_t ldw3; const float* b3; const float* s3; const float* t3; int32_t relu3;
    float* out; int32_t ldo;
    int32_t* overflow; int32_t* status;
} morig_pointconv_args;
int morig_pointconv_fused(const morig_pointconv_args* a, void* stream);

/* --------------------------------------------------------------------------------------------
 * Small vertex-parallel operators.
 */
/* strided 2-D copy  dst[r*ldd + c] = src[r*lds + c]  (feature slicing input_flow[:, 3t:3t+3],
 * models/rignet.py:86; torch.cat column placement :65). No alignment requirement. */
int morig_copy2d(const float* src, int32_t lds, float* dst, int32_t ldd, int32_t rows, int32_t cols, void* stream);
/* copy into a zero-padded slot: dst[r][c] = c < cols ? src[r][c] : 0 for c < slot_cols; with split != 0 the slot
 * (slot_cols % 32 == 0, 128-byte aligned) is written in the split-fp16 activation layout; *overflow as in morig_gemm. */
int morig_copy2d_pad(const float* src, int32_t lds, int32_t rows, int32_t cols, float* dst, int32_t ldd,
                     int32_t slot_cols, int32_t split, int32_t* overflow, void* stream);
/* `replicas` such copies in one launch: copy r reads the source window shifted by r * src_col_step columns and writes
 * the destination window shifted by r * dst_row_step rows. Keyframe features (input_flow[:, 3t:3t+3] -> replica t,
 * rignet.py:86): src_col_step = 3; position rows / position-branch columns of the motion replicas: src_col_step = 0. */
int morig_copy2d_pad_rep(const float* src, int32_t lds, int32_t rows, int32_t cols, int32_t src_col_step, float* dst,
                         int32_t ldd, int32_t slot_cols, int32_t replicas, int64_t dst_row_step, int32_t split,
                         int32_t* overflow, void* stream);
/* K tails (morig_gemm_args.X_tail) of up to MORIG_MAX_TAILS concatenations in ONE launch: tail t, row v =
 * [src[v][col_a[t] .. + wa) | src[v][col_b[t] .. + wb) | zeros up to 32] as one split-fp16 chunk at dst + t * tail_stride + v * 32 floats
 * (wa + wb <= 32; dst 128-byte aligned, tail_stride % 32 == 0). Builds [pos_tpl | pos_geo] (models/basic_modules.py:216-217) of every
 * GCUMotion unit of a forward from the buffer the paired position branches wrote (morig_edgeconv_x3). *overflow as in morig_gemm. */
#define MORIG_MAX_TAILS 8
int morig_pack_tails(const float* src, int32_t lds, int32_t rows, const int32_t* col_a /* host [n_tails] */,
                     const int32_t* col_b /* host [n_tails] */, int32_t wa, int32_t wb, int32_t n_tails, float* dst,
                     int64_t tail_stride, int32_t* overflow, void* stream);

/* gather columns: dst[r*ldd + c] = src[r*lds + cols[c]]  (skin_input column selection,
 * models/rignet.py:158-171). cols: int32 [n_cols] on device. */
int morig_gather_cols(const float* src, int32_t lds, const int32_t* cols, int32_t n_cols,
                      float* dst, int32_t ldd, int32_t rows, void* stream);

/* seg_out[r*n_nodes + v] = r*n_graphs + batch[v]  for r < replicas (int64 PyG batch vector in) */
int morig_make_seg(const int64_t* batch, int32_t n_nodes, int32_t n_graphs, int32_t replicas,
                   int32_t* seg_out, void* stream);

/* row-wise L2 normalisation  y = x / max(||x||_2, 1e-12)  (F.normalize, models/rignet.py:87,98;
 * models/corrnet.py:48,60) with a replica-to-token transposition on the way out:
 * input row m = r*rows_per_rep + v  ->  output address y + v*ld_row + r*ld_rep.               */
int morig_rownorm(const float* x, int32_t ldx, int32_t rows_per_rep, int32_t replicas, int32_t cols,
                  float* y, int32_t ld_row, int32_t ld_rep, void* stream);

/* CLS-token temporal attention, exact restructuring of TemporalAttn.forward up to (not incl.) w_o
 * (models/rignet.py:36-45): only token 0 of the output is consumed, so per vertex and head h
 *     score_t = <tok_t, g_h>,  g_h = W_k,h^T (W_q,h cls) / sqrt(d)      (tok_0 = cls, tok_1..T = frames)
 *     y_h     = sum_t softmax(score)_t * tok_t
 * x: [n][T][C] frames; g: [heads][C]; cls: [C]; y: [n][heads*C] (then one GEMM with W_o,h W_v,h). */
int morig_cls_attention(const float* x, int32_t n, int32_t T, int32_t C, int32_t heads,
                        const float* g, const float* cls, float* y, int32_t ldy, void* stream);

/* reductions over keyframes for aggr_method 'mean' / 'max' (models/rignet.py:92-95):
 * x: [n][T][C] -> y[n][C]; mode 0 = mean, 1 = max */
int morig_frame_reduce(const float* x, int32_t n, int32_t T, int32_t C, int32_t mode, float* y, int32_t ldy, void* stream);

/* --------------------------------------------------------------------------------------------
 * CorrNet point branch (models/corrnet.py:50-73, models/basic_modules.py:66-138). Clouds are contiguous
 * row ranges given by int32 offset arrays ptr[n_clouds + 1] (PyG's sorted `batch` vector).
 */
/* The plain AND the 4-aligned (MORIG_CSR_PAD4) CSR of one square graph from ONE pass over the COO: the count pass and the edge
 * ranks inside a segment are shared, both fills happen in one kernel (the rig networks need both per graph and per forward:
 * models/basic_modules.py:188-189 is normalised once instead of once per layer). Same results as two morig_csr_build_bipartite
 * calls up to the order of a target's edges inside its segment. ws: 2 * n_nodes + 1 ints of scratch. */
/* [ABI 3] flags: 0, or MORIG_CSR_MIN4 (the plain CSR's segments are filled up to 4 rows; src_sorted / dst_sorted then need the padded
 * CSR's capacity n_edges + 4 * n_nodes). */
int morig_csr_build_dual(const int64_t* edge_index, int64_t n_edges, int32_t n_nodes, int32_t* rowptr, int32_t* src_sorted,
                         int32_t* dst_sorted, int32_t* rowptr4, int32_t* src_sorted4, int32_t* dst_sorted4, int32_t* ws,
                         int32_t flags, int32_t* status, void* stream);
/* The same normalised CSR as morig_csr_build_bipartite(MORIG_CSR_SKIP_NEGATIVE) for the slot table morig_ball_query
 * writes (target k owns slots [k*max_nbrs, (k+1)*max_nbrs) of row 0 of `coo`, max_nbrs <= 64, unused = -1): counted
 * and filled per target without atomics (replaces torch_cluster.radius -> PointConv's remove/add_self_loops,
 * basic_modules.py:77-84). Capacity of src_sorted / dst_sorted: n_nodes * (max_nbrs + 1); cursor: n_nodes + 1 ints. */
int morig_csr_from_slots(const int64_t* coo, int32_t n_nodes, int32_t max_nbrs, int32_t n_src_nodes,
                         int32_t* rowptr, int32_t* src_sorted, int32_t* dst_sorted, int32_t* cursor, int32_t* status,
                         void* stream);
/* torch_cluster.fps (basic_modules.py:75): per cloud out_ptr[b+1]-out_ptr[b] samples, first = ptr[b] +
 * start[b] (start == NULL: first point), then repeatedly the point farthest from the chosen set
 * (lowest index on ties). idx_out: GLOBAL row indices, clouds concatenated. */
int morig_fps(const float* pos, int32_t ldp, const int32_t* ptr, const int32_t* out_ptr, const int32_t* start,
              int32_t n_clouds, int32_t max_cloud_points, int32_t* idx_out, void* stream);
/* torch_cluster.radius, CUDA-kernel semantics (basic_modules.py:77): for every centre y the first max_nbrs
 * points x of its cloud, in index order, with |x-y|^2 < r^2. coo: int64 [2][n_centres*max_nbrs]
 * (row 0 = x index, row 1 = centre index; unused slots -1), the layout morig_csr_build_bipartite reads. */
int morig_ball_query(const float* x, int32_t ldx, const int32_t* ptr_x, const float* y, int32_t ldy,
                     const int32_t* ptr_y, int32_t n_clouds, int32_t n_centres, float radius,
                     int32_t max_nbrs, int64_t* coo, void* stream);
/* radius_cpu (basic_modules.py:9-29), the ball query of the reference's no-CUDA branch: for every y ALL x with
 * |x-y| <= r (inclusive; no batch vector); a row with more than max_nbrs (<= 64) hits keeps a uniformly random
 * subset of exactly max_nbrs (the reference: torch.multinomial on the 0/1 row; here Algorithm-R reservoir sampling
 * with a counter-based hash of (seed, row, hit number): same distribution over subsets, a different random stream).
 * coo: the slot table of morig_ball_query; counts[ny]: hits per row before the cap. */
int morig_radius_sample(const float* x, int32_t ldx, int32_t nx, const float* y, int32_t ldy, int32_t ny, float radius,
                        int32_t max_nbrs, uint32_t seed, int64_t* coo, int32_t* counts, void* stream);
/* torch_geometric.nn.knn_interpolate (basic_modules.py:134), k <= 3: weights 1/clamp(d^2, 1e-16).
 * idx_ws/wgt_ws: scratch [n_targets][3]. */
int morig_knn_interpolate(const float* feat, int32_t ldf, int32_t C, const float* pos_x, int32_t ldx,
                          const int32_t* ptr_x, const float* pos_y, int32_t ldy, const int32_t* ptr_y,
                          int32_t n_clouds, int32_t n_targets, int32_t max_targets_per_cloud, int32_t k,
                          int32_t* idx_ws, float* wgt_ws, float* out, int32_t ldo, void* stream);
/* The two halves of morig_knn_interpolate, for callers that know the geometry before the features (the CorrNet point branch
 * runs the three searches on its geometry stream, under the convolutions): the k nearest sources of every target and their
 * weights -- slots past k, or past the cloud's point count, carry weight 0 -- then the weighted mean of feature rows. */
int morig_knn_search(const float* pos_x, int32_t ldx, const int32_t* ptr_x, const float* pos_y, int32_t ldy,
                     const int32_t* ptr_y, int32_t n_clouds, int32_t n_targets, int32_t max_targets_per_cloud, int32_t k,
                     int32_t* idx, float* wgt, void* stream);
int morig_knn_apply(const float* feat, int32_t ldf, int32_t C, const int32_t* idx, const float* wgt, int32_t n_targets,
                    float* out, int32_t ldo, void* stream);
/* knn(out_pts, out_vtx, 1, cosine=True) on L2-normalised rows (corrnet.py:64): arg-max dot product
 * within the cloud; C must be 64. */
int morig_cosine_nn(const float* v, int32_t ldv, const int32_t* ptr_v, const float* p, int32_t ldp,
                    const int32_t* ptr_p, int32_t n_clouds, int32_t max_rows_per_cloud, int32_t C,
                    int32_t* nn, float* sim, void* stream);
/* ---- DeformNet glue (/root/reference/models/deformnet.py) ----
 * pred_vismask = sigmoid(logit), then (m - min) / (max - min) per mesh (deformnet.py:42-46). ptr: [n_meshes + 1]
 * vertex offsets. A constant mask gives 0/0 = NaN, as the reference. */
int morig_sigmoid_minmax(const float* x, int32_t ldx, const int32_t* ptr, int32_t n_meshes, float* out, int32_t ldo,
                         void* stream);
/* knn(x, y, k, batch_x, batch_y, cosine=True) on L2-normalised rows (deformnet.py:49 and :92), 1 <= k <= 8, C must
 * be 64: idx[i][t] = global x row of the t-th most similar candidate of the same cloud (lowest index on ties), -1 where
 * there are fewer than k. split = 1 (x == y, ptr_x == ptr_y, vis required): rows with vis < 0.5 query the rows with
 * vis >= 0.5 -- the visible / invisible partition of :57-63 without the compaction; other rows get -1. */
int morig_cosine_knn(const float* y, int32_t ldy, const int32_t* ptr_y, const float* x, int32_t ldx,
                     const int32_t* ptr_x, int32_t n_clouds, int32_t max_rows_per_cloud, int32_t C, int32_t k,
                     const float* vis, int32_t ld_vis, int32_t split, int32_t* idx, void* stream);
/* scatter_add(v * w) / scatter_add(w) over the k neighbours of every vertex (deformnet.py:50-54 and :93-95); l1 rows
 * are [flow(3) | vis] = the feature of GCNDeform (:97).
 *   mode 0: every vertex i, w = <feat_s[j], feat_q[i]> * vis[i], v = pos_s[j] - pos_q[i]; also writes l1[i][3] = vis[i]
 *   mode 1: vertices with vis < 0.5, w = <feat_s[j], feat_q[i]>, v = l1[j][0:3] (the flow of visible vertex j). */
int morig_flow_vote(int32_t mode, const int32_t* idx, int32_t k, int32_t n, const float* feat_q, int32_t ldq,
                    const float* feat_s, int32_t lds, int32_t C, const float* pos_q, int32_t ldpq,
                    const float* pos_s, int32_t ldps, const float* vis, int32_t ld_vis, float* l1, int32_t ld_l1,
                    void* stream);
/* ---- joint extraction after the path (/root/reference/evaluate/eval_rigging.py:80-95). The reference
 * runs these steps in numpy float64, one mesh at a time; all point arrays here are float64 [n][3], contiguous. ----
 * inside_check (utils/mst_utils.py:15-29): keep[i] = 1 when round((p - translate) / scale * dims0) (half to even) lies
 * inside the 88^3 grid (hard-coded there) on a filled voxel. vox88: [88][88][88] bytes; translate: 3 HOST doubles. */
int morig_inside_check(const double* pts, int32_t n, const uint8_t* vox88, const double* translate, double scale,
                       double dims0, uint8_t* keep, void* stream);
/* sklearn.cluster.estimate_bandwidth (eval_rigging.py:89): *bandwidth (device) = mean over points of the distance to
 * the k-th nearest neighbour, the point itself included; the caller passes k = max(int(n * quantile), 1).
 * kth_ws: scratch [n] doubles. */
int morig_knn_bandwidth(const double* pts, int32_t n, int32_t k, double* kth_ws, double* bandwidth, void* stream);
/* meanshift_cluster (utils/cluster_utils.py:14-38): at most max_iter - 1 steps of p_j += 0.3 (weighted mean - p_j) with
 * kernel max(h^2 - d^2, 0) * weights[source] (weights may be NULL); the loop condition "total displacement > 1e-3" is
 * evaluated on the device (state: [max_iter] doubles, step t accumulates its squared displacement in state[t]), so the
 * call enqueues max_iter - 1 launches and never synchronises. bandwidth: device pointer to one double.
 * The result is in buf_a when *result_in_a (HOST int, written before return) is 1, else in buf_b. */
int morig_meanshift(const double* pts, const float* weights, int32_t n, const double* bandwidth, int32_t max_iter,
                    double* buf_a, double* buf_b, double* state, int32_t* result_in_a, void* stream);
/* nms_meanshift (utils/cluster_utils.py:41-66), two steps around the caller's np.argsort(counts)[::-1] (numpy's
 * unstable sort fixes the visiting order among equal counts, so it stays in numpy):
 * counts[j] = #{i : |p_i - p_j| <= bandwidth}; then the greedy pass over `order` writes alive[n]. */
int morig_nms_counts(const double* pts, int32_t n, const double* bandwidth, int32_t* counts, void* stream);
int morig_nms_greedy(const double* pts, const float* attn, int32_t n, const double* bandwidth, const int32_t* order,
                     double thrd_density, float thrd_attn, uint8_t* alive, void* stream);
/* The same four stages over the point sets of SEVERAL meshes at once (the reference loops over models, eval_rigging.py:62-98):
 * sets concatenated, mesh b = rows [ptr[b], ptr[b + 1]) (int32 [n_meshes + 1] on the device), n_all rows in total, max_n = the
 * largest set (sizes the launch grids). Per mesh the arithmetic and its order are those of the one-set entry points.
 *   bandwidth [n_meshes]; k of mesh b = max(int(n_b * quantile), 1) (sklearn estimate_bandwidth);
 *   state [n_meshes][max_iter]; order_local: the visiting order of every mesh as indices LOCAL to the mesh, concatenated. */
int morig_knn_bandwidth_batched(const double* pts, const int32_t* ptr, int32_t n_meshes, int32_t n_all, int32_t max_n,
                                double quantile, double* kth_ws, double* bandwidth, void* stream);
int morig_meanshift_batched(const double* pts, const float* weights, const int32_t* ptr, int32_t n_meshes, int32_t n_all,
                            int32_t max_n, const double* bandwidth, int32_t max_iter, double* buf_a, double* buf_b,
                            double* state, int32_t* result_in_a, void* stream);
int morig_nms_counts_batched(const double* pts, const int32_t* ptr, int32_t n_meshes, int32_t n_all, int32_t max_n,
                             const double* bandwidth, int32_t* counts, void* stream);
int morig_nms_greedy_batched(const double* pts, const float* attn, const int32_t* ptr, int32_t n_meshes, int32_t n_all,
                             const double* bandwidth, const int32_t* order_local, double thrd_density, float thrd_attn,
                             uint8_t* alive, void* stream);
/* Mean-shift over SPATIALLY SORTED point sets: the caller sorts every mesh's points by morig_morton_keys (key = mesh index above a
 * 30-bit Morton code of the position inside [-2, 2)^3), runs morig_meanshift_sorted on the permuted points / weights and scatters the
 * result back. Source tiles whose bounding box lies farther than the bandwidth from a target block's are skipped: the skipped pairs
 * have kernel value 0 exactly, so the result is the unsorted one up to the order of the additions. bbox_ws: 2 * n_meshes * ceil(max_n / 32) * 6
 * doubles (one box per 32 consecutive points, two generations). */
int morig_morton_keys(const double* pts, const int32_t* ptr, int32_t n_meshes, int32_t n_all, int64_t* keys, void* stream);
/* morig_nms_counts_batched over point sets in the same sorted order (e.g. the modes as morig_meanshift_sorted leaves them): boxes
 * of 32 consecutive points farther than the bandwidth from a target block's are skipped -- they hold no neighbour. Counts are
 * integers: identical to the unsorted kernel's. bbox_ws: n_meshes * ceil(max_n / 32) * 6 doubles. */
int morig_nms_counts_sorted(const double* pts, const int32_t* ptr, int32_t n_meshes, int32_t n_all, int32_t max_n,
                            const double* bandwidth, double* bbox_ws, int32_t* counts, void* stream);
int morig_meanshift_sorted(const double* pts, const float* weights, const int32_t* ptr, int32_t n_meshes, int32_t n_all,
                           int32_t max_n, const double* bandwidth, int32_t max_iter, double* buf_a, double* buf_b,
                           double* state, double* bbox_ws, int32_t* result_in_a, void* stream);
/* dst[r] = src[idx[r]] (pos[idx], out_pts[nn]); idx < 0 -> zeros */
int morig_gather_rows(const float* src, int32_t lds, const int32_t* idx, int32_t rows, int32_t cols,
                      float* dst, int32_t ldd, void* stream);

/* --------------------------------------------------------------------------------------------
 * Geodesic-ball graph on the device: data_proc/common_ops.py:214-226 get_geo_edges (radius = 0.06, max_nn = 15), the producer of
 * `geo_edge_index` (datasets/dataset_rig.py:86,122), batched over the meshes of a batch. For every vertex i: the members j != i
 * of its mesh with dist(i, j) <= radius (inclusive) in index order; a row with more than max_nn (<= 64) members keeps a uniformly
 * random subset of exactly max_nn (the reference: np.random.choice(members, max_nn, replace=False) on numpy's global stream;
 * here Algorithm-R reservoir sampling on a counter hash of (seed, row, member number): the same distribution over subsets).
 *   slots   [n_nodes][max_nn] int32, unused = -1        counts [n_nodes] = min(members, max_nn)
 *   members [n_nodes] uncapped member counts (may be NULL)
 *   offsets [n_nodes + 1] exclusive prefix sums of counts (offsets[n_nodes] = number of edges)
 *   scan_ws scratch, >= ceil(n_nodes / 2048) ints
 * morig_geo_ball_graph: Euclidean distance between positions (what SURVEY 8(d)'s synthetic recipe puts in the geodesic's place):
 *   LDS-tiled brute force, d^2 = (dx*dx + dy*dy) + dz*dz in fp32 (no fma) against fp32(radius)^2; meshes = contiguous row
 *   ranges mesh_ptr[n_meshes + 1].
 * morig_geo_ball_graph_dist: the reference's own input, a precomputed n x n float64 distance matrix of ONE mesh
 *   (calc_surface_geodesic, common_ops.py:162-211); the diagonal counts as dist + 10 (common_ops.py:218).
 * morig_geo_ball_fill: rows [i, member] in row order into an int64 COO [2][n_out] (row 0 = i, row 1 = member: the layout of
 *   np.loadtxt(geo_e).T, dataset_rig.py:86); self_loops != 0 appends the n_nodes pairs (i, i) that add_self_loops appends
 *   (dataset_rig.py:122): n_out = offsets[n_nodes] (+ n_nodes). */
int morig_geo_ball_graph(const float* pos, int32_t ldp, const int32_t* mesh_ptr, int32_t n_meshes, int32_t n_nodes,
                         float radius, int32_t max_nn, uint32_t seed, int32_t* slots, int32_t* counts, int32_t* members,
                         int32_t* offsets, int32_t* scan_ws, void* stream);
int morig_geo_ball_graph_dist(const double* dist, int64_t ldd, int32_t n_nodes, double radius, int32_t max_nn, uint32_t seed,
                              int32_t* slots, int32_t* counts, int32_t* members, int32_t* offsets, int32_t* scan_ws,
                              void* stream);
int morig_geo_ball_fill(const int32_t* slots, const int32_t* offsets, int32_t n_nodes, int32_t max_nn, int32_t self_loops,
                        int64_t* coo, int64_t n_out, void* stream);

/* --------------------------------------------------------------------------------------------
 * The path's one collective (the reference has no distributed code, this is the build's own sharding): meshes are
 * sharded whole, one process per GPU, and the per-mesh output rows are all-gathered over RCCL / xGMI once per forward.
 * The product's Python layer issues it through torch.distributed (backend "nccl" IS RCCL on ROCm: morig_amd/dist.py); these
 * exports are the same collective for a host that owns its communicator: `comm` is an ncclComm_t created once per process
 * (morig_rccl_unique_id on rank 0, the 128 bytes handed to every rank by the host's own means, then morig_rccl_comm_init).
 * Equal row counts per rank: one morig_allgather_rows; ragged meshes: morig_allgather_counts first, then a padded gather.
 * Status MORIG_E_HIP + morig_rccl_last_error() = the ncclResult_t. */
#define MORIG_RCCL_UNIQUE_ID_BYTES 128
int morig_rccl_unique_id(void* id_out /* MORIG_RCCL_UNIQUE_ID_BYTES */);
int morig_rccl_comm_init(int32_t n_ranks, int32_t rank, const void* unique_id, void** comm_out);
int morig_rccl_comm_destroy(void* comm);
int morig_rccl_last_error(void);
/* recv [n_ranks * rows][cols] <- every rank's send [rows][cols], rank-major (ncclAllGather on `stream`) */
int morig_allgather_rows(void* comm, const float* send, float* recv, int64_t rows, int32_t cols, void* stream);
int morig_allgather_counts(void* comm, const int64_t* send_one, int64_t* recv_n_ranks, void* stream);

/* --------------------------------------------------------------------------------------------
 * Live per-kernel timing (HIP events on the launch stream) for bench.py's roofline object.
 */
#define MORIG_PROF_KINDS 48
int         morig_prof_enable(int on);                 /* returns previous state */
int         morig_prof_reset(void);
const char* morig_prof_name(int kind);                  /* NULL past the last kind */
/* the ONE kernel symbol every launch of this kind runs, as rocprofv3 --kernel-trace prints it (a prefix: template arguments
 * that do not matter are cut); NULL for kinds that cover several kernels (index / copy / point-cloud helpers) */
const char* morig_prof_symbol(int kind);
/* synchronises the recorded events; per kind: launches, total ms, algorithmic flops, algorithmic bytes */
int         morig_prof_collect(int kind, int64_t* launches, double* total_ms, double* flops, double* bytes);

/* Probe of the matrix pipes' POWER-CAPPED rate (bench.py `roofline.mfma_power_capped_tflops`; SURVEY 8(d) "verify on the box"): enqueues
 * `launches` persistent kernels in which every wave of the chip issues `iters` x 24 v_mfma_f32_32x32x16_f16 from registers
 * (mode 0: random fp16 operands, mode 1: zeros). scratch: >= 1 float of device memory; *flops (optional): dense f16 flops enqueued. */
int morig_ubench_mfma(int mode, int iters, int launches, float* scratch, double* flops, void* stream);

/* ---- train-mode forward support (forward half; training/train_rig.py:136-195) -------------------------------
 * In model.train() every BatchNorm1d normalises with the statistics of the current batch -- over vertices in the dense
 * MLPs, over EDGES inside the per-edge MLPs (models/basic_modules.py:31-36, 153-155, 192-195) -- so the layers run unfused:
 * contraction (morig_gemm / morig_edge_hidden) -> statistics -> affine. */
/* per-column mean and BIASED variance of x[rows][cols] (fp64 accumulation, fixed summation order). rows_dev != NULL: the row
 * count is read on the device (E' = rowptr[n]) and `rows` is only the capacity. workspace: >= ceil(rows/256)*2*cols doubles.
 * count (optional): receives the row count as a float. */
int morig_col_stats(const float* x, int32_t ldx, int32_t rows, const int32_t* rows_dev, int32_t cols, double* workspace,
                    int64_t workspace_doubles, float* mean, float* var, float* count, void* stream);
/* out[r][c] = scale[c] * x[r][c] + shift[c] (BatchNorm once its statistics are known); out == NULL: in place */
int morig_col_affine(float* x, int32_t ldx, int32_t rows, const int32_t* rows_dev, int32_t cols, const float* scale,
                     const float* shift, float* out, int32_t ldo, void* stream);
/* BatchNorm1d in training mode after the column statistics (torch.nn.functional.batch_norm; models/basic_modules.py:31-36): the batch
 * affine s = gamma / sqrt(var + eps), t = beta - mean * s, rstd = 1 / sqrt(var + eps) (may be NULL), and -- when running_mean /
 * running_var are given -- their update with momentum and the UNBIASED variance (count: [1] float on the device, the rows the
 * statistics were taken over), num_batches_tracked += 1 (may be NULL). gamma / beta NULL: 1 / 0. */
int morig_bn_finalize(const float* mean, const float* var, const float* count, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                      float* s, float* t, float* rstd, int32_t n, void* stream);
/* Z[e] = relu(A[dst_e] + B[src_e]) for the E' = rowptr[n_nodes] edges of a CSR: the first edge ReLU
 * (models/basic_modules.py:193-195 after the per-vertex split of Linear1), materialised for its batch statistics. With mean != NULL
 * the same pass also takes them: mean / var / count exactly as morig_col_stats over the E' rows of Z would (fp64, fixed order;
 * workspace as there, sized on edge_capacity); mean == NULL: workspace / var / count are not used. */
int morig_edge_gather_relu(const float* A, int32_t lda, const float* B, int32_t ldb, const int32_t* rowptr, int32_t n_nodes,
                           const int32_t* src_sorted, const int32_t* dst_sorted, int32_t edge_capacity, int32_t H,
                           float* Z, int32_t ldz, double* workspace, int64_t workspace_doubles, float* mean, float* var, float* count,
                           void* stream);
/* out[v][c] = max over rows e in [rowptr[v], rowptr[v+1]) of scale[c] * Z[e][c] + shift[c] (scale == shift == NULL: plain max;
 * empty segment: 0 as torch_scatter): max aggregation behind the last BatchNorm of an edge MLP, and scatter_max over a mesh's
 * vertices (models/rignet.py:63) */
int morig_segmax_affine(const float* Z, int32_t ldz, const int32_t* rowptr, int32_t n_segments, int32_t H, const float* scale,
                        const float* shift, float* out, int32_t ldo, void* stream);

/* --------------------------------------------------------------------------------------------
 * Train-mode BACKWARD operators (backward half; csrc/train_bwd.hip): what torch.autograd computes for the two
 * starred blocks in model.train() -- Seq(Linear, ReLU, BatchNorm1d) over vertices (models/basic_modules.py:31-36) and the
 * per-edge MLP with BatchNorm statistics over edges and max aggregation (:153-155, :179-202). dX = dU W goes through morig_gemm.
 */
/* per column: sum_dz[c] = sum_r dz[r][c] (= dbeta) and sum_dzx[c] = sum_r dz[r][c] * (y[r][c] - mean[c]) * rstd[c] (= dgamma) of
 * a training-mode BatchNorm1d whose INPUT was y; y == NULL: the plain column sum only (= dbias of a Linear). fp64 accumulation,
 * fixed order. workspace: >= ceil(rows/256) * 2 * cols doubles; rows_dev as in morig_col_stats. */
int morig_bn_backward_stats(const float* dz, int32_t ldz, const float* y, int32_t ldy, int32_t rows, const int32_t* rows_dev,
                            int32_t cols, const float* mean, const float* rstd, double* workspace, int64_t workspace_doubles,
                            float* sum_dz, float* sum_dzx, void* stream);
/* du[r][c] = [y > 0] * gamma * rstd * (dz - sum_dz / n - xhat * sum_dzx / n): the BatchNorm and the ReLU in front of it
 * (y = ReLU output = BatchNorm input, n = rows). du may alias dz. sum_du (NULL or [cols]): the column sums of du (fp64 accumulation,
 * fixed order: the bias gradient of the Linear in front) from the same pass; workspace as morig_bn_backward_stats, needed only then. */
int morig_bn_relu_backward(const float* dz, int32_t ldz, const float* y, int32_t ldy, int32_t rows, const int32_t* rows_dev,
                           int32_t cols, const float* mean, const float* rstd, const float* gamma, const float* sum_dz,
                           const float* sum_dzx, float* du, int32_t ldu, double* workspace, int64_t workspace_doubles, float* sum_du,
                           void* stream);
/* morig_segmax_affine that also records which row won: arg[v][c] = row index (first on ties), -1 for an empty segment; zwin
 * (NULL or [n_segments][ldw]) receives Z[arg[v][c]][c], the winner's value in front of the affine (0 for an empty segment): the
 * backward statistics then need no gather. Few long segments (per-mesh pooling) and many short ones (edges) take different kernels. */
int morig_segmax_affine_arg(const float* Z, int32_t ldz, const int32_t* rowptr, int32_t n_segments, int32_t H, const float* scale,
                            const float* shift, float* out, int32_t ldo, int32_t* arg, int32_t ld_arg, float* zwin, int32_t ldw,
                            void* stream);
/* the two BatchNorm sums for the BatchNorm in FRONT of a max aggregation, from the per-segment gradient dout [n_segments][cols]
 * and the arg-max table (the per-row gradient is one-hot per (segment, column) and never materialised); the BatchNorm input of
 * the winning rows comes from zwin [n_segments][ldw] when given (coalesced), otherwise it is gathered from the rows Z */
int morig_segmax_bn_backward_stats(const float* dout, int32_t ldd, const int32_t* arg, int32_t ld_arg, const float* Z, int32_t ldz,
                                   const float* zwin, int32_t ldw, int32_t n_segments, int32_t cols, const float* mean,
                                   const float* rstd, double* workspace, int64_t workspace_doubles, float* sum_dz, float* sum_dzx,
                                   void* stream);
/* du[e][c] for every row e < rowptr[n_segments] (seg_of_row[e] = its segment): the BatchNorm (+ the ReLU in front of it when
 * relu != 0) applied to that one-hot gradient; n = rowptr[n_segments]. Rows rowptr[n_segments] <= e < row_capacity are set to 0.
 * sum_du (NULL or [cols]): the column sums of du over the live rows, fp64 accumulation in a fixed order (the bias gradient of the
 * Linear behind the BatchNorm, from the same pass); workspace: >= ceil(row_capacity / slab) * 2 * cols doubles as in
 * morig_bn_backward_stats, needed only with sum_du. */
int morig_segmax_bn_relu_backward(const float* dout, int32_t ldd, const int32_t* arg, int32_t ld_arg, const float* Z, int32_t ldz,
                                  const int32_t* rowptr, int32_t n_segments, const int32_t* seg_of_row, int32_t row_capacity,
                                  int32_t cols, const float* mean, const float* rstd, const float* gamma, const float* sum_dz,
                                  const float* sum_dzx, int32_t relu, float* du, int32_t ldu, double* workspace,
                                  int64_t workspace_doubles, float* sum_du, void* stream);
/* morig_bn_relu_backward and morig_edge_scatter_backward in one, deterministic: with d[e] = [Y > 0] gamma rstd (dG - sum_dz / n -
 * xhat sum_dzx / n) (n = rowptr[n_nodes]; mean == NULL: d = dG), dA[v] = sum of d over the CSR segment of v and dB[u] = sum of d
 * over the edges out of u, walked through the transposed graph: rowptr_t [n_src_nodes + 1], perm_t[k] = row e of the k-th edge in
 * (source, row) order. d is evaluated where it is summed and never stored; both sums run in a fixed order (no atomics).
 * ZA / ZB (NULL, or [n_nodes][ldza] / [n_src_nodes][ldzb] with src_sorted / dst_sorted): Y is not read, Y[e] = relu(ZA[dst e] + ZB[src e])
 * is rebuilt where it is needed (what morig_edge_gather_relu stored, same expression, same bits): one of the two rows is fixed for a
 * whole segment, the other comes out of the caches, and the edges x H buffer is not read from HBM again. */
int morig_edge_bn_scatter_backward(const float* dG, int32_t ldg, const float* Y, int32_t ldy, const int32_t* rowptr,
                                   const int32_t* rowptr_t, const int32_t* perm_t, int32_t n_nodes, int32_t n_src_nodes, int32_t H,
                                   const float* mean, const float* rstd, const float* gamma, const float* sum_dz, const float* sum_dzx,
                                   float* dA, int32_t lda, float* dB, int32_t ldb, const float* ZA, int32_t ldza, const float* ZB,
                                   int32_t ldzb, const int32_t* src_sorted, const int32_t* dst_sorted, void* stream);
/* backward of Z[e] = A[dst_e] + B[src_e]: dA[v] = sum of dG over the CSR segment of v (fixed order), dB[u] = sum of dG over the
 * edges with source u (float atomics: summation order, hence the last bits, vary run to run). dB ([n_src_nodes][ldb]) is zeroed here. */
int morig_edge_scatter_backward(const float* dG, int32_t ldg, const int32_t* rowptr, const int32_t* src_sorted, int32_t n_nodes,
                                int32_t n_src_nodes, int32_t H, float* dA, int32_t lda, float* dB, int32_t ldb, void* stream);
/* the sums morig_bn_backward_stats would take over dh = dU2 W2 and Z1 (the BatchNorm of the FIRST layer of an edge MLP), from
 * products that exist already: M = dU2^T Z1 [h_out][h_in] (morig_gemm_tn, what dW2 is made of), db2 = column sums of dU2 [h_out], W2
 * [h_out][h_in] (Linear2, out x in): sum_dz[c] = sum_k db2[k] W2[k][c], sum_dzx[c] = rstd[c] sum_k W2[k][c] (M[k][c] - db2[k] mean[c]);
 * fp64, no pass over the edge rows. */
int morig_edge_bn_sums_from_products(const float* M, int32_t ldm, const float* db2, const float* W2, int32_t ldw, const float* mean,
                                     const float* rstd, int32_t h_out, int32_t h_in, float* sum_dz, float* sum_dzx, void* stream);
/* out[N][K] = A^T B over the rows (A [rows][N], B [rows][K], fp32 MFMA): the weight gradient dW = dU^T X. The row range is split
 * over workgroups and the partial products are summed in a fixed order. workspace: morig_gemm_tn_workspace(rows, N, K) floats.
 * Arithmetic: bf16 x 3 split MFMAs by default (both operands split in the kernel: ~16 mantissa bits, float32 exponent range, 2x the
 * exact kernel's rate); the environment variable MORIG_TRAIN_BWD=f32 selects the exact-float32 MFMA kernel. N, K <= 32: plain
 * float32 FMAs (exact products) on a kernel without MFMA tiles, whatever the setting. */
int64_t morig_gemm_tn_workspace(int32_t rows, int32_t N, int32_t K);
int morig_gemm_tn(const float* A, int32_t lda, const float* B, int32_t ldb, int32_t rows, const int32_t* rows_dev, int32_t N, int32_t K,
                  float* workspace, int64_t workspace_floats, float* out, int32_t ldo, void* stream);
/* the same with every row of B centred on b_shift [K] first: C = A^T (B - 1 b_shift^T); b_shift = NULL: plain A^T B. For products that are
 * used as  M - colsum(A) (x) mean  afterwards (the BatchNorm sums of the first edge layer, morig_edge_bn_sums_from_products): with the rows
 * centred on that mean the split contraction itself carries no cancellation */
int morig_gemm_tn_shift(const float* A, int32_t lda, const float* B, int32_t ldb, const float* b_shift, int32_t rows, const int32_t* rows_dev,
                        int32_t N, int32_t K, float* workspace, int64_t workspace_floats, float* out, int32_t ldo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MORIG_HIP_H */
