/*
 * slamhip - C ABI of the MI355X (gfx950) RGB-D frame-pair hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes and returns
 * 0 on success or a negative SLAMHIP_E_* code (never throws, never aborts);
 * slamhip_last_error() gives the message of the calling thread's last failure.
 *
 * Pointer conventions
 *   - "host" functions take host pointers and run on the CPU (pose maths).
 *   - every other function takes DEVICE pointers (hipMalloc'd / torch CUDA
 *     tensors), is asynchronous on `stream` (a hipStream_t passed as void*,
 *     NULL = the null stream) and never allocates or frees device memory.
 *   - images are contiguous row-major: depth [B,H,W] uint16, colour
 *     [B,H,W,3] uint8; neighbour frames [B,k,H,W(,3)]; poses [B,k,16] float64
 *     (row-major 4x4).  K = {fx, fy, cx, cy}; `factor` is the depth divisor
 *     (1 in the reference, 5000 for metric TUM depth).
 *
 * The reference (sanjeevk42/SLAM) is pure Python and has no FFI of its own; the
 * file:line cited at each declaration is the Python function whose arithmetic
 * the entry point replaces.  All coordinate arithmetic is IEEE float64 in the
 * reference's operation order (no contraction), so masks, counts and pixel
 * indices are bit-identical to the reference's numpy results.
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMHIP_OK 0
#define SLAMHIP_E_ARG (-1)      /* bad argument (null pointer, non-positive size, ...) */
#define SLAMHIP_E_HIP (-2)      /* a HIP runtime call or kernel launch failed */
#define SLAMHIP_E_SIZE (-3)     /* size beyond what the entry point supports */

/* number of reduced quantities per (pair, neighbour) */
#define SLAMHIP_NSUMS 4         /* sum|r_photo|, sum r_photo^2, sum|r_geo|, sum r_geo^2 */
#define SLAMHIP_NCOUNTS 4       /* n_valid0, n_inb, n_warp, n_geo */

/* bits of the per-pixel mask byte written by slamhip_warp_materialize */
#define SLAMHIP_M_VALID0 1      /* x,y,z all != 0           (test_import.py:156) */
#define SLAMHIP_M_INB 2         /* 0<=u<=W and 0<=v<=H      (test_import.py:174) */
#define SLAMHIP_M_WARP 4        /* 0<=u<=W-1, 0<=v<=H-1, d>0 (bilinear taps exist) */
#define SLAMHIP_M_GEO 8         /* M_WARP and all four depth-2 taps != 0 */

const char* slamhip_last_error(void);
int slamhip_version(void);

/* How partial sums meet in the encoder entry points (channel statistics, batch-norm backward sums, filter-gradient
 * slices, split-K products; the reference's TensorFlow kernels, slam/network/cnn_model.py:100-113,146-160, give no
 * order either).  Default: float atomics into a cleared output - the last bits differ between runs.  With
 * SLAMHIP_DETERMINISTIC=1 in the environment (read once per process; 1 is returned here) every part is stored to its
 * own slot of a scratch buffer and the slots are added in index order: identical bits on identical inputs.  That
 * scratch buffer is the only device memory the library allocates (grow-only, one per device, outside stream
 * capture; calls of this mode on different streams of one device would race on it). */
int slamhip_deterministic(void);
/* Zero fill by a kernel on `stream`, any address and size (never a memset node in a captured graph). */
int slamhip_zero_fill(void* ptr, size_t bytes, void* stream);
/* on != 0: the calling thread promises that it clears the accumulators of the encoder entry points itself (sums2c,
 * red2c, dw, dbias: e.g. one slamhip_zero_fill over a gradient arena per training step), and the entry points skip
 * their own clearing launches until it is switched off again.  Returns the previous setting. */
int slamhip_caller_clears_accumulators(int on);

/* ---- host: pose / label maths (float64, reference quirks kept) ------------ */

/* slam/preprocess/test_import.py:18-47 - [tx ty tz qx qy qz qw] (float32 rows, as
 * the callers cast them at :199) -> row-major 4x4 float64; q is not normalised. */
int slamhip_quat_to_transformation(const float* gt7, int n, double* T4x4);

/* test_import.py:80-86 - [R^T, -R^T t; 0 1]. */
int slamhip_inverse_trans(const double* T4x4, int n, double* inv4x4);

/* slam/preprocess/import_data.py:12-42 - [x, y, z, rotation vector] of an absolute pose
 * [tx ty tz qx qy qz qw] (float64 in, as the reference casts the text row): Euler angles
 * from the quaternion components in the reference's order, rotation matrix from the
 * angles, w = theta / (2 sin theta) * (R - R^T)v. */
int slamhip_absolute_position(const double* gt7, int n, double* pos6);

/* test_import.py:50-77 (clamp=0) / slam/network/model_input.py:46-76 (clamp=1):
 * SO(3) log + the reference's translation solve; guards give zeros, an exact
 * identity rotation gives NaN. */
int slamhip_trans_to_twist(const double* T4x4, int n, int clamp, double* twist6);

/* test_import.py:89-110 (guard=1) / model_input.py:88-106 (guard=0). */
int slamhip_twist_to_trans(const double* twist6, int n, int guard, double* T4x4);

/* test_import.py:199-207,219 for n independent pairs: rel = inv(T(a)) . T(b) and
 * twist = trans_to_twist(rel).  rel4x4 and twist6 may each be NULL. */
int slamhip_pose_batch(const float* gt7_a, const float* gt7_b, int n, int clamp,
                       double* rel4x4, double* twist6);

/* ---- device: relative pose for ground truth already resident in HBM ------- */

/* test_import.py:199-207 - gt7_a/gt7_b float32 [n,7] (device) -> rel [n,16] float64. */
int slamhip_relative_pose_device(const float* gt7_a, const float* gt7_b, int n,
                                 double* rel4x4, void* stream);

/* ---- device: per-pixel geometry, materialising forms ---------------------- */

/* test_import.py:148-155 dense: xyz [B,H,W,3] (float64 if xyz_f64 else float32,
 * rounded once from the float64 value) and valid0 [B,H,W] (0/1, the compaction
 * rule of :156).  Either output may be NULL. */
int slamhip_backproject(const uint16_t* depth, int B, int H, int W, const double K[4],
                        double factor, void* xyz, int xyz_f64, uint8_t* valid0, void* stream);

/* test_import.py:159-176 dense: transform xyz by rel (one pose per image,
 * [B,16]), project to (u,v,d), evaluate the in-image predicate against
 * dim = [dim_u, dim_v] and count.  xyz [B,H,W,3] must be float64 for
 * bit-exact parity (xyz_f64=1); valid0 may be NULL (then recomputed from xyz);
 * rel4x4 may be NULL (no transform: xyz is projected as it is, :163-168 alone).
 * Outputs (each may be NULL): xyz2 and uvd [B,H,W,3] (float64 if out_f64),
 * iu_iv int32 [B,H,W,2] = floor(u), floor(v) (saturating, NaN -> INT32_MIN),
 * inb uint8 [B,H,W], counts uint32 [B,2] = {n_valid0, n_inb} (zeroed on-stream). */
int slamhip_reproject(const void* xyz, int xyz_f64, const uint8_t* valid0, const double* rel4x4,
                      const double K[4], double dim_u, double dim_v, int B, int H, int W,
                      void* xyz2, void* uvd, int out_f64, int32_t* iu_iv, uint8_t* inb,
                      uint32_t* counts, void* stream);

/* test_import.py:171-176 on an already projected point list: uvd float64 [n,3];
 * mask uint8 [n] (may be NULL) and count uint32 [1] (zeroed on-stream) of the
 * points with 0<=u<=dim_u and 0<=v<=dim_v. */
int slamhip_overlap_count(const double* uvd, long long n, double dim_u, double dim_v, uint8_t* mask,
                          uint32_t* count, void* stream);

/* test_import.py:148-176 as the frame dropping of _get_data calls it (:204-214), in one pass
 * over the anchor's depth plane: depth uint16 [B,H,W] and rel4x4 float64 [B,4,4] ->
 * counts uint32 [B,2] = {points _point_cloud keeps (x, y, z all non-zero), of those the points
 * whose projection into the other frame has 0 <= u <= dim_u and 0 <= v <= dim_v}; the
 * reference's overlap is counts[1] / counts[0].  Bit-identical to slamhip_backproject +
 * slamhip_reproject (counts) without the point cloud in between: 2 bytes read per pixel.
 * counts is cleared on the stream.  At most 65535 pairs per call. */
int slamhip_overlap_fused(const uint16_t* depth, const double* rel4x4, const double K[4], double factor,
                          double dim_u, double dim_v, int B, int H, int W, uint32_t* counts, void* stream);

/* The same in ONE launch: the per-frame counters and a ticket live in `workspace` (16 bytes per frame,
 * slamhip_overlap_fused_workspace_bytes, 16-byte aligned) which the caller zero-fills once (slamhip_zero_fill);
 * every call leaves it zero-filled again - the workgroup that finishes a frame last moves the totals to `counts` and
 * clears the record - so there is no clearing kernel.  Calls that share a workspace must not overlap; after a failed
 * or interrupted call zero-fill it again.  A frame's two counts and its tickets share one 64-bit word: frames of more
 * than some 2^25 pixels do not fit (SLAMHIP_E_SIZE; slamhip_overlap_fused has no such limit). */
size_t slamhip_overlap_fused_workspace_bytes(int B);
int slamhip_overlap_fused_ws(const uint16_t* depth, const double* rel4x4, const double K[4], double factor,
                             double dim_u, double dim_v, int B, int H, int W, uint32_t* counts, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- device: fused frame-pair kernel (the benchmarked path) --------------- */

/* Bytes of workspace slamhip_pair_loss_fused needs for a batch of B anchors with
 * k neighbours each (128 bytes per pair).  No counterpart in the reference. */
size_t slamhip_pair_loss_workspace_bytes(int B, int k);

/* test_import.py:148-176 + bilinear warp + residual reduction, nothing
 * materialised.  depth1 [B,H,W], rgb1 [B,H,W,3]: anchor frames; depth2
 * [B,k,H,W], rgb2 [B,k,H,W,3]: the k neighbour frames of each anchor; rel
 * [B,k,16].  Outputs: sums float64 [B,k,SLAMHIP_NSUMS], counts uint32
 * [B,k,SLAMHIP_NCOUNTS], fully overwritten.
 * `workspace` is device memory of at least slamhip_pair_loss_workspace_bytes(B,k)
 * bytes, 8-byte aligned, that the CALLER ZERO-FILLS ONCE after allocating it;
 * every successful call leaves it zero-filled again, so one launch and no
 * memset runs on the stream.  Calls that share a workspace must not overlap
 * (use one workspace per stream). */
int slamhip_pair_loss_fused(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                            const uint8_t* rgb2, const double* rel4x4, const double K[4],
                            double factor, int B, int k, int H, int W, double* sums,
                            uint32_t* counts, void* workspace, size_t workspace_bytes,
                            void* stream);

/* Zero-fills a workspace of slamhip_pair_loss_fused on `stream`.  A call that was
 * interrupted (a failed launch, a device error, a process that reuses the buffer for
 * another shape after one) may leave accumulators or tickets set, and every later call
 * on that workspace would then return wrong sums without an error: callers run this
 * after any failure before they use the workspace again.  No counterpart in the
 * reference (the workspace is this library's own). */
int slamhip_pair_loss_reset_workspace(void* workspace, size_t workspace_bytes, void* stream);

/* slamhip_pair_loss_fused for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): it has no
 * counterpart in the reference.  Same inputs, outputs and arithmetic, no workspace
 * argument: the accumulator records live in a device array built into the library
 * (SLAMHIP_STATIC_MAX_PAIRS records of 128 bytes on every GPU the library runs on, held
 * until the process ends), which the call clears on `stream` before the launch (with a kernel:
 * a captured call holds kernel nodes only).  It
 * calls no hipMalloc / hipFree.  B * k beyond SLAMHIP_STATIC_MAX_PAIRS is rejected with
 * SLAMHIP_E_SIZE.  All calls on one GPU share that GPU's array: calls that overlap on
 * one GPU (different streams or threads) race and must be serialised by the caller.  The first call on
 * a device looks the array's address up (hipGetSymbolAddress); make that call outside a graph capture
 * (a first call on a capturing stream returns SLAMHIP_E_ARG).  `stream` must belong to the CURRENT device
 * (every GPU holds its own copy of the array; SLAMHIP_E_ARG otherwise). */
#define SLAMHIP_STATIC_MAX_PAIRS 8192
int slamhip_pair_loss_fused_static(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                                   const uint8_t* rgb2, const double* rel4x4, const double K[4],
                                   double factor, int B, int k, int H, int W, double* sums,
                                   uint32_t* counts, void* stream);

/* Same arithmetic as slamhip_pair_loss_fused, per-pixel results written out
 * (each output may be NULL): warped_rgb float32 [B,k,H,W,3], warped_depth
 * float32 [B,k,H,W], r_photo float32 [B,k,H,W,3], r_geo float32 [B,k,H,W],
 * mask uint8 [B,k,H,W] (SLAMHIP_M_* bits), iu_iv int32 [B,k,H,W,2], uvd
 * float64 [B,k,H,W,3]. */
int slamhip_warp_materialize(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                             const uint8_t* rgb2, const double* rel4x4, const double K[4],
                             double factor, int B, int k, int H, int W, float* warped_rgb,
                             float* warped_depth, float* r_photo, float* r_geo, uint8_t* mask,
                             int32_t* iu_iv, double* uvd, void* stream);

/* ---- device: pose-regression encoder (bfloat16 operands, float32 accumulate) --
 * Activations are NHWC bfloat16 (passed as void*); master parameters float32 in
 * the reference's layouts: conv filters HWIO [3,3,Cin,Cout] (cnn_model.py:90-95),
 * fully connected [in,out] (cnn_model.py:120-121). */

/* slam/network/cnn_model.py:100-101 (tf.nn.conv2d, 3x3, stride 1, SAME) as an
 * implicit GEMM on the matrix cores.  act [B,H,W,C] bf16; w_nk [N][9*C] bf16,
 * element (n, tap*C + c) = filter[tap][c][n] (slamhip_conv_weight_layouts makes
 * it); outputs [B,H,W,N] as bf16 and/or float32 (either may be NULL).
 * flip=1 visits the taps in reverse: with w_nk = the [Cin][9*Cout] layout and
 * act = dL/dout this is the gradient with respect to the input.  C % 64 == 0. */
int slamhip_conv3x3_igemm_bf16(const void* act, const void* w_nk, void* out_bf16, float* out_f32,
                               int B, int H, int W, int C, int N, int flip, void* stream);

/* The same forward convolution with the per-channel statistics of the next step
 * (cnn_model.py:104, tf.nn.moments) taken in its epilogue: sums2c[n] = sum z,
 * sums2c[N+n] = sum z^2 over all pixels of the bfloat16 output (float32 [2N],
 * overwritten), saving slamhip_channel_stats' pass over z.  `workspace`: device
 * memory of slamhip_conv_stats_workspace_bytes(B*H*W, N) bytes, contents
 * irrelevant before and after.  No counterpart in the reference (a fusion). */
size_t slamhip_conv_stats_workspace_bytes(long long pixels, int N);
int slamhip_conv3x3_igemm_bf16_stats(const void* act, const void* w_nk, void* out_bf16, int B, int H,
                                     int W, int C, int N, float* sums2c, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* cnn_model.py:74,123 (tf.matmul) - out[m][n] = sum_k a[m][k] * b[n][k]; a [M][K],
 * b [N][K] bf16, K % 64 == 0. */
int slamhip_gemm_nt_bf16(const void* a_mk, const void* b_nk, void* out_bf16, float* out_f32, int M,
                         int N, int K, void* stream);

/* Gradient of the filter (taps=9: dw [9][Cin][Cout], x [B,H,W,Cin], dy
 * [B,H,W,Cout]) or of a fully connected weight (taps=1, H=W=1: dw = x^T dy,
 * [Cin][Cout]); dw float32, overwritten.  What tf.gradients derives for
 * cnn_model.py:100,123.  Cin % 64 == 0 and Cout % 64 == 0. */
int slamhip_wgrad_bf16(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin,
                       int Cout, int taps, void* stream);

/* First layer of the network (cnn_model.py:38, Cin = 4): direct convolution and
 * filter gradient from the float32 HWIO filter; Cout must divide 256 for the
 * gradient. */
int slamhip_conv3x3_small_fwd(const void* x, const float* w_hwio, void* out_bf16, int B, int H,
                              int W, int Cin, int Cout, void* stream);
int slamhip_conv3x3_small_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W,
                                int Cin, int Cout, void* stream);
/* ... and with the output's channel statistics, as slamhip_conv3x3_igemm_bf16_stats
 * (Cout % 64 == 0). */
int slamhip_conv3x3_small_fwd_stats(const void* x, const float* w_hwio, void* out_bf16, int B, int H,
                                    int W, int Cin, int Cout, float* sums2c, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* cnn_model.py:104-107 (tf.nn.moments over axes [0,1,2] +
 * tf.nn.batch_normalization with the batch statistics, in training and in
 * inference alike; biased variance).
 * channel_stats: sums2c[c] = sum z, sums2c[C+c] = sum z^2 over the P rows of z
 * [P][C] bf16 (float32 [2C], overwritten).  bn_relu_fwd: y = relu(gamma *
 * (z - mean) / sqrt(var + eps) + beta) (cnn_model.py:107,113).  bn_relu_bwd: dz
 * from dy through the ReLU (gate recomputed from z, y is not read) and the
 * normalisation, and red2c[c] = dbeta,
 * red2c[C+c] = dgamma (float32 [2C], overwritten).  C % 64 == 0. */
int slamhip_channel_stats(const void* z, float* sums2c, long long P, int C, void* stream);
int slamhip_bn_relu_fwd(const void* z, const float* sums2c, const float* gamma, const float* beta,
                        void* y, long long P, int C, float eps, void* stream);
int slamhip_bn_relu_bwd(const void* dy, const void* z, const float* sums2c, const float* gamma,
                        const float* beta, float* red2c, void* dz, long long P, int C, float eps,
                        void* stream);

/* Backward of the first layer (4 input channels, 64 filters, batch-norm, no pool;
 * cnn_model.py:37,104-113) in two passes: red2c as slamhip_bn_relu_bwd, then the filter
 * gradient dw [3,3,4,64] of slamhip_conv3x3_small_wgrad with dz computed on the way from dy
 * (gradient at the layer's output) and z, so that dz is neither written nor read.  Same
 * arithmetic and the same bfloat16 rounding of dz as the two separate calls. */
int slamhip_bn_relu_bwd_first_layer(const void* dy, const void* z, const float* sums2c, const float* gamma,
                                    const float* beta, float* red2c, const void* x, float* dw, int B, int H,
                                    int W, float eps, void* stream);

/* The first layer (4 input channels, 64 filters, batch-norm, no pool; cnn_model.py:37,100-113) without a
 * stored z.  Forward: one pass for the channel statistics of z (sums2c, as slamhip_conv3x3_small_fwd_stats),
 * a second pass that recomputes z and writes y = relu(batch-norm(z)) only: the values of
 * slamhip_conv3x3_small_fwd_stats + slamhip_bn_relu_fwd bit for bit, for one tensor write instead of two
 * writes and a read.  workspace: slamhip_conv_stats_workspace_bytes(B*H*W, 64) bytes. */
int slamhip_conv3x3_c4_bn_relu_fwd(const void* x, const float* w_hwio, const float* gamma, const float* beta,
                                   void* y_bf16, float* sums2c, void* workspace, size_t workspace_bytes, int B,
                                   int H, int W, float eps, void* stream);
/* Backward of that layer, as slamhip_bn_relu_bwd_first_layer, with z recomputed from x and the filter inside
 * both passes instead of read (same values: the forward's operand order).  workspace: 512 bytes per 64 * ceil(
 * ceil(B*H*W / 2048) / 64) pixels, at most 1 MB. */
int slamhip_bn_relu_bwd_first_layer_recompute(const void* dy, const void* x, const float* w_hwio,
                                              const float* sums2c, const float* gamma, const float* beta,
                                              float* red2c, float* dw, void* workspace, size_t workspace_bytes,
                                              int B, int H, int W, float eps, void* stream);

/* The two calls above fused with the 2x2 stride-2 max-pool that follows the last
 * layer of every block (cnn_model.py:40,45,50,56,62).  Forward: z [B,H,W,C] -> y (same
 * shape; NULL = not written, nothing downstream of a pooled layer reads it) and
 * pooled [B,H/2,W/2,C] in one pass.  Backward: dpooled [B,H/2,W/2,C] is
 * the gradient at the pooled tensor; it goes to each window's first maximum of y
 * (recomputed from z), through the ReLU gate and the batch-norm backward into dz
 * [B,H,W,C]; red2c as above.  H and W even. */
int slamhip_bn_relu_pool_fwd(const void* z, const float* sums2c, const float* gamma, const float* beta,
                             void* y, void* pooled, int B, int H, int W, int C, float eps, void* stream);
int slamhip_bn_relu_pool_bwd(const void* dpooled, const void* z, const float* sums2c, const float* gamma,
                             const float* beta, float* red2c, void* dz, int B, int H, int W, int C,
                             float eps, void* stream);

/* The two backward calls with their passes apart, for normalisation statistics kept over several
 * data-parallel ranks (the reference is single-device: cnn_model.py:100-113 normalises over the
 * one batch it sees; one batch spread over N ranks gives the same numbers only if the sums do).
 * phases bit 0: the reduction pass (writes red2c, dz may be NULL); bit 1: the pass that writes dz
 * from red2c.  Between the two the caller averages red2c over the ranks, and sums2c after the
 * forward's statistics; every rank passes its own row count.  phases = 3 is the plain call. */
int slamhip_bn_relu_bwd_phases(const void* dy, const void* z, const float* sums2c, const float* gamma,
                               const float* beta, float* red2c, void* dz, long long P, int C, float eps,
                               int phases, void* stream);
int slamhip_bn_relu_pool_bwd_phases(const void* dpooled, const void* z, const float* sums2c,
                                    const float* gamma, const float* beta, float* red2c, void* dz, int B,
                                    int H, int W, int C, float eps, int phases, void* stream);

/* cnn_model.py:123,74 / cnn_model_noBN.py (bias_add + relu) - y = act(z + bias) on a float32
 * [M][N] product (relu=0: the linear output layer); backward: dz = dy * (y>0),
 * dbias[n] = sum_m dz. */
int slamhip_bias_act_fwd(const float* z, const float* bias, void* y_bf16, float* y_f32, int M,
                         int N, int relu, void* stream);
int slamhip_bias_act_bwd(const float* dy, const void* y_bf16, void* dz_bf16, float* dbias, int M,
                         int N, int relu, void* stream);

/* cnn_model.py:40,45,50,56,62 (tf.nn.max_pool 2x2, stride 2, VALID) and its gradient
 * (to the first maximum of each window in row-major order).  C % 8 == 0. */
int slamhip_maxpool2_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream);
int slamhip_maxpool2_bwd(const void* x, const void* y, const void* dy, void* dx, int B, int H,
                         int W, int C, void* stream);

/* Parameter casts / re-layouts into the operand forms above. */
int slamhip_cast_f32_bf16(const float* src, void* dst, long long n, void* stream);
int slamhip_transpose_cast_bf16(const float* src_rc, void* dst_cr, int R, int C, void* stream);
/* The same re-layout from a bf16 source ([R][C] -> [C][R]). */
int slamhip_transpose_bf16(const void* src_rc, void* dst_cr, int R, int C, void* stream);
int slamhip_conv_weight_layouts(const float* w_hwio, void* w_fwd, void* w_dgrad, int Cin, int Cout,
                                void* stream);
/* The same for n_layers <= 16 filters that lie in one flat float32 buffer (layer l at flat_params + offsets[l]), in
 * one launch: what a training step needs after every parameter update.  offsets, cin, cout, w_fwd, w_dgrad are HOST
 * arrays of n_layers entries; a null w_fwd[l] or w_dgrad[l] skips that layout. */
int slamhip_conv_weight_layouts_multi(const float* flat_params, int n_layers, const long long* offsets,
                                      const int* cin, const int* cout, void* const* w_fwd, void* const* w_dgrad,
                                      void* stream);

/* cnn_model.py:139-140 - loss[0] = tf.nn.l2_loss((out - gt) . loss_weight) =
 * 0.5 * sum of squares; d_out (may be NULL) = its gradient with respect to out.
 * out, gt [B][D] float32, loss_weight [D][D] row-major, D <= 8. */
int slamhip_l2loss(const float* out, const float* gt, const float* loss_weight, float* loss,
                   float* d_out, int B, int D, void* stream);

/* slam/network/lstm_model.py:52-55 and the general form of the loss above:
 * loss[0] = scale * sum(((out - gt) . loss_weight)^2), d_out its gradient
 * (scale = 1/B gives the LSTM trainer's loss, 0.5 gives slamhip_l2loss). */
int slamhip_weighted_sq_loss(const float* out, const float* gt, const float* loss_weight, float scale,
                             float* loss, float* d_out, int B, int D, void* stream);

/* slam/main/vgg16_posenet.py:16-25 - rows [x y z qw qx qy qz]: loss[0] =
 * sum((x^ - x)^2) + beta * sum((q^ - q)^2) (beta = 1100 in the reference);
 * d_out (may be NULL) [B][7]. */
int slamhip_posenet_loss(const float* out7, const float* gt7, float beta, float* loss, float* d_out,
                         int B, void* stream);

/* slam/network/cnn_model_noBN.py:104-111 - the convolution layer without
 * batch-norm: y = relu(z + bias) on NHWC bfloat16 z [P][C] (C % 8 == 0), and its
 * backward: dz = dy * (y > 0), dbias[c] = sum dz (float32 [C], overwritten;
 * C % 64 == 0). */
int slamhip_bias_relu_fwd(const void* z, const float* bias, void* y, long long P, int C, void* stream);
int slamhip_bias_relu_bwd(const void* dy, const void* y, void* dz, float* dbias, long long P, int C,
                          void* stream);

/* cnn_model.py:162-166 - tf.train.AdamOptimizer update of one parameter tensor:
 * g = grad * grad_scale; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr * sqrt(1-b2^step)/(1-b1^step) * m / (sqrt(v) + eps).  step >= 1. */
int slamhip_adam_step(float* param, const float* grad, float* m, float* v, long long n, float lr,
                      float beta1, float beta2, float eps, int step, float grad_scale,
                      void* stream);
/* The same update (cnn_model.py:146-160); in the same pass the first shadow_n updated
 * parameters are also written as bf16 to shadow_bf16 (the operand copy of the fully
 * connected weights, which lie first in the flat parameter buffer). */
int slamhip_adam_step_shadow(float* param, const float* grad, float* m, float* v, long long n, float lr,
                             float beta1, float beta2, float eps, int step, float grad_scale,
                             void* shadow_bf16, long long shadow_n, void* stream);

/* The same update for a training step that is recorded once in a HIP graph and replayed (the kernel
 * arguments of a replayed launch are frozen): lr_t_dev points to one float in device memory holding
 * lr * sqrt(1 - beta2^step) / (1 - beta1^step) of the current step, which the caller rewrites before
 * every replay.  Same arithmetic as slamhip_adam_step_shadow (cnn_model.py:146-160). */
int slamhip_adam_step_dev(float* param, const float* grad, float* m, float* v, long long n,
                          const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale,
                          void* shadow_bf16, long long shadow_n, void* stream);

/* ---- device: input preparation of the RGB-D trainers --------------------------
 * slam/network/model_input.py:427-450 + :360-362,517-522 for a batch of raw frames
 * resident in HBM: rgb uint8 [B,Hs,Ws,3], depth uint16 [B,Hs,Ws] ->
 * depth' = 255 * depth / max(depth) (integer division; max per frame),
 * 4-channel concat, bilinear resize to Ho x Wo (sample centres (i + 0.5) * Hs / Ho
 * - 0.5, clamped), float32, minus the per-channel mean of the whole batch.
 * out_f32 [B,Ho,Wo,4]; out_bf16 (may be NULL) the same rounded to bfloat16 (the
 * encoder's input format); mean4 (may be NULL) float64 [4] the subtracted mean.
 * workspace: slamhip_prep_rgbd_workspace_bytes(B) bytes, contents irrelevant. */
size_t slamhip_prep_rgbd_workspace_bytes(int B);
int slamhip_prep_rgbd(const uint8_t* rgb, const uint16_t* depth, int B, int Hs, int Ws, int Ho, int Wo,
                      float* out_f32, void* out_bf16, double* mean4, void* workspace,
                      size_t workspace_bytes, void* stream);

/* slam/network/model_input.py:185-217 - the same preparation as the reference's
 * in-graph reader does it: depth decoded to 8 bits (tf.image.decode_png's default
 * dtype keeps the upper byte of a 16-bit sample), both images resized with
 * TensorFlow 0.x's bilinear rule (source coordinate = index * in / out, float32),
 * then depth * 255 / max(resized depth) per frame, 4-channel concat, no mean
 * subtraction.  Same workspace as slamhip_prep_rgbd. */
int slamhip_prep_rgbd_tf(const uint8_t* rgb, const uint16_t* depth, int B, int Hs, int Ws, int Ho, int Wo,
                         float* out_f32, void* workspace, size_t workspace_bytes, void* stream);

/* cnn_model.py:167-170 - the other two optimizers the configuration may name:
 * tf.train.GradientDescentOptimizer (p -= lr g) and tf.train.RMSPropOptimizer
 * with momentum 0 (ms = decay ms + (1-decay) g^2; p -= lr g / sqrt(ms + eps);
 * TensorFlow's defaults are decay 0.9, eps 1e-10 and ms starting at one). */
int slamhip_sgd_step(float* param, const float* grad, long long n, float lr, float grad_scale,
                     void* stream);
int slamhip_rmsprop_step(float* param, const float* grad, float* ms, long long n, float lr,
                         float decay, float eps, float grad_scale, void* stream);

/* ---- device: recurrent head of the CNN-LSTM trainer ----------------------------
 * slam/network/lstm_model.py:25-49 - L stacked LSTM cells of n units applied for one
 * time step to x [B][n], then out = m_top . fc_w + fc_b.  The cell is TensorFlow
 * 0.x's LSTMCell without peepholes or projection (an un-vendored dependency of the
 * reference): (i, j, f, o) = split([x, m_prev] . W + b); c = sigmoid(f + forget_bias)
 * c_prev + sigmoid(i) tanh(j); m = sigmoid(o) tanh(c).  float32, row-major:
 * W [L][2n][4n], bias [L][4n], fc_w [n][D], fc_b [D]; state_in / state_out
 * [B][L][2n] with [c, m] per layer (the MultiRNNCell state the reference feeds as
 * zeros, main/lstm_rgbd.py:60); gates [B][L][4n] receives the activated gates for the
 * backward pass.  n <= 64, D <= 64. */
int slamhip_lstm_head_fwd(const float* x, const float* state_in, const float* W, const float* bias,
                          const float* fc_w, const float* fc_b, float forget_bias, int B, int n, int L,
                          int D, float* gates, float* state_out, float* out, void* stream);
/* Gradients of a loss with d loss / d out = d_out [B][D]: d_x [B][n] (flows into the
 * encoder), dW, dbias, dfc_w, dfc_b (shapes of the parameters, overwritten; batch
 * sums in row order).  d_pre [B][L][4n] is scratch that receives the gradients at the
 * gate pre-activations.  No gradient is produced for state_in (a fed constant). */
int slamhip_lstm_head_bwd(const float* d_out, const float* x, const float* state_in, const float* W,
                          const float* fc_w, const float* gates, const float* state_out, int B, int n,
                          int L, int D, float* d_pre, float* d_x, float* dW, float* dbias, float* dfc_w,
                          float* dfc_b, void* stream);

/* ---- device: layer kernels of the GoogLeNet / PoseNet encoder --------------------
 * slam/network/google_net.py:14-215 builds on kaffe.tensorflow.network.Network (an
 * un-vendored dependency): conv = tf.nn.conv2d SAME + bias + ReLU, max_pool / avg_pool
 * SAME unless stated, lrn = tf.nn.local_response_normalization, concat on channels,
 * fc = x.W + b.  SAME follows TensorFlow: out = ceil(in / stride), total padding
 * max((out-1)*stride + k - in, 0), the smaller half in front.  Activations are NHWC
 * bfloat16; the GEMMs are slamhip_gemm_nt_bf16 / slamhip_wgrad_bf16 on the im2col matrix.
 * Every *_add entry ADDS into its destination (a tensor that feeds several branches
 * collects their gradients; the caller zeroes the buffer once per step). */
int slamhip_conv_out_shape(int H, int W, int KH, int KW, int stride, int same, int* Ho, int* Wo);
/* col [B*Ho*Wo][Kp] bfloat16, K = KH*KW*C in (ky, kx, c) order, Kp = K rounded up to 64
 * (zero filled); SAME padding; C % 8 == 0. */
int slamhip_im2col_bf16(const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int stride,
                        void* stream);
/* dx [B,H,W,C] += the transpose of the gather above applied to dcol [B*Ho*Wo][Kp]. */
int slamhip_col2im_add_bf16(const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW,
                            int stride, void* stream);
/* Epilogue of a convolution / fc layer into a channel slice: y[m*ld + off + n] =
 * act(z[m*Np + n] + bias[n]) for n < N (z float32 [M][Np]); y_bf16 and/or y_f32.  And its
 * backward: dz [M][Np] bfloat16 = dy[m*ld + off + n] (dy_bf16 or dy_f32) gated by y > 0
 * when relu, zero for n >= N; dbias [Np] float32 = column sums (overwritten). */
int slamhip_bias_act_slice_fwd(const float* z, const float* bias, void* y_bf16, float* y_f32, long long M,
                               int N, int Np, int ld, int off, int relu, void* stream);
int slamhip_bias_act_slice_bwd(const void* dy_bf16, const float* dy_f32, const void* y_bf16, void* dz,
                               float* dbias, long long M, int N, int Np, int ld, int off, int relu,
                               void* stream);
/* K x K pooling with the given stride, SAME (same != 0) or VALID; is_max selects max or
 * average (average over the cells inside the image).  Backward adds into dx; a max
 * window's gradient goes to its first maximum in (ky, kx) order. */
int slamhip_pool_fwd(const void* x, void* y, int B, int H, int W, int C, int K, int stride, int same,
                     int is_max, void* stream);
int slamhip_pool_bwd_add(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int K,
                         int stride, int same, int is_max, void* stream);
/* y = x / (bias + alpha * sum_{|j-c| <= radius} x_j^2)^beta over the channel axis of
 * x [P][C] bfloat16 (google_net.py:18,21: radius 2, alpha 2e-5, beta 0.75, bias 1). */
int slamhip_lrn_fwd(const void* x, void* y, long long P, int C, int radius, float bias, float alpha,
                    float beta, void* stream);
int slamhip_lrn_bwd_add(const void* x, const void* dy, void* dx, long long P, int C, int radius, float bias,
                        float alpha, float beta, void* stream);
/* float32 [P][C] -> bfloat16 [P][Cp] with zero channels C..Cp-1 (the 3- or 4-channel
 * network input widened to the 8 channels im2col works in). */
int slamhip_pad_cast_bf16(const float* src, void* dst, long long P, int C, int Cp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLAMHIP_H */
