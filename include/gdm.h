/*
 * gdm.h -- C ABI of libgdm_hip.so: the MI355X (gfx950) kernels under the GAN training hot path of
 * marja-w/gan-des-midi-music-gen (GAN_DES/SIMNN.py, MMGAN_MIDI_DES/network_tests.py).
 *
 * The reference has no native/FFI boundary of its own: everything below its Python classes is
 * PyTorch/ATen.  Each entry point here therefore names the ATen dispatch (and the reference call site) it replaces.
 *
 * Contract (all entry points):
 *   - plain C: device pointers + sizes, no torch types; `stream` is a hipStream_t passed as void*.
 *   - caller owns all memory; nothing is allocated, freed or synchronised here; work is only enqueued on `stream`.
 *   - returns 0 on success, a negative GDM_E* code otherwise; gdm_last_error() gives the thread-local message.
 *   - re-entrant across host threads (autograd runs backward on its own thread); no global mutable state.
 *   - `dtype` arguments: GDM_F32 (exact-fp32 mode, v_mfma_f32_16x16x4_f32) or GDM_BF16 (bf16 storage/MFMA operands,
 *     v_mfma_f32_16x16x32_bf16, fp32 accumulation).  Parameters, gradients of parameters, optimizer state, batch-norm
 *     statistics and losses are always fp32.
 */
#ifndef GDM_H
#define GDM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GDM_F32 = 0, GDM_BF16 = 1 };
enum { GDM_ACT_NONE = 0, GDM_ACT_RELU = 1, GDM_ACT_LEAKY = 2, GDM_ACT_SIGMOID = 3 };
enum { GDM_OK = 0, GDM_EINVAL = -1, GDM_ELAUNCH = -2, GDM_EWORKSPACE = -3 };

/* ---- probes -------------------------------------------------------------------------------------------------- */
const char* gdm_last_error(void);
int gdm_version(void);          /* 1000*major + minor */
const char* gdm_arch(void);     /* "gfx950" */
int gdm_build_flavor(void);     /* 0 = shipped flags; 1 = built with experiment switches (GDM_HIPCC_FLAGS): not benchable */

/* ---- dense layers (aten::addmm / aten::mm: nn.Linear fwd, dX, dW; ConvTranspose2d as GEMM) ---------------------
 * C[m,n] = act( sum_k A[m,k]*B[k,n] + bias_n[n] + bias_m[m] ), arbitrary element strides (transposes are free).
 * Replaces F.linear at SIMNN.py:140-141, network_tests.py:77,112,139,160 and their autograd mm's.
 * compute_dtype selects the MFMA; operands of the other type are converted while staged into LDS.
 * split_k > 1 partitions K over blockIdx.z into fp32 slabs in `workspace` (>= split_k*M*N*4 bytes) that a second
 * kernel sums in fixed order (deterministic) before bias/activation.                                              */
int gdm_gemm(const void* A, int a_dtype, int64_t sam, int64_t sak,
             const void* B, int b_dtype, int64_t sbk, int64_t sbn,
             void* C, int c_dtype, int64_t scm, int64_t scn,
             int M, int N, int K,
             const float* bias_n, const float* bias_m, int act, float slope,
             int compute_dtype, int split_k, void* workspace, size_t workspace_bytes, void* stream);

/* ---- loss (aten::binary_cross_entropy_with_logits, mean) --------------------------------------------------------
 * x: n fp32 values fed to BCEWithLogitsLoss (for model 1 these are already sigmoid outputs: SIMNN.py:141 + 289),
 * target: one label value for the whole batch (0.9/0.1/1.0 at SIMNN.py:284,308,326; 1/0 at network_tests.py:286-287).
 * Writes loss[0] (mean) and, if dx != NULL, dx[i] = grad_scale * (sigmoid(x[i]) - target) / n.
 * If fuse_sigmoid_backward the chain through model 1's final sigmoid is fused: x[i] is sigmoid(z[i]) and dx[i] is
 * d loss / d z[i].  accumulate_loss adds the mean to loss[0] instead of overwriting it (disc_loss = fake + real,
 * SIMNN.py:314).  n <= 65536 (one workgroup, fixed-order reduction => deterministic).                              */
int gdm_bce_with_logits(const float* x, float target, int n, float grad_scale, float* loss, float* dx,
                        int fuse_sigmoid_backward, int accumulate_loss, void* stream);

/* ---- optimizer (torch.optim.Adam single-tensor step, SIMNN.py:258-259,316; network_tests.py:253-254,308) -------
 * One flat fp32 range: p, g, m (exp_avg), v (exp_avg_sq) of n elements; `step` is the 1-based step count.
 * bias corrections are computed on the host in double like torch does.  grad_scale multiplies g on the fly
 * (1/world_size after a summed gradient all-reduce; 1 otherwise).                                                  */
int gdm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float beta1,
                  float beta2, float eps, float grad_scale, void* stream);

/* the same update with the step counter and hyper-parameters resident on the device, so that a captured hipGraph can
 * be replayed: hyper = 8 floats {step (int32 bits), lr, beta1, beta2, eps, grad_scale, -, -}; every call increments
 * the step and recomputes the bias corrections on the device (in double), then updates the flat range.             */
int gdm_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* hyper, void* stream);
/* The same update for ONE parameter viewed as (N, C, P) whose gradient g_pc is laid out (N, P, C) -- model 1's
 * fc1.weight (128, 32, H2*W2): its weight-gradient GEMM produces the channels-last flatten order -- and whose updated
 * value is also written to shadow_pc (N, P, C) in shadow_dtype, the operand copy the forward / dX GEMMs read.  p, m, v
 * keep the reference's (N, C, P) order.  Replaces two gdm_permute_pc passes around the optimizer.  advance_step: 1 =
 * increment the step and recompute the bias corrections first (like gdm_adam_step_dev); 0 = use the record as it is
 * (a second range of the same optimizer step).                                                                      */
int gdm_adam_step_dev_pc(float* p, const float* g_pc, float* m, float* v, int N, int C, int P, void* shadow_pc,
                         int shadow_dtype, float* hyper, int advance_step, void* stream);

/* ---- batch norm, training mode, rows x channels matrices (aten::native_batch_norm + activation) ----------------
 * y: (rows, channels) fp32 pre-norm values (row-major).  Computes per-channel batch mean / biased variance with a
 * fixed-order Welford merge, updates running_mean/var (momentum, unbiased var) and num_batches_tracked, and writes
 * out = act(gamma*(y-mean)*invstd + beta) in `out_dtype`.  save_mean/save_invstd (channels) are kept for backward.
 * Covers BatchNorm1d+Sigmoid (network_tests.py:78-79,113-114; rows=B) and BatchNorm2d+ReLU on channels-last
 * activations (SIMNN.py:105-108; rows=B*H*W).  workspace >= gdm_bn_workspace_bytes(rows, channels).
 * training=0 normalises with the running statistics instead and leaves them untouched.                            */
size_t gdm_bn_workspace_bytes(int rows, int channels);
int gdm_bn_act_fwd(const float* y, int rows, int channels, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                   int act, void* out, int out_dtype, float* save_mean, float* save_invstd, int training,
                   void* workspace, size_t workspace_bytes, void* stream);
/* dout: gradient w.r.t. `out` (out_dtype); out: the forward's output (used for act'); y: the pre-norm input.
 * Writes dy (rows,channels) fp32 and dgamma/dbeta (channels).                                                     */
int gdm_bn_act_bwd(const void* dout, const void* out, int out_dtype, const float* y, int rows, int channels,
                   const float* gamma, const float* save_mean, const float* save_invstd, int act, float* dy,
                   float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* The statistics half of gdm_bn_act_fwd alone (training mode): per-channel batch mean / 1/sqrt(var+eps) into save_*,
 * running statistics and num_batches_tracked updated; the consumer applies the normalisation itself (the fused generator
 * kernels below do it while they load their input).  gdm_bn_finalize merges `chunks` x channels (n, mean, M2) partials
 * laid out [chunk][channel][3] -- what gdm_simnn_gen_convt_bn leaves -- in chunk order.                              */
int gdm_bn_stats(const float* y, int rows, int channels, float* running_mean, float* running_var,
                 int64_t* num_batches_tracked, float momentum, float eps, float* save_mean, float* save_invstd,
                 void* workspace, size_t workspace_bytes, void* stream);
/* Batch statistics that span several calls (data-parallel ranks, SURVEY.md 8e "exact mode"): gdm_bn_partials writes the
 * per-row-chunk Welford triples (n, mean, M2) of y (rows, channels) into partials (gdm_bn_partial_chunks(rows) x channels
 * x 3 floats); the partials of all ranks, concatenated along the chunk axis (all-gather), go through gdm_bn_finalize with
 * the GLOBAL row count; gdm_bn_apply = act((y - mean) * invstd * gamma + beta). */
int gdm_bn_partial_chunks(int rows);
int gdm_bn_partials(const float* y, int rows, int channels, float* partials, void* stream);
int gdm_bn_apply(const float* y, int rows, int channels, const float* gamma, const float* beta, const float* mean,
                 const float* invstd, int act, void* out, int out_dtype, void* stream);
int gdm_bn_finalize(const float* ws, int chunks, int rows, int channels, float momentum, float eps, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, void* stream);

/* ---- model 1 generator, layers 2..4 fused (GAN_DES/SIMNN.py:105-110; forward only, training-mode BatchNorm, the
 * reference's default geometry: 128 -> 64 -> 32 -> 1 channels, 4x4 -> 8x8 -> 16x16 -> 20x20).  Activations are
 * channels-last fp32 row matrices (B*H*W, C) holding the PRE-normalisation convolution outputs.
 * gdm_simnn_gen_pack: conv1.weight (noise_dim,128,4,4), conv2.weight (128,64,4,4), conv3.weight (64,32,4,4) -> bf16
 *   GEMM images (conv1: [pos*128+co][k], conv2/3: parity classes).
 * gdm_simnn_gen_first: ConvTranspose2d(noise_dim -> 128, k4) on the 1x1 noise = a GEMM, y1 (B*16, 128), plus the exact
 *   training-mode BatchNorm statistics of its 128 channels (save_mean / save_invstd, running statistics,
 *   num_batches_tracked) in the same launch: a workgroup owns 16 channels for the whole batch (2 <= B <= 256).
 * gdm_simnn_gen_convt_bn(layer 2|3): BatchNorm(mean, invstd, gamma, beta) + ReLU applied to yin while it is staged,
 *   ConvTranspose2d(k4,s2,p1) as four 2x2-tap implicit GEMMs on bf16 MFMA -> yout, plus one (n, mean, M2) partial per
 *   workgroup and output channel in ws_partials (gdm_simnn_gen_convt_chunks(layer, B) x Cout x 3 floats) for
 *   gdm_bn_finalize.
 * gdm_simnn_gen_last: BatchNorm + ReLU on load, ConvTranspose2d(32 -> 1, k5, s1, p0), sigmoid -> out (B, 20*20).     */
size_t gdm_simnn_gen_pack_bytes(void);
int gdm_simnn_gen_pack(const float* w1, int noise_dim, const float* w2, const float* w3, void* pack, void* stream);
int gdm_simnn_gen_first(const float* noise, int B, int noise_dim, const void* pack, float* y1, float momentum, float eps,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean,
                        float* save_invstd, void* stream);
int gdm_simnn_gen_convt_chunks(int layer, int B);
int gdm_simnn_gen_convt_bn(int layer, const float* yin, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, int B, const void* pack, float* yout, float* ws_partials, void* stream);
int gdm_simnn_gen_last(const float* yin, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       const float* w4, int B, float* out, void* stream);


/* ---- elementwise helpers ---------------------------------------------------------------------------------------*/
/* out = act(x + bias[col]) and its backward dx = dout * act'(out); (rows, cols) row-major. */
int gdm_bias_act_fwd(const float* x, const float* bias, int rows, int cols, int act, float slope, void* out,
                     int out_dtype, void* stream);
int gdm_act_bwd(const void* dout, const void* out, int dtype, int64_t n, int act, float slope, void* dx, void* stream);
/* column sums of a (rows, cols) matrix: out[c] = sum_r x[r,c] (bias gradients). deterministic two-stage. */
int gdm_colsum(const void* x, int dtype, int rows, int cols, float* out, void* workspace, size_t workspace_bytes,
               void* stream);
int gdm_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* counter[0] (device int) += number of NaN / +-Inf elements of x (bit test: the library's arithmetic is compiled
 * without NaN semantics).  What torch.autograd.set_detect_anomaly(True) (network_tests.py:211) turns into an exception
 * in the reference is found with this: the host side checks inputs, parameters, losses and gradients when anomaly
 * mode is on, or on request (check_finite()). */
int gdm_nonfinite_count(const void* x, int dtype, int64_t n, int* counter, void* stream);

/* ---- model 1 discriminator, convolution trunk (SIMNN.py:123-125,136-139) ---------------------------------------
 * conv1: Conv2d(1,16,k2,s1,p1)+ReLU+MaxPool2 fused (aten::convolution/relu/max_pool2d_with_indices):
 *   x (B,H,W) fp32 -> p1 (B,H1,W1,16) channels-last `dtype`, code1 = B*H1*4*Q1 uint64 (Q1 = ceil(W1/4); 8 bytes per
 *   pooled pixel, rows padded to whole quads of four pixels), an opaque image of 16-bit fields, one per (pixel, channel
 *   group g of four): nibble k of a field belongs to channel 4g+k, bits [1:0] = argmax position (dy*2+dx, first maximum in
 *   scan order), bit 2 = that channel passes gradient (pooled value > 0); fields are ordered
 *   [image][pooled row][quad = pw/4][g][pw%4], pixels >= W1 of a row's last quad hold 0; H1=(H+1)/2, W1=(W+1)/2.
 *   code1 is only ever read back by gdm_simnn_conv1_bwd_weight / _bwd_data / gdm_simnn_conv2_bwd_fused.
 * conv2: Conv2d(16,32,k3,s1,p1)+ReLU+MaxPool2 fused, implicit GEMM on MFMA:
 *   p1 -> p2 (B,H2,W2,32) channels-last, code2 (B,H2,W2,16) uint8: one byte per channel pair (2j, 2j+1) =
 *   8 * (c_even + 5 * c_odd), c = 0..3 argmax position in scan order, 4 = ReLU-dead; H2=H1/2, W2=W1/2.  The reference flattens channel-major (x.view(-1, 32*32*54), SIMNN.py:139);
 *   the host keeps fc1's weight permuted to the channels-last order instead (gdm_permute_pc), which is the same
 *   linear map.
 * conv2 weights are consumed from a packed image built by gdm_simnn_conv2_pack (forward and flipped-backward MFMA
 * operand layouts, in `dtype`); rebuild it whenever the weights change.                                            */
int gdm_simnn_conv1_fwd(const float* x, const float* w, const float* bias, int B, int H, int W, void* p1,
                        uint64_t* code1, int dtype, void* stream);
/* the same for a batch that is the concatenation of two input tensors, images 0 .. bsplit-1 from x0 and bsplit .. B-1
 * from x1 (the discriminator step's [real ; generated] batch, SIMNN.py:306-310, in one launch, without a torch.cat) */
int gdm_simnn_conv1_fwd_pair(const float* x0, const float* x1, int bsplit, const float* w, const float* bias, int B,
                             int H, int W, void* p1, uint64_t* code1, int dtype, void* stream);
size_t gdm_simnn_conv2_pack_bytes(int dtype);
int gdm_simnn_conv2_pack(const float* w, int dtype, void* pack, void* stream);
int gdm_simnn_conv2_fwd(const void* p1, const void* pack, const float* bias, int B, int H1, int W1, void* p2,
                        uint8_t* code2, int dtype, void* stream);
/* backward of the conv2 block w.r.t. its input: dp2 (B,H2,W2,32) + code2 -> dp1 (B,H1,W1,16).  conv1's ReLU/pool
 * routing is not applied here; gdm_simnn_conv1_bwd_weight applies it through code1. */
int gdm_simnn_conv2_bwd_data(const void* dp2, const uint8_t* code2, const void* pack, int B, int H1, int W1, void* dp1,
                             int dtype, void* stream);
/* the same data gradient with conv1's weight gradient fused into its epilogue: dp1 is routed through code1 and
 * contracted with the input windows while it is still in registers, so it never goes to HBM (dp1_or_null = NULL).
 * Samples [0,bsplit) read their input from x0, samples [bsplit,B) from x1 (the 2B batch [real ; fake]).
 * The kernel leaves one slab of 80 partial sums per workgroup in `workspace`
 * (>= gdm_simnn_conv2_bwd_fused_workspace_bytes(B,H1,W1)); gdm_simnn_conv2_bwd_fused_finish (same B,H1,W1,workspace)
 * adds the slabs in fixed order into dw1 (16,1,2,2), db1 (16).                                                      */
size_t gdm_simnn_conv2_bwd_fused_workspace_bytes(int B, int H1, int W1);
int gdm_simnn_conv2_bwd_fused(const void* dp2, const uint8_t* code2, const void* pack, int B, int H1, int W1,
                              const uint64_t* code1, const float* x0, const float* x1, int bsplit, int H, int W,
                              void* dp1_or_null, int dtype, void* workspace, size_t workspace_bytes, void* stream);
int gdm_simnn_conv2_bwd_fused_finish(int B, int H1, int W1, float* dw1, float* db1, void* workspace,
                                     size_t workspace_bytes, void* stream);
/* dW2 (32,16,3,3), db2 (32): deterministic slab reduction; workspace >= gdm_simnn_conv2_bwd_weight_workspace_bytes */
size_t gdm_simnn_conv2_bwd_weight_workspace_bytes(int B, int H1, int W1);
int gdm_simnn_conv2_bwd_weight(const void* dp2, const uint8_t* code2, const void* p1, int B, int H1, int W1,
                               float* dw, float* db, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* dW1 (16,1,2,2), db1 (16) from dp1 routed through code1 (pool argmax + ReLU mask) and the input x. */
size_t gdm_simnn_conv1_bwd_weight_workspace_bytes(int B, int H, int W);
int gdm_simnn_conv1_bwd_weight(const void* dp1, const uint64_t* code1, const float* x, int B, int H, int W,
                               float* dw, float* db, int dtype, int accumulate, void* workspace, size_t workspace_bytes,
                               void* stream);
/* dx (B,H,W) fp32 = gradient w.r.t. the discriminator's spectrogram input (aten::convolution_backward's input
 * gradient of SIMNN.py:136 behind the ReLU/pool routing of code1); dp1 (B,H1,W1,16) as written by
 * gdm_simnn_conv2_bwd_fused(dp1_or_null != NULL) or gdm_simnn_conv2_bwd_data; w = conv1.weight (16,1,2,2). */
int gdm_simnn_conv1_bwd_data(const void* dp1, const uint64_t* code1, const float* w, int B, int H, int W, float* dx,
                             int dtype, void* stream);

/* disc_opt.step() of model 1 (SIMNN.py:316) in ONE launch: Adam on fc1.weight with the two layout changes of
 * gdm_adam_step_dev_pc (gradient and operand copy in (N,P,C) order) + Adam on the n_small remaining parameters (one
 * contiguous range p_small / g_small / m_small / v_small that contains conv2.weight) + the rebuild of conv2's packed
 * images (gdm_simnn_conv2_pack) from the updated weights.  hyper: the 8-float device record of gdm_adam_step_dev (its
 * step counter is advanced); done: a device record of GDM_SIMNN_ADAM_RECORD_INTS ints owned by this optimizer (two-level
 * completion counters and the cached bias-correction terms of the next step), zero before the first launch and to be
 * zeroed again whenever the host rewrites `hyper`.  Bit-identical to the sequence gdm_adam_step_dev(small) +
 * gdm_adam_step_dev_pc(big) + gdm_simnn_conv2_pack. */
#define GDM_SIMNN_ADAM_RECORD_INTS 1056
int gdm_simnn_adam_step(float* p_big, const float* g_big_pc, float* m_big, float* v_big, int N, int C, int P,
                        void* shadow_pc, float* p_small, const float* g_small, float* m_small, float* v_small,
                        int n_small, const float* conv2_weight, void* pack, int dtype, float* hyper, int* done,
                        void* stream);

/* head of model 1's discriminator, forward + loss + backward in one launch (SIMNN.py:140-141, 289/311/329):
 * h1 (n,128) fp32 = relu(fc1) -> prob (n) = sigmoid(fc2), loss[0] (+)= sum over the two label halves of the batch
 * means of BCEWithLogits(prob, y) (rows [0,n0) label y0, rows [n0,n) label y1; the sigmoid OUTPUT is fed to the
 * logits loss exactly like the reference does), and if dh1 != NULL the gradients dh1 (n,128) (already through fc1's
 * ReLU), dw2 (128), db2 (1), db1 (128 = column sums of dh1).  Row slices of 32 are reduced per workgroup, slices are
 * summed in order by a second launch.  workspace >= gdm_simnn_head_workspace_bytes(n).                              */
size_t gdm_simnn_head_workspace_bytes(int n);
int gdm_simnn_head(const float* h1, const float* w2, const float* b2, int n, int n0, float y0, float y1, float* prob,
                   float* loss, int accumulate_loss, void* dh1, int dh1_dtype /* GDM_F32 | GDM_BF16: the fc1 GEMMs' operand */,
                   float* dw2, float* db2, float* db1, void* workspace, size_t workspace_bytes, void* stream);

/* ---- model 2 -----------------------------------------------------------------------------------------------------
 * One generator block in one launch: out = act(BatchNorm1d(x W^T + b)) for up to gdm_linear_bn_act_max_rows() rows
 * (network_tests.py:75-80,110-115: aten::addmm + native_batch_norm + sigmoid).  x (M,K), w (N,K) fp32 row-major;
 * bf16 MFMA operands, fp32 accumulation, exact two-pass batch statistics kept inside the workgroup that owns the
 * columns; running statistics / num_batches_tracked updated in training mode.  y_out (M,N) = pre-norm values,
 * optional (backward needs them).
 * groups >= 1: x, y_out, out hold `groups` batches of M rows one after the other (save_mean / save_invstd: groups x N);
 * every batch is normalised with its own statistics and the running statistics take the updates in batch order -- the
 * two forwards a generator makes per training iteration (network_tests.py:294, 312) in one launch.  stat_repeats >= 1
 * applies each running-statistics update that many times (a forward repeated on identical inputs).                  */
typedef struct gdm_linear_bn_job {      /* the arguments of gdm_linear_bn_act_fwd that differ between blocks */
  const float *x, *w, *bias, *gamma, *beta;
  float *running_mean, *running_var;
  int64_t* num_batches_tracked;
  float *y_out, *out, *save_mean, *save_invstd;
  int M, N, K, groups, stat_repeats;
} gdm_linear_bn_job;
int gdm_linear_bn_act_max_rows(void);
/* 1 or 2 independent blocks in ONE launch (the k-th blocks of model 2's two generators, network_tests.py:186-187: the
 * generators have the same depth and do not depend on each other).  Same arithmetic per job as gdm_linear_bn_act_fwd. */
int gdm_linear_bn_act_fwd_multi(const gdm_linear_bn_job* jobs, int n_jobs, float momentum, float eps, int act,
                                int training, void* stream);
/* out_j (M_j, Ka_j + Kb_j) = [a_j | b_j] row by row for up to four jobs in one launch (torch.cat(dim=1) of the
 * generators' inputs, network_tests.py:87, 119: noise next to the conditioning vector).  b_j may be NULL (copy).   */
typedef struct gdm_concat_job {
  const float *a, *b;
  float* out;
  int M, Ka, Kb;
} gdm_concat_job;
int gdm_concat_cols_multi(const gdm_concat_job* jobs, int n_jobs, void* stream);
int gdm_linear_bn_act_fwd(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                          float eps, int act, int training, int M, int N, int K, float* y_out, float* out,
                          float* save_mean, float* save_invstd, int groups, int stat_repeats, void* stream);
/* DiscriminatorCNN(roll_size=(2,128,T)) forward + BCE-with-logits loss + full backward as one persistent kernel that
 * keeps a whole sample and its activations in LDS (network_tests.py:147-160 and the criterion calls at 304-305, 313;
 * replaces aten::convolution x2, leaky_relu x2, addmm, binary_cross_entropy_with_logits and their backwards).
 * Samples [0,bsplit) are read from xa (bsplit,2,128,T) with label ya; samples [bsplit,B) from the two planes p0, p1
 * (B-bsplit,128,T) (piano_roll, durations: the reference's stack().permute() view, network_tests.py:290) with label
 * yb.  loss[0] (+)= mean_a + mean_b; logits (B).  want_grad: dw1 (16,2,4,4), db1, dw2 (32,16,4,4), db2, dwfc (1,K),
 * dbfc.  Weights come from gdm_dcnn_pack (bf16 MFMA images + permuted fc weight + biases; rebuild after every update).
 * bf16 only; gdm_dcnn_fused_supported(T) tells whether the roll length fits (T = 50: yes).                           */
int gdm_dcnn_fused_supported(int T);
size_t gdm_dcnn_pack_bytes(int T);
int gdm_dcnn_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* wfc,
                  const float* bfc, int T, void* pack, void* stream);
size_t gdm_dcnn_fused_workspace_bytes(int B, int T, int want_grad);
int gdm_dcnn_fused(const float* xa, int bsplit, const float* p0, const float* p1, int B, int T, float ya, float yb,
                   const void* pack, float* logits, float* loss, int accumulate_loss, int want_grad, float* dw1,
                   float* db1, float* dw2, float* db2, float* dwfc, float* dbfc, void* workspace,
                   size_t workspace_bytes, void* stream);
/* The same pass with the optimizer step fused into the gradient's final summation (one rank: nothing to exchange between
 * gradient and update): disc_opt.step() (network_tests.py:308, torch.optim.Adam) is applied by the threads that finish
 * the gradient elements, and the updated weights are written straight into `pack` (in place: the next launch reads
 * them).  param / exp_avg / exp_avg_sq: w1, b1, w2, b2, wfc, bfc; hyper = the 8-float device record of
 * gdm_adam_step_dev (its step counter is advanced); done = one device int, zero before the first launch. */
typedef struct gdm_dcnn_adam {
  float* param[6];
  float* exp_avg[6];
  float* exp_avg_sq[6];
  float* hyper;
  int* done;
} gdm_dcnn_adam;
int gdm_dcnn_fused_adam(const float* xa, int bsplit, const float* p0, const float* p1, int B, int T, float ya, float yb,
                        void* pack, float* logits, float* loss, int accumulate_loss, float* dw1, float* db1, float* dw2,
                        float* db2, float* dwfc, float* dbfc, const gdm_dcnn_adam* opt, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- deterministic discrete-event simulator core (host code; SIMULATOR/simulation_v3.py:25-74, 426-743) -------------
 * One run of Sim(adj, distributions, queue_list, seeds=[seed], logging_mode='Music').run(number_of_customers) for 'normal'
 * distributions (loc[i], scale[i] per node; what both bridges construct, matrix_sim_process.py:72-74 / 50-54) and
 * probability routing.  adj (dim,dim) row-major float64: diagonal > 0 marks a source, <= 0 a server; queue_cap[i] =
 * queue_list[i].  The per-node generators are numpy legacy RandomStates seeded from RandomState(seed).randint(3, 9999999)
 * like the reference's; routing draws come from the GLOBAL numpy stream, whose state travels in and out through
 * mt_key[624] / mt_pos / has_gauss / cached_gauss (np.random.get_state() / set_state()).  The run ends when the event
 * list is empty, when number_of_customers customers have been generated (stop_reason 1), or after max_events processed
 * events (stop_reason 2; the reference stops on a wall-clock limit instead, simulation_v3.py:496-499; 0 = no cap).
 * out[0..n_out) = the 'Music' log records in order: kind 0 arrival / 1 departure (value = clock), 2 processing (value =
 * service time).  GDM_EWORKSPACE when out_capacity is too small (n_out then holds the count needed). */
typedef struct gdm_des_event {
  double value;
  int64_t event_id;
  int32_t node;
  int32_t kind;
} gdm_des_event;
int gdm_des_run(const double* adj, int dim, const double* loc, const double* scale, const int32_t* queue_cap, int64_t seed,
                int64_t number_of_customers, int64_t max_events, uint32_t* mt_key, int* mt_pos, int* has_gauss,
                double* cached_gauss, gdm_des_event* out, int64_t out_capacity, int64_t* n_out, int* stop_reason);

/* ---- generic convolution lowering helpers (model 2 discriminator, model 1 generator) ----------------------------
 * im2col for Conv2d fwd / dW and col2im (gather form, deterministic) for Conv2d dX and ConvTranspose2d fwd.
 * Activations are channels-last (B,H,W,C) unless `src_planar` (NCHW fp32 input planes, e.g. the piano-roll).
 * cols: (B*OH*OW, C*KH*KW) row-major with k = (c*KH + kh)*KW + kw -- torch's weight order, so Conv2d weights
 * (Cout, Cin*KH*KW) and ConvTranspose2d weights (Cin, Cout*KH*KW) are GEMM operands in place.                       */
int gdm_im2col(const void* src, int src_dtype, int src_planar, int B, int H, int W, int C, int KH, int KW,
               int stride, int pad, int OH, int OW, void* cols, int cols_dtype, void* stream);
/* dst[b,h,w,c] = sum over (oh,ow,kh,kw) with oh*stride-pad+kh==h, ow*stride-pad+kw==w of cols[(b,oh,ow),(c,kh,kw)].
 * dst_planar: bit 0 = write (B,C,H,W) instead of channels-last; bit 1 = the columns of `cols` are ordered (kh,kw,c)
 * (tap-major: what a GEMM with the weight permuted to (Cin, KH, KW, Cout) produces -- coalesced reads) instead of
 * torch's (c,kh,kw); bits 4-5 = GDM_ACT_* applied to the sum (RELU / SIGMOID; LEAKY is not offered here).            */
int gdm_col2im(const void* cols, int cols_dtype, int B, int H, int W, int C, int KH, int KW, int stride, int pad,
               int OH, int OW, void* dst, int dst_dtype, int dst_planar, void* stream);

/* dst (B,C,P) = src (B,P,C) transposed per batch element, with dtype conversion (channels-last <-> channel-major
 * flatten order: activations, and fc1's weight / weight gradient viewed as (128, 32, H2*W2) <-> (128, H2*W2, 32)). */
int gdm_permute_pc(const void* src, int src_dtype, int B, int P, int C, void* dst, int dst_dtype, void* stream);

/* aten::max_pool2d_with_indices / its backward, kernel 2 stride 2 (floor), channels-last (B,H,W,C) -> (B,H/2,W/2,C)
 * (F.max_pool2d(x, 2, 2) of the SimNN branch, GAN_DES/SIMNN.py:156,158).  idx: window position 0..3 of the first
 * maximum in scan order, one byte per output element (NULL: forward only).                                          */
int gdm_maxpool2_fwd(const void* src, int dtype, int B, int H, int W, int C, void* dst, uint8_t* idx_or_null,
                     void* stream);
int gdm_maxpool2_bwd(const void* dout, int dtype, const uint8_t* idx, int B, int H, int W, int C, void* dx,
                     void* stream);

/* ---- batched DES-matrix prologue (numpy head of matrix_to_midi, MMGAN_MIDI_DES/matrix_sim_process.py:33-117, and of
 * matrix_to_wav, GAN_DES/matrix_sim_process.py:21-95): generated matrix -> what simulation_v3.Sim is constructed from.
 * g: B samples of an (S,S) fp32 matrix, sample_stride floats apart (the generator output in place); dim = number of
 * DES nodes (S - 3 / S - 5); rows dim.. are parameter rows.  The reference interleaves this arithmetic with draws from
 * numpy's global RNG whose consumption depends on the data, so the host advances the stream between the two calls:
 * gdm_des_scan    |m| -> thr_mask (B,S) u8 = |m[dim][x]| > threshold (NULL: skip); instruments (B,dim) i32 =
 *                 int(|m[dim+1][i]| * 126); note_levels (B,dim) i32 = int(|m[dim+2][i]| * 126), with note_mod:
 *                 max(0, . % 128); zero_mask (B,dim) u64, bit x of row i set iff |m[i][x]| == 0 (x < dim);
 *                 norm_aux: aux (B,2,dim) fp32 = rows dim+3, dim+4 divided by their sequential float32 sum over all S
 *                 entries (model 1's distribution rows); flags (B) i32, bit 0 = the sample holds a non-finite value.
 * gdm_des_routing src_mask (B,dim) u8 (1 = source node), residue_col (B,dim) i32 (column that absorbs 1 - sum(row);
 *                 < 0: none) -> out (B,dim,dim) fp64: source columns and diagonal zeroed, rows divided by their
 *                 float64 sum in numpy's pairwise order (0/0 -> 0), residue added, diagonal +1 (source) / -1 (server).
 * S <= 64.  Results are bit-identical to numpy's on finite inputs (tests/golden/des_prologue.npz).                  */
int gdm_des_scan(const float* g, int64_t sample_stride, int B, int S, int dim, float threshold, int note_mod,
                 int norm_aux, uint8_t* thr_mask_or_null, int32_t* instruments, int32_t* note_levels,
                 uint64_t* zero_mask, float* aux_or_null, int32_t* flags, void* stream);
int gdm_des_routing(const float* g, int64_t sample_stride, int B, int S, int dim, const uint8_t* src_mask,
                    const int32_t* residue_col, double* out, void* stream);

/* ---- piano-roll rasteriser (the scatter of generate_piano_roll, MMGAN_MIDI_DES/datasets.py:29-45), batched over files.
 * Messages of row r = file * 128 + note are ev_*[row_ptr[r] .. row_ptr[r+1]) in file order; ev_step = one-second time
 * step of the message, ev_vel = velocity of a note_on (0..127) or -1 for a note_off.  Writes roll, dur (n_files,128,W)
 * fp32: roll[note][step] = last note_on velocity, dur[note][on:off] = off - on of the note_off that closed it.
 * The host has already cut each file's list where the reference's loop stops.                                        */
int gdm_piano_roll_raster(const int32_t* row_ptr, const int32_t* ev_step, const int32_t* ev_vel, int n_files, int W,
                          float* roll, float* dur, void* stream);

/* ---- mel-spectrogram featuriser (GAN_DES/util.py:37-61: torchaudio MelSpectrogram + AmplitudeToDB) --------------
 * The producer of model 1's discriminator input.  DFT and mel filter bank are gdm_gemm calls in exact fp32 (window
 * folded into the [cos | sin] matrix); these three functions are the kernels around them.
 * gdm_stft_frames: x (B windows of L samples, x_stride apart) -> out (B*frames, n_fft) centred frames, reflect padding
 *   (torch.stft center=True, pad_mode="reflect"; frames = 1 + L / hop).
 * gdm_power_spectrum: c (rows, 2*nfreq) = [re | im] -> p (rows, ldp) = re^2 + im^2 (Spectrogram power=2); columns
 *   nfreq..ldp-1 are zero padding for the following GEMM.
 * gdm_power_to_db: mel (B, frames, n_mels) -> out (B, n_mels, frames) = max(10 log10(max(mel, amin)),
 *   window max - top_db)  (AmplitudeToDB(stype="power", top_db); top_db < 0 = no floor).                             */
int gdm_stft_frames(const float* x, int B, int64_t L, int64_t x_stride, int hop, int n_fft, int frames, float* out,
                    void* stream);
int gdm_power_spectrum(const float* c, int64_t rows, int nfreq, int ldp, float* p, void* stream);
int gdm_power_to_db(const float* mel, int B, int frames, int n_mels, float top_db, float amin, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDM_H */
