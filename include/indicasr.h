/*
 * indicasr.h -- C ABI of libindicasr_hip.so: hand-written HIP/CDNA4 (gfx950) kernels for the
 * Conformer hybrid RNNT-CTC + continual-learning training step of FrozenWolf-Cyber/Indic-CL-ASR.
 *
 * Conventions (SURVEY.md §8b "C ABI the replacement exports"):
 *   - plain C symbols, plain pointers and sizes; every pointer is a DEVICE pointer unless it says host;
 *   - the CALLER owns every buffer including the workspace (size from the pure ia_*_workspace_bytes());
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*), never synchronise, never allocate;
 *   - no global mutable state: re-entrant across streams, one host thread per process as in the reference;
 *   - return IA_OK (0) or a negative ia_status; mirrors the reference's RNNTStatus SUCCESS / INVALID_VALUE
 *     (NeMo/nemo/collections/asr/parts/numba/rnnt_loss/utils/global_constants.py:66-68), which its Python
 *     side turns into RuntimeError (rnnt.py:84-85,115-116) -- ours does the same in _lib.py.
 *
 * Reference paths cited below: K/ = NeMo/nemo/collections/asr/parts/numba/rnnt_loss/,
 *                              A/ = NeMo/nemo/collections/asr/, R/ = repository root of the reference.
 */
#ifndef INDICASR_H
#define INDICASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ia_stream_t; /* hipStream_t */

enum ia_status {
    IA_OK = 0,
    IA_INVALID_VALUE = -1,      /* bad dims / null pointer / misaligned buffer */
    IA_WORKSPACE_TOO_SMALL = -2,
    IA_LAUNCH_FAILED = -3,      /* hipGetLastError() != hipSuccess after enqueue */
    IA_UNSUPPORTED = -4,        /* size outside what the kernels were built for (documented per entry) */
};

/* Library / build identification. Returns a static string such as "indicasr-hip gfx950 r1". */
const char* ia_version(void);

/* ------------------------------------------------------------------------------------------------
 * RNNT (transducer) loss, fused log-softmax front end + alpha/beta wavefront + fused gradient.
 * Replaces: GPURNNT.compute_cost_and_score  K/utils/cuda_utils/gpu_rnnt.py:125-231  and its five Numba
 *           kernels (reduce_max/reduce_exp K/utils/cuda_utils/reduce.py:299-360, compute_alphas/betas/grad
 *           K/utils/cuda_utils/gpu_rnnt_kernel.py:73-407, compute_costs_data K/utils/rnnt_helper.py:106-116),
 *           reached from rnnt_loss_gpu K/rnnt.py:138-236 and _RNNTNumba.forward K/rnnt_pytorch.py:40-91.
 *
 * logits  [B,T,U1,V] f32 contiguous, 16-byte aligned (raw joint output, NOT log-softmaxed: GPU semantics)
 * labels  [B,U1-1] i64;  act_lens [B] i64 (1..T);  label_lens [B] i64 (0..U1-1)
 * costs   [B] f32 out:  -(log-likelihood)*(1+fastemit)
 * grads   [B,T,U1,V] f32 out, d(cost_b)/d(logits); zero outside each utterance's valid lattice.
 *         May be NULL (forward/score only) and MAY ALIAS `logits` (in-place).
 * clamp   <= 0 disables gradient clamping (reference: clamp > 0.0 enables, gpu_rnnt_kernel.py:399-403).
 * Limits: U1 <= 1024 (IA_UNSUPPORTED beyond).
 * Workspace: ia_rnnt_workspace_bytes(B,T,U1) bytes, 256-byte aligned.
 */
size_t ia_rnnt_workspace_bytes(int B, int T, int U1);
int ia_rnnt_loss(const float* logits, const int64_t* labels, const int64_t* act_lens, const int64_t* label_lens,
                 int B, int T, int U1, int V, int blank, float fastemit_lambda, float clamp,
                 float* costs, float* grads, void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* The same computation split at the autograd boundary (what the product's training step calls):
 *   ia_rnnt_forward   K1 (denominators + gathers) + K2 (alpha; beta too when need_backward) + costs.
 *   ia_rnnt_backward  per-cell scalars + the streaming gradient kernel, using the state ia_rnnt_forward left in
 *                     `workspace` for the SAME logits.  cost_grad [B] f32 (device; NULL = ones) is the upstream
 *                     d(loss)/d(cost_b): it is folded into the kernel, so the lattice is written exactly once
 *                     (the reference multiplies the stored grads by grad_output in a second pass,
 *                     K/rnnt_pytorch.py:88-91).  The reference clamps BEFORE that scaling, so clamp > 0 requires
 *                     cost_grad == NULL (IA_INVALID_VALUE otherwise; scale the result afterwards).
 *                     grads may alias logits.  ev_start/ev_stop: optional CALLER-OWNED hipEvent_t
 *                     (void*, NULL to skip) recorded on `stream` right around the gradient kernel (rnnt_grad) so a
 *                     benchmark can time the HBM-bound kernel inside a live training step.
 */
int ia_rnnt_forward(const float* logits, const int64_t* labels, const int64_t* act_lens, const int64_t* label_lens,
                    int B, int T, int U1, int V, int blank, float fastemit_lambda, int need_backward, float* costs,
                    void* workspace, size_t workspace_bytes, ia_stream_t stream);
int ia_rnnt_backward(const float* logits, const int64_t* labels, const int64_t* act_lens, const int64_t* label_lens,
                     int B, int T, int U1, int V, int blank, float fastemit_lambda, float clamp,
                     const float* cost_grad, float* grads, void* workspace, size_t workspace_bytes, ia_stream_t stream,
                     void* grad_kernel_start_event, void* grad_kernel_stop_event);
/* Test/debug helper: copies the forward/backward variables left in `workspace` by ia_rnnt_loss into
 * dense [B,T,U1] f32 tensors (zero outside the valid lattice), the layout of the reference's
 * alphas/betas workspace (gpu_rnnt.py:267-293) that its kernel tests compare
 * (NeMo/tests/collections/asr/numba/rnnt_loss/utils/test_gpu_rnnt_kernel.py:52-188). */
int ia_rnnt_export_alphas_betas(const void* workspace, size_t workspace_bytes, const int64_t* act_lens,
                                const int64_t* label_lens, int B, int T, int U1, float* alphas, float* betas,
                                ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused RNNT joint (f16 matrix-core path).  Replaces RNNTJoint.joint_after_projection A/modules/rnnt.py:1587-1665
 * (f.unsqueeze(2)+g.unsqueeze(1) -> ReLU -> Dropout -> Linear(H->V)) fused with the loss front end
 * (reduce_max/reduce_exp K/utils/cuda_utils/reduce.py:299-360 and the blank/label gathers of logp()
 * K/utils/cuda_utils/gpu_rnnt_kernel.py:41-64).  The [B,T,U1,H] hidden tensor is never materialised in forward.
 *
 *   f [B,T,H] f16 (encoder projection), g [B,U1,H] f16 (prediction projection), W [272,H] f16 (rows >= V zero; the
 *   caller pre-multiplies by 1/(1-dropout_p)), bias [V] f32.  Limits: V <= 272, H % 64 == 0 (IA_UNSUPPORTED else).
 *   logits out: [B*T*U1, LD] f16, LD = ia_joint_ld(V) (row stride in elements, 16-byte rows, columns >= V hold -65504: exp() of them is exactly 0);
 *               rows outside an utterance's valid lattice are left untouched.
 *   workspace : same layout/size as ia_rnnt_workspace_bytes(B,T,U1); receives the denominators and the
 *               diagonal-major blank/label log-prob side arrays, i.e. exactly the state ia_rnnt_forward's first
 *               kernel leaves, so ia_rnnt_lattice() continues from here.
 *   dropout   : counter-based keep mask keyed by (seed, lattice cell, hidden unit / 8), regenerated identically by
 *               ia_joint_hidden / ia_joint_dh_reduce; keep probability 1 - round(256 p)/256.
 * ia_rnnt_lattice: alpha (and beta when need_backward) wavefront + costs from the side arrays in `workspace`.
 * ia_joint_backward_g: G = kappa * cost_grad[b] * d cost_b / d logits written IN PLACE over the f16 logits rows
 *               (zero outside the lattice and in the pad columns) -- a GEMM-ready [B*T*U1, LD] f16 operand.
 *               kappa > 0: power-of-two range scale chosen by the caller so that kappa*|cost_grad| ~ 1.
 *               ev_start/ev_stop: optional caller-owned hipEvent_t recorded around the streaming gradient kernel.
 * ia_joint_hidden: hidden[cell, 0:H] = keep * relu(f+g) (un-scaled), hidden[cell, H] = 1, rest of the LDH-wide row 0.
 * ia_joint_dh_reduce: d f[b,t,:] = (1/kappa) sum_u mask * dH[b,t,u,:], d g[b,u,:] = (1/kappa) sum_t ... (both written;
 *               16-frame partial rows of d g in `scratch` = ia_joint_dh_reduce_scratch_bytes, then a finishing sum: no
 *               atomics, deterministic).  dH = G @ W[0:LD,:] is a library GEMM by the caller.
 */
int ia_joint_ld(int V);
int ia_joint_fwd(const void* f, const void* g, const void* W, const float* bias, const int64_t* labels,
                 const int64_t* act_lens, const int64_t* label_lens, int B, int T, int U1, int H, int V, int blank,
                 float dropout_p, unsigned seed, void* logits, int LD, void* workspace, size_t workspace_bytes,
                 ia_stream_t stream);
/* ia_joint_fwd_box: the same, but the logits of every cell with t < box_t[b], u < box_u[b] are kept (box >= the valid
 * lattice; NULL, NULL = ia_joint_fwd): the MAS / LwF terms of the CL scripts read the whole narrowed sub-batch tensors
 * the reference stashes (A/modules/rnnt.py:1463-1496), padded cells included.
 *
 * Continual-learning terms on the f16 lattice (csrc/joint_extra.hip), per-utterance weights w_sq / w_kd [B] f32:
 *   ia_joint_extra_reduce  sums4[0] = sum_b w_sq[b] sum_{box_b, v<V} z^2       (MAS importance, cl_baseline_mas.py:258-265)
 *                          sums4[1] = sum_b w_kd[b] sum_{box_b, v<V} e^t (t-z)  (LwF: F.kl_div(z, exp(t)), cl_baseline_lwf.py:242-257)
 *                          sums4[2] = max |z|, sums4[3] = max t (the host sizes the f16 gradient scale from them)
 *                          teacher = NULL skips the second; scratch: f32 x ia_joint_extra_scratch_elems().
 *   ia_joint_extra_grad    E [cells, LD] f16 = upstream2[0]*w_sq*2z - upstream2[1]*w_kd*e^t inside the box, 0 elsewhere
 *                          (upstream2: 2 device floats, already multiplied by the gradient scale kappa of the lattice)
 *   ia_lattice_add_f16     G += E (n f16 elements, n % 8 == 0): folded into the transducer gradient before the joint's
 *                          backward kernels, which are then given the box extents as lengths. */
int ia_joint_fwd_box(const void* f, const void* g, const void* W, const float* bias, const int64_t* labels,
                     const int64_t* act_lens, const int64_t* label_lens, const int64_t* box_t, const int64_t* box_u, int B,
                     int T, int U1, int H, int V, int blank, float dropout_p, unsigned seed, void* logits, int LD,
                     void* workspace, size_t workspace_bytes, ia_stream_t stream);
int64_t ia_joint_extra_scratch_elems(void);
int ia_joint_extra_reduce(const void* logits, const void* teacher, const int64_t* box_t, const int64_t* box_u,
                          const float* w_sq, const float* w_kd, int B, int T, int U1, int V, int LD, float* sums4,
                          float* scratch, ia_stream_t stream);
int ia_joint_extra_grad(const void* logits, const void* teacher, const int64_t* box_t, const int64_t* box_u, const float* w_sq,
                        const float* w_kd, const float* upstream2, int B, int T, int U1, int V, int LD, void* E,
                        ia_stream_t stream);
int ia_lattice_add_f16(void* G, const void* E, int64_t n, ia_stream_t stream);
int ia_rnnt_lattice(const int64_t* act_lens, const int64_t* label_lens, int B, int T, int U1, float fastemit_lambda,
                    int need_backward, float* costs, void* workspace, size_t workspace_bytes, ia_stream_t stream);
int ia_joint_backward_g(void* logits_inout, const int64_t* labels, const int64_t* act_lens, const int64_t* label_lens,
                        int B, int T, int U1, int V, int LD, int blank, float fastemit_lambda, const float* cost_grad,
                        float kappa, void* gt_out, int S, int Kc, float* dbias_out, float* dbias_scratch, void* workspace,
                        size_t workspace_bytes, ia_stream_t stream, void* grad_kernel_start_event,
                        void* grad_kernel_stop_event);
/* The same; skip_dead_frames != 0 (fused dbias variant only, i.e. dbias_out != NULL): the kernel neither reads nor zero-fills
 * the 64-cell tiles that lie entirely behind frame act_lens[b] + 7 of an utterance -- valid when G is consumed by
 * ia_joint_dh_fused and ia_joint_dw_fused (with act_lens), which read nothing there. */
int ia_joint_backward_g_skip(void* logits_inout, const int64_t* labels, const int64_t* act_lens, const int64_t* label_lens, int B,
                             int T, int U1, int V, int LD, int blank, float fastemit, const float* cost_grad, float kappa,
                             void* gt_out, int S, int Kc, float* dbias_out, float* dbias_scratch, void* workspace,
                             size_t workspace_bytes, int skip_dead_frames, ia_stream_t stream, void* ev_start, void* ev_stop);
/* dbias_out (optional, NULL to skip; only with gt_out == NULL): receives sum_cells G[cell, v] for v < LD (f32, un-scaled:
 * the bias gradient of the per-language head times kappa), accumulated by the gradient kernel itself in registers;
 * dbias_scratch = ia_joint_backward_g_dbias_scratch_elems(LD) floats of per-workgroup partial rows. */
int64_t ia_joint_backward_g_dbias_scratch_elems(int LD);
/* gt_out (optional, NULL to skip): additionally receives G transposed in the chunked K-contiguous layout
 * GT[s][v][kc] f16 (S chunks of Kc cells, Kc % 64 == 0, S*Kc >= B*T*U1, cells beyond the lattice zero), the A operand
 * of the split-K weight-gradient GEMM  dW[v,h] = sum_s GT[s] @ HT[s]^T  with HT from ia_joint_hidden_t:
 * HT[s][hh][kc] = keep*relu(f+g) for hh < H, 1 for hh == H (dbias row), 0 for H < hh < LDH. */
int ia_joint_hidden_t(const void* f, const void* g, void* hidden_t, int B, int T, int U1, int H, int LDH, int S, int Kc,
                      float dropout_p, unsigned seed, ia_stream_t stream);
int ia_joint_hidden(const void* f, const void* g, void* hidden, int B, int T, int U1, int H, int LDH, float dropout_p,
                    unsigned seed, ia_stream_t stream);
int ia_joint_dh_reduce(const void* dh, const void* f, const void* g, const int64_t* act_lens,
                       const int64_t* label_lens, float* df, float* dg, int B, int T, int U1, int H, float inv_kappa,
                       float dropout_p, unsigned seed, void* scratch, ia_stream_t stream);
size_t ia_joint_dh_reduce_scratch_bytes(int B, int T, int U1, int H);
/* ia_joint_dh_fused: the dH GEMM, the relu/dropout mask and both reductions of ia_joint_dh_reduce in ONE kernel (dH only
 * exists as MFMA accumulators; workgroup = utterance x 16 frames x 320 hidden units, waves partitioned along the hidden
 * axis so both reductions are wave-local).  df[b,t,:] written for t < act_len rounded up to 16 (caller zeroes df);
 * dg[b,u,:] written for every u (16-frame partial rows in `scratch` = ia_joint_dh_fused_scratch_bytes + a finishing
 * sum; no atomics).  G = ia_joint_backward_g's in-place output [B*T*U1, LD] f16; Wt = W transposed and zero padded to
 * [H, ia_joint_dh_k()] f16 (Wt[h][v] = W[v][h], dropout scale folded as in the forward).
 * Supported when ia_joint_dh_fused_supported(U1, H, LD): H % 80 == 0, LD <= 288. */
int ia_joint_dh_fused_supported(int U1, int H, int LD);
int ia_joint_dh_k(void);
size_t ia_joint_dh_fused_scratch_bytes(int B, int T, int U1, int H);
int ia_joint_dh_fused(const void* G, const void* Wt, const void* f, const void* g, const int64_t* act_lens,
                      const int64_t* label_lens, float* df, float* dg, int B, int T, int U1, int H, int LD,
                      float inv_kappa, float dropout_p, unsigned seed, void* scratch, ia_stream_t stream);
/* ... with optional bf16 images of both results (df_bf16 [B,T,H], dg_bf16 [B,U1,H]: the A operands of the joint's enc / pred
 * projection backward GEMMs; the f32 outputs may then be NULL) and zero_dead_df != 0: the finishing kernel zeroes the rows of
 * d f in the 16-frame chunks the main kernel skips, so the caller does not have to clear d f beforehand. */
int ia_joint_dh_fused_ex(const void* G, const void* Wt, const void* f, const void* g, const int64_t* act_lens,
                         const int64_t* label_lens, float* df, float* dg, void* df_bf16, void* dg_bf16, int zero_dead_df, int B,
                         int T, int U1, int H, int LD, float inv_kappa, float dropout_p, unsigned seed, void* scratch,
                         ia_stream_t stream);
/* ia_joint_dw_fused: weight gradient of the per-language Linear(H -> V) of the joint (A/modules/rnnt.py:1694-1703,
 * applied in joint_after_projection :1633-1647) straight from G and the encoder / prediction projections:
 *   dW[v*H + h] = sum_cells G[cell,v] * keep*relu(f[b,t,h] + g[b,u,h])          ([LD, H] f32)
 * (un-scaled: the caller divides by kappa and the dropout keep probability; the bias gradient comes from
 * ia_joint_backward_g's dbias_out).  The hidden tensor is regenerated tile by
 * tile in LDS (same counter-based dropout mask as the forward), so neither hidden^T nor a transposed copy of G exists.
 * Split-K over the lattice cells: partial tiles in `scratch` (ia_joint_dw_fused_scratch_elems floats) + a finishing sum.
 * The steps are laid out per utterance and the utterance's prediction rows g[b, :, tile] stay resident in LDS; with act_lens
 * (optional, may be NULL) the steps behind an utterance's last live frame -- where G is zero -- are skipped.
 * Supported when ia_joint_dw_fused_supported(U1, H, LD): U1 <= 128, H % 8 == 0, LD % 8 == 0, LD <= 288; B*T*U1 < 2^31. */
int ia_joint_dw_fused_supported(int U1, int H, int LD);
int64_t ia_joint_dw_fused_scratch_elems(int B, int T, int U1, int H, int LD);
int ia_joint_dw_fused(const void* G, const void* f, const void* g, const int64_t* act_lens, int B, int T, int U1, int H, int LD,
                      float dropout_p, unsigned seed, float* dW, float* scratch, ia_stream_t stream);
/* The same with the label counts: when act_lens and label_lens are both given (and T >= 8, 8 <= U1 <= 256, T*U1 < 2^23,
 * B <= 4096) a step is an 8-frame x 8-label tile and an utterance's steps cover its live frames x live labels only -- the
 * labels u > label_lens[b] of a live frame hold zeros in G as well (A/parts/numba/rnnt_loss/utils/cuda_utils/
 * gpu_rnnt_kernel.py:351-403 writes nothing there), so they are skipped like the frames behind the utterance's end.
 * Other shapes run the flat 64-cell steps of ia_joint_dw_fused.  G must be finite on the frames act_lens[b] .. act_lens[b] + 7
 * (the last frame tile of an utterance covers up to seven of them; their contribution is switched off through zeroed
 * hidden rows, and 0 x NaN is NaN): ia_joint_backward_g_skip zero-fills exactly those and leaves the rest alone. */
int ia_joint_dw_fused_ex(const void* G, const void* f, const void* g, const int64_t* act_lens, const int64_t* label_lens, int B,
                         int T, int U1, int H, int LD, float dropout_p, unsigned seed, float* dW, float* scratch,
                         ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Conformer block forward building blocks (bf16 projections, fp32 residual stream).
 *
 * ia_gemm_bf16:  out = alpha * dropout(act(A[M,K] @ W[N,K]^T + bias)) + R       (matrix cores, bf16 in / f32 acc)
 *   Replaces nn.Linear / Conv1d(k=1) + the elementwise kernels around them in ConformerFeedForward
 *   (A/parts/submodules/conformer_modules.py:385-404), the attention projections (multi_head_attention.py:69-96,
 *   117-119), the pointwise convolutions (conformer_modules.py:340-366) and the residual updates of
 *   ConformerLayer.forward (:141-214).  A [M,lda] bf16, W [N,ldw] bf16 (nn.Linear layout), bias [N] f32 or NULL,
 *   act 0 none / 1 SiLU / 2 ReLU, dropout keyed by (seed, row, column/8) with keep scale folded in, R [M,ldr] f32
 *   residual or NULL (may alias outF), outF [M,ldof] f32 and/or outH [M,ldoh] bf16.  K % 64 == 0, N % 8 == 0.
 * ia_layernorm:  nn.LayerNorm over the last axis (conformer_modules.py:86-139); optional chained second LayerNorm
 *   (g2,b2) applied to the first one's result; outF receives the FIRST norm's fp32 result, outH the final bf16.
 * ia_glu_dwconv: GLU(dim=channels) -> zero frames >= lens[b] -> depthwise conv1d (ksz odd <= 31, 'same' zero padding)
 *   (conformer_modules.py:345-352, causal_convs.py:72-150).  x2 [B,T,2d] bf16, w [d,ksz] f32, z [B,T,d] f32;
 *   bn_sum/bn_sumsq [d] f32 = per-channel sums over all B*T frames for train-mode BatchNorm (written, not accumulated:
 *   workgroup partial rows in `scratch` -- f32 x ia_dwconv_scratch_elems -- then a column-sum pass; deterministic).
 * ia_bn_silu:    BatchNorm1d (train: batch statistics from the sums, running stats updated with `momentum`,
 *   unbiased variance, num_batches_tracked += 1; eval: running stats) followed by SiLU; out [n_rows,d] bf16
 *   (conformer_modules.py:353-362).
 */
int ia_gemm_bf16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, int act,
                 float dropout_p, unsigned seed, float alpha, const float* R, int ldr, float* outF, int ldof,
                 void* outH, int ldoh, ia_stream_t stream);
/* The same with two more epilogue features (trainable blocks): out_pre [M,N] bf16 = the bias-added value before act /
 * dropout (the activation is applied to that rounded value: one launch instead of GEMM + ia_silu_dropout); act = 3 with aux
 * [M,N] bf16: out = bf16(acc) * SiLU'(aux), then the dropout mask (GEMM + ia_silu_dropout_bwd in one launch); act = 4 (N % 128
 * == 0, bf16 output only, no dropout / residual): GLU over the column halves of every 128-column tile, outH [M, N/2] =
 * (acc[:, 128t+c] + b) * sigmoid(acc[:, 128t+64+c] + b'), c < 64 -- pointwise_conv1 + GLU (conformer_modules.py:340-349) with
 * the weight rows regrouped value | gate per tile (ia_block_params.w_pw1_glu). */
int ia_gemm_bf16_ex(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, int act,
                    float dropout_p, unsigned seed, float alpha, const float* R, int ldr, float* outF, int ldof, void* outH,
                    int ldoh, void* out_pre, int ldpre, const void* aux, int ldaux, ia_stream_t stream);
/* ... with `flags`: bit 0 = outH holds IEEE half instead of bf16 (the joint's f16 operands f = enc(x), g = pred(dec) come
 * straight out of their projections, rnnt.py:1590-1596: no cast pass). */
int ia_gemm_bf16_ex2(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, int act,
                     float dropout_p, unsigned seed, float alpha, const float* R, int ldr, float* outF, int ldof, void* outH,
                     int ldoh, void* out_pre, int ldpre, const void* aux, int ldaux, int flags, ia_stream_t stream);
/* ConvSubsampling 'striding' x4 (A/parts/submodules/subsampling.py:217-253,385-437), channels-last, no transposes:
 *   ia_subsample_conv1: feats [B,Fm,Tm] f32 (preprocessor layout) -> relu(conv 1->C, 3x3, s2, p1) as [B,T1,F1,C] bf16
 *                       (w1 [C,9] f32 = conv.0.weight, b1 [C]); T1 = (Tm-1)/2+1, F1 = (Fm-1)/2+1.
 *   ia_subsample_conv2: implicit-GEMM 3x3 / s2 / p1 convolution C -> N on the matrix cores (A fragments gathered from
 *                       the channels-last image, zero padding by predication, bias + ReLU epilogue):
 *                       in [B,T1,F1,C] bf16, w2r [N, 9*C] bf16 with k = (dt*3+df)*C + ci (= conv.2.weight permuted
 *                       to [N,3,3,C]), out [B,T2,F2,N] bf16.  C % 64 == 0.
 *   The final Linear(C*F2 -> d) is ia_gemm_bf16 on the [B*T2, F2*N] view with the weight's columns permuted from the
 *   reference's (c,f) order to (f,c). */
/* Block-scaled fp8 (MX: e4m3 elements, one e8m0 scale byte per 32 consecutive k) on v_mfma_scale_f32_16x16x128_f8f6f4, the
 * instruction that issues fp8 at twice the bf16 rate (csrc/gemm_mxfp8.hip):
 *   ia_quantize_mxfp8   q [M, ldq] e4m3, scales [M, lds] e8m0 (K / 32 bytes per row used): x ~ q * 2^(scale - 127) per block;
 *                       x bf16 or f32 [M, K], K % 32 == 0, ldq % 16 == 0, lds % 4 == 0
 *   ia_gemm_mxfp8       the operator of ia_gemm_bf16 on such operands; K % 128 == 0, N % 8 == 0. */
int ia_quantize_mxfp8(const void* x, int is_f32, int ld, int64_t M, int K, void* q, int ldq, void* scales, int lds, ia_stream_t stream);
int ia_gemm_mxfp8(const void* Aq, int lda, const void* a_scales, int ldsa, const void* Wq, int ldw, const void* w_scales, int ldsw,
                  int M, int N, int K, const float* bias, int act, float dropout_p, unsigned seed, float alpha, const float* R, int ldr,
                  float* outF, int ldof, void* outH, int ldoh, ia_stream_t stream);

/* Device-resident greedy transducer decoding (csrc/greedy_decode.hip): the frame-synchronous loop of
 * GreedyBatchedRNNTInfer (A/parts/submodules/rnnt_greedy_decoding.py:711-909) in one launch, one persistent workgroup per
 * utterance, no host read per micro-step.  f_all [B,T,Hj] f32 = joint.enc(encoder output); out_len [B]; EW [(V+1), 4Hp] f32 =
 * W_ih embedding[row] + b_ih + b_hh for rows 0..V-2 = the language's labels, row_blank = embedding[blank_idx], row_sos = zero
 * input; Whh [4Hp,Hp], Wpred [Hj,Hp] + bpred, Whead [V,Hj] + bhead (the language's head), all f32.  tokens [B, cap] int32,
 * counts [B]; *overflow set if an utterance emitted more than cap symbols.  Hp, Hj multiples of 4. */
int ia_greedy_decode_lds_bytes(int Hp, int Hj, int V);
int ia_greedy_rnnt_decode(const float* f_all, const int64_t* out_len, const float* EW, const float* Whh, const float* Wpred,
                          const float* bpred, const float* Whead, const float* bhead, int B, int T, int Hp, int Hj, int V,
                          int blank, int row_blank, int row_sos, int max_symbols, int* tokens, int cap, int* counts,
                          int* overflow, ia_stream_t stream);
/* The same loop with W_hh, W_pred and the head as row-major bf16 (the images the training step multiplies with; Hp, Hj
 * multiples of 8): half the bytes per emitted symbol of a loop that is bound by the CU's L2 bandwidth.  Activations, the EW
 * table, biases and every accumulation stay fp32. */
int ia_greedy_rnnt_decode_bf16w(const float* f_all, const int64_t* out_len, const float* EW, const void* Whh_bf16,
                                const void* Wpred_bf16, const float* bpred, const void* Whead_bf16, const float* bhead, int B, int T,
                                int Hp, int Hj, int V, int blank, int row_blank, int row_sos, int max_symbols, int* tokens, int cap,
                                int* counts, int* overflow, ia_stream_t stream);
/* The bf16 decode restructured (same reference loop, rnnt_greedy_decoding.py:711-909; Hj % 32 == 0, Hp % 8 == 0): the head is
 * evaluated for 16 frames at once on the matrix cores (activations rounded to bf16 as the training step's joint does; blanks
 * advance the frame, the first non-blank emits and the evaluation restarts there), and the per-symbol GEMVs are split over
 * `cluster` workgroups per utterance that hand h' and g over through `scratch`.  ia_greedy_decode_cluster = the cluster size
 * the library would pick (0: dimensions not taken, the GEMV loop of ia_greedy_rnnt_decode_bf16w decodes; Hp % (8 cluster) == 0,
 * Hj % (4 cluster) == 0, 8 ceil(B / 8) cluster <= 256); scratch = ia_greedy_decode_scratch_bytes, 16-byte aligned (NULL allowed for
 * cluster <= 1).  *overflow: bit 0 = an utterance emitted more than cap symbols, bit 1 = a cluster hand-off timed out.
 * ia_greedy_rnnt_decode_bf16w itself runs this kernel with cluster = 1 when it takes the dimensions. */
size_t ia_greedy_decode_scratch_bytes(int B, int Hp, int Hj);
int ia_greedy_decode_cluster(int B, int Hp, int Hj);
int ia_greedy_rnnt_decode_bf16w_ex(const float* f_all, const int64_t* out_len, const float* EW, const void* Whh_bf16,
                                   const void* Wpred_bf16, const float* bpred, const void* Whead_bf16, const float* bhead, int B,
                                   int T, int Hp, int Hj, int V, int blank, int row_blank, int row_sos, int max_symbols, int* tokens,
                                   int cap, int* counts, int* overflow, int cluster, void* scratch, size_t scratch_bytes,
                                   ia_stream_t stream);

/* Host-side scoring of the step's monitor (csrc/host_metrics.hip, no device work): edit distances of n hypothesis / reference
 * pairs, units as int32 ids (a[a_off[i] .. a_off[i+1]) against b[b_off[i] .. b_off[i+1])) -- editdistance.eval of
 * A/metrics/wer.py:58-60, batched. */
int ia_edit_distance_batch(const int32_t* a, const int64_t* a_off, const int32_t* b, const int64_t* b_off, int n, int64_t* out);

/* fp8 (OCP e4m3) projections of the frozen prefix (csrc/gemm_fp8.hip; BASELINE configs[4] "fp8 MFMA"; no reference
 * semantics -- tolerance vs the fp32 oracle stated in tests/test_fp8_gpu.py):
 *   ia_quantize_fp8_rows   q [M, ldq] e4m3 = x / scale[m], scale[m] = amax(row m) / 448 (1 for a zero row); x bf16 or f32
 *                          [M, K] (row stride ld), K % 8 == 0, ldq % 16 == 0 (padding bytes zeroed)
 *   ia_gemm_fp8            out = alpha*dropout(act((Aq Wq^T) o a_scale[m] o w_scale[n] + bias)) + R : the operator of
 *                          ia_gemm_bf16 (same epilogue / dropout mask), operands e4m3 with per-row scales; K % 16 == 0. */
int ia_quantize_fp8_rows(const void* x, int is_f32, int ld, int64_t M, int K, void* q, int ldq, float* scale, ia_stream_t stream);
int ia_gemm_fp8(const void* Aq, int lda, const float* a_scale, const void* Wq, int ldw, const float* w_scale, int M, int N, int K,
                const float* bias, int act, float dropout_p, unsigned seed, float alpha, const float* R, int ldr, float* outF,
                int ldof, void* outH, int ldoh, ia_stream_t stream);
int ia_subsample_conv1(const float* feats, int B, int Fm, int Tm, int C, const float* w1, const float* b1, void* out,
                       ia_stream_t stream);
int ia_subsample_conv2(const void* in_cl, int B, int T1, int F1, int C, const void* w2r, const float* b2, int N, void* out,
                       ia_stream_t stream);
/* ia_gemm_bf16_ln: projection into the residual stream + LayerNorm of the updated rows in one launch (N == 256: a workgroup of
 * the 64 x 256-tile kernel owns whole rows):  x = R + alpha * dropout(A @ W^T + bias) -> outF (may alias R),
 * LN(x) * ln_g + ln_b -> outH (bf16).  The attention's out-projection followed by the convolution module's LayerNorm
 * (conformer_modules.py:171-186).  Same operands, dropout mask and summation order as ia_gemm_bf16 + ia_layernorm. */
int ia_gemm_bf16_ln_supported(int N, int K);
int ia_gemm_bf16_ln(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, float dropout_p,
                    unsigned seed, float alpha, const float* R, int ldr, float* outF, int ldof, const float* ln_g,
                    const float* ln_b, float ln_eps, void* outH, int ldoh, ia_stream_t stream);
int ia_layernorm(const float* x, int ldx, int N, int d, const float* g1, const float* b1, float eps, float* outF,
                 int ldf, const float* g2, const float* b2, void* outH, int ldh, ia_stream_t stream);
int ia_glu_dwconv(const void* x2, const int64_t* lens, int B, int T, int d, int ksz, const float* w, const float* bias,
                  float* z, float* bn_sum, float* bn_sumsq, float* scratch, ia_stream_t stream);
int64_t ia_dwconv_scratch_elems(int B, int T, int d, int ksz); /* covers ia_glu_dwconv and ia_dwconv_time_wgrad */
/* ia_colsum_bf16: out[n] += sum_m x[m,n] (f32 atomics, caller zeroes): bias gradient of a projection. */
int ia_colsum_bf16(const void* x, int M, int N, int ld, float* out, ia_stream_t stream);
/* Depthwise conv over time on fp32 [B,T,d] with autograd pieces (CausalConv1D as configured by the Conformer conv
 * module, causal_convs.py:72-150): y = bias + sum_j w[c][j] x[t+j-half]; flip=1 (bias NULL) gives the data gradient;
 * ia_dwconv_time_wgrad writes dw [d,ksz] and db [d] (db may be NULL) through partial rows in `scratch`. */
int ia_dwconv_time(const float* x, int B, int T, int d, int ksz, const float* w, const float* bias, int flip, float* y,
                   ia_stream_t stream);
int ia_dwconv_time_wgrad(const float* x, const float* dy, int B, int T, int d, int ksz, float* dw, float* db,
                         float* scratch, ia_stream_t stream);
/* Backward of GLU -> depthwise conv in two launches (trainable blocks): ia_dwconv_glu_bwd = ia_dwconv_time(dz, flip 1) followed
 * by ia_glu_bwd, without the dG tensor (dc2 [B*T, 2d] bf16); ia_dwconv_glu_wgrad = ia_dwconv_time_wgrad on mask(GLU(c2)),
 * regenerated in the window loads instead of read from ia_glu_mask's output.  Same results as the four-launch sequence. */
int ia_dwconv_glu_bwd(const float* dz, const void* c2, const int64_t* lens, int B, int T, int d, int ksz, const float* w, void* dc2,
                      ia_stream_t stream);
int ia_dwconv_glu_wgrad(const void* c2, const int64_t* lens, const float* dy, int B, int T, int d, int ksz, float* dw, float* db,
                        float* scratch, ia_stream_t stream);
int ia_bn_silu(const float* z, int64_t n_rows, int d, const float* bn_sum, const float* bn_sumsq, const float* gamma,
               const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked,
               float momentum, float eps, int training, void* out, ia_stream_t stream);

/* ia_bn_silu + the pointwise convolution behind it in ONE launch (csrc/gemm_bnsilu.hip; conformer_modules.py:354-366):
 * out = alpha * dropout(SiLU(BN(z))[M,K] @ W[N,K]^T + bias) + R, BN over the K channels from the batch sums (training; fp32
 * bn_sum / bn_sumsq, or the fixed-point accumulators of ia_glu_dwconv_fixed when bn_sums_fixed != NULL) or
 * the running statistics; bit-identical to ia_bn_silu followed by ia_gemm_bf16.  K <= 1024, K % 8 == 0. */
int ia_gemm_bnsilu_supported(int K);
int ia_gemm_bnsilu_bf16(const float* z, int ldz, int64_t n_rows, const float* bn_sum, const float* bn_sumsq,
                        const float* gamma, const float* beta, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float momentum, float eps, int training, const void* W, int ldw, int M,
                        int N, int K, const float* bias, float dropout_p, unsigned seed, float alpha, const float* R, int ldr,
                        float* outF, int ldof, void* outH, int ldoh, const long long* bn_sums_fixed, ia_stream_t stream);
/* ... keeping SiLU(BN(z)) as well (outA [M,K] bf16, written by the first column tile): the trainable blocks' forward, whose
 * backward needs it for the weight gradient of the pointwise convolution. */
int ia_gemm_bnsilu_bf16_keep(const float* z, int ldz, int64_t n_rows, const float* bn_sum, const float* bn_sumsq,
                             const float* gamma, const float* beta, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, float momentum, float eps, int training, const void* W, int ldw, int M,
                             int N, int K, const float* bias, float dropout_p, unsigned seed, float alpha, const float* R, int ldr,
                             float* outF, int ldof, void* outH, int ldoh, const long long* bn_sums_fixed, void* outA, int ldoa,
                             ia_stream_t stream);
/* ia_glu_dwconv with the BatchNorm sums accumulated into 64-bit fixed-point integers (units of 2^-24; 8 copies of
 * [sum(d) | sumsq(d)] = 16 d values, the workgroups spread over the copies, the reader adds them; zeroed by the caller): deterministic without partial rows and a finishing launch.  ia_gemm_bnsilu_bf16 reads them through
 * `bn_sums_fixed` (then bn_sum / bn_sumsq may be NULL). */
/* ... and the same for an input the GLU has already been applied to (g [B*T, d] bf16: ia_gemm_bf16_ex with act 4 on the
 * regrouped pointwise_conv1 weight): frames >= lens[b] read as zero, depthwise conv, z, fixed-point BatchNorm sums. */
int ia_dwconv_gated_fixed(const void* g, const int64_t* lens, int B, int T, int d, int ksz, const float* w, const float* bias,
                          float* z, long long* bn_sums_fixed, ia_stream_t stream);
int ia_glu_dwconv_fixed(const void* x2, const int64_t* lens, int B, int T, int d, int ksz, const float* w, const float* bias,
                        float* z, long long* bn_sums_fixed, ia_stream_t stream);

/* ia_relpos_attention: RelPositionMultiHeadAttention.forward core (A/parts/submodules/multi_head_attention.py:197-250,
 * rel_shift :184-195, masking :108-111) without materialising any [B,h,T,T] / [B,h,T,2T-1] tensor.
 *   qkv [B*T, 3*H*dk] bf16 (q | k | v, head-major inside each third), pos_proj [2T-1, H*dk] bf16 (linear_pos(pos_emb)),
 *   bias_u / bias_v [H,dk] f32, lens [B] i64, ctx out [B*T, H*dk] bf16 (zero rows for queries >= lens[b]).
 *   vt_scratch: ia_attn_vt_elems(B,T,H) bf16 elements, caller-owned (holds V^T, filled by this call).
 *   Attention dropout keyed by (seed, b, h, i, j).  Limits: dk == 64, T <= 384 (IA_UNSUPPORTED otherwise). */
size_t ia_attn_vt_elems(int B, int T, int H);
int ia_relpos_attention(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                        const int64_t* lens, int B, int T, int H, int dk, float dropout_p, unsigned seed,
                        void* vt_scratch, void* ctx, ia_stream_t stream);
/* Backward of ia_relpos_attention, row pass (one wave = 16 queries): recomputes the probabilities and writes three bf16
 * matrices on which every remaining contraction is a plain (batched) GEMM with 16-byte aligned rows:
 *   Pd    [B,H,T,Ts]  dropout(P)                          dV     = Pd^T dctx
 *   dS    [B,H,T,Ts]  P o (keep*dctx V^T - dctx.ctx)/sqrt(dk)   dK = dS^T (q+u),  d(q+u) = dS K
 *   dBand [H,B,T,Rs]  dS skewed to column pad0 + (T-1-i+j)       d(q+v) = dBand p, dp = dBand^T (q+v) summed over B
 * with Ts, Rs, pad0 from ia_relpos_attention_bwd_dims (Ts = ceil8(T), pad0 = (8 - T%8)%8, Rs = ceil8(pad0 + 2T-1)).
 * ctx = the forward's output, dctx its gradient (bf16 [B*T, H*dk]); same seed / dropout_p / limits as the forward. */
int ia_relpos_attention_bwd_dims(int T, int* Ts, int* Rs, int* pad0);
int ia_relpos_attention_bwd(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                            const int64_t* lens, const void* ctx, const void* dctx, int B, int T, int H, int dk,
                            float dropout_p, unsigned seed, void* Pd, void* dS, void* dBand, void* Qu, void* Qv, void* K,
                            void* dO, ia_stream_t stream);
/* The row pass also writes the head-major bf16 operands of those GEMMs: Qu = q+u, K, dO as [B,H,T,dk], Qv = q+v as
 * [H,B,T,dk].  ia_attn_bwd_unpack folds their outputs back: dqkv [B*T, 3*H*dk] bf16 = (dQu + dQv | dK | dV) and
 * dbias_u / dbias_v [H*dk] f32 = column sums of dQu / dQv (block partial rows in `scratch`, f32 x
 * ia_attn_bwd_unpack_scratch_elems).  dQu, dK, dV are [B,H,T,dk], dQv is [H,B,T,dk] (bf16).  H*dk <= 2048. */
int ia_attn_bwd_unpack(const void* dQu, const void* dQv, const void* dK, const void* dV, void* dqkv, float* dbias_u,
                       float* dbias_v, int B, int T, int H, int dk, float* scratch, ia_stream_t stream);
int64_t ia_attn_bwd_unpack_scratch_elems(int B, int T, int H);

/* ------------------------------------------------------------------------------------------------
 * Rel-pos attention forward with a key-tile loop and online softmax (csrc/attention_flash.hip): the same function as
 * ia_relpos_attention (RelPositionMultiHeadAttention.forward, multi_head_attention.py:197-250) without its limits:
 * any T (30 s audio: T' = 751), head dim any multiple of 4 up to 64 (d = 144 / 4 heads = 36), no V^T scratch.
 * qkv [B*T, 3*H*dk] bf16 (q|k|v), pos_proj [>= 2T-1, H*dk] bf16 (row r <-> relative position T-1-r), bias_u/bias_v
 * [H*dk] f32, lens [B] i64 -> ctx [B*T, H*dk] bf16 (rows of padded queries are zero).  Attention dropout draws its
 * own mask (one hash per (head, query, 4 keys)); it is the forward of no-autograd passes (frozen prefix, teacher,
 * eval), so no backward has to reproduce it. */
int ia_relpos_attention_flash_supported(int T, int dk);
int ia_relpos_attention_flash(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                              const int64_t* lens, int B, int T, int H, int dk, float dropout_p, unsigned seed, void* ctx,
                              ia_stream_t stream);
/* The same forward, also returning lse [B*H, T] f32 (log-sum-exp of each query's scaled scores; 0 for padded queries): what
 * the key-tiled backward recomputes the probabilities from. */
int ia_relpos_attention_flash_lse(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                                  const int64_t* lens, int B, int T, int H, int dk, float dropout_p, unsigned seed, void* ctx,
                                  float* lse, ia_stream_t stream);
/* Key-tiled backward of the same attention core (csrc/attention_flash_bwd.hip; autograd of multi_head_attention.py:197-250):
 * no [T,T] matrices in HBM.  In: the forward's qkv / pos_proj / biases / lens / ctx / lse, dctx [B*T, d] bf16 and the
 * forward's dropout p / seed.  Out: dqkv [B*T, 3d] bf16 (every row written), dpl [pl_rows, d] bf16 (rows >= 2T-1 zero),
 * dbias_u / dbias_v [H*dk] f32.  Scratch: dBand [H, B*T, Rs] bf16 (the band-skewed score gradient on the absolute
 * relative-position axis, Rs from _dims), QvHM [H, B*T, 64] bf16 (q + pos_bias_v head-major), ws f32 x _ws_elems: the
 * position-projection gradient is the TN GEMM dBand_h^T QvHM_h per head (ia_gemm_tn_bf16), issued by this call. */
int ia_relpos_attention_flash_bwd_dims(int T, int* Rs, int* pad0);
int64_t ia_relpos_attention_flash_bwd_ws_elems(int B, int T, int H, int dk);
int ia_relpos_attention_flash_bwd(const void* qkv, const void* pos_proj, const float* bias_u, const float* bias_v,
                                  const int64_t* lens, const void* ctx, const void* dctx, const float* lse, int B, int T, int H,
                                  int dk, float dropout_p, unsigned seed, void* dqkv, void* dpl, int pl_rows, float* dbias_u,
                                  float* dbias_v, void* dBand, void* QvHM, float* ws, ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Persistent single-layer LSTM: the recurrence of the RNNT prediction network (RNNTDecoder.predict
 * A/modules/rnnt.py:683-792 -> LSTMDropout C/parts/rnn.py:151-235 -> torch.nn.LSTM, gate order i,f,g,o, zero initial
 * state) as ONE launch per direction instead of ~6 launches per time step.
 *   ia_lstm_forward : Gx [U,B,4H] f32 = x W_ih^T + b_ih + b_hh (a GEMM by the caller), Whh [4H,H] bf16
 *                     -> Hout [U,B,H] f32; gates [U,B,4H] f32 (activated) and Cs [U,B,H] f32 are saved for the backward
 *                     (both NULL for inference).
 *   ia_lstm_backward: dHout [U,B,H] f32, saved gates/Cs, WhhT [H,4H] bf16 (W_hh transposed)
 *                     -> dG [U,B,4H] f32 = gradient w.r.t. the gate pre-activations; dW_ih = dG^T x, dW_hh = dG[1:]^T Hout[:-1],
 *                     db = sum dG, dx = dG W_ih are GEMMs/reductions by the caller.
 *   scratch: ia_lstm_scratch_bytes(B,H), 256-byte aligned, caller-owned (hand-off buffers + arrival counter; the
 *            launcher zeroes the counter words -- the first 128 bytes -- on the stream).  After completion scratch word
 *            [1] != 0 means a bounded spin gave up (results invalid) -- the kernels cannot hang.  Word [32] is the same
 *            flag but STICKY: the launcher never clears it, so a caller that keeps one scratch buffer per stream can
 *            poll it once per step (the Python side turns it into RuntimeError, ops/lstm.py).  The spin bound can be
 *            lowered through the environment variable IA_LSTM_SPIN_LIMIT (tests force a timeout with it).
 *   ia_lstm_lds_bytes(H, backward): LDS bytes one workgroup needs (> 160 KiB => IA_UNSUPPORTED from the launchers).
 *   Limits: B <= 32 per call (split larger batches: rows are independent), H % 32 == 0, H/16 workgroups must be
 *           co-resident (H <= 4096).  One workgroup per 16 hidden units keeps its W_hh slice in LDS for the whole sequence.
 */
size_t ia_lstm_scratch_bytes(int B, int H);
int ia_lstm_lds_bytes(int H, int backward);
int ia_lstm_forward(const float* Gx, const void* Whh_bf16, float* Hout, float* gates, float* Cs, int U, int B, int H,
                    void* scratch, size_t scratch_bytes, ia_stream_t stream);
int ia_lstm_backward(const float* dHout, const float* gates, const float* Cs, const void* WhhT_bf16, float* dG, int U,
                     int B, int H, void* scratch, size_t scratch_bytes, ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Feature normalisation + SpecAugment in one pass over the log-mel tensor.
 * Replaces normalize_batch('per_feature') A/parts/preprocessing/features.py:59-76 (a Python loop over the batch),
 * the length masking :458-462 and spec_augment_kernel A/parts/numba/spec_augment/spec_aug_numba.py:26-95.
 *   x, y [B,F,T] f32 (y may alias x); seq_len [B] i64; eps = 1e-5 added to the (unbiased) std.
 *   freq_starts/widths [B,n_freq_masks] i32, time_starts/widths [B,n_time_masks] i32 (device; counts may be 0):
 *   frequency spans mask whole rows, time spans only frames below seq_len[b].  Limit: T <= 4096. */
/* Log-mel front end (FilterbankFeatures.forward A/parts/preprocessing/features.py:400-444) as five launches:
 *   ia_feat_frames   dither (counter-based N(0,1) keyed by (seed,b,sample)) + pre-emphasis (first sample kept, :414) +
 *                    centred, reflect-padded framing: frames [B*Tm, ldf] f32 (columns >= win zero)
 *   ia_gemm_f32      spec = frames @ basis^T   -- windowed real DFT as an exact-fp32 matrix-core GEMM
 *                    (basis [2*half, ldf]: rows k < half: w[n] cos(2 pi k (n+off)/n_fft), rows half+k: the sines)
 *   ia_feat_power    power [M, ldp] = re^2 + im^2 (:424-433), columns >= nbins zero
 *   ia_gemm_f32      mel = power @ fb^T        -- fb [n_mels, ldp] (Slaney filterbank, :327-333,440)
 *   ia_feat_logmel_t out [B, n_mels, Tm] = log(mel + guard) (:444)
 * ia_gemm_f32: C[M,N] = A[M,K] @ W[N,K]^T in exact fp32 (v_mfma_f32_16x16x4_f32); K % 16 == 0. */
int ia_feat_frames(const float* audio, int B, int L, int Tm, int win, int hop, float preemph, float dither, unsigned seed,
                   float* frames, int ldf, ia_stream_t stream);
int ia_gemm_f32(const float* A, int lda, const float* W, int ldw, int M, int N, int K, float* C, int ldc,
                ia_stream_t stream);
int ia_feat_power(const float* spec, int64_t M, int lds, int half, int nbins, float* power, int ldp, ia_stream_t stream);
int ia_feat_logmel_t(const float* mel, int B, int Tm, int F, int ldm, float guard, float* out, ia_stream_t stream);
/* The same features in one pass over the frames (csrc/frontend_fft.hip; n_fft = 512, n_mels <= 128): nothing between the
 * pre-emphasised signal and the log-mel tensor touches HBM.
 *   ia_feat_preemph     y [B,L] = x' - preemph x'(n-1), x' = x + dither randn(seed, b, n)            (:408-414)
 *   ia_feat_logmel_fft  centred reflect-padded frames of y (:415-423 torch.stft), window [win] centred in n_fft, FFT (two
 *                       frames per complex transform; twiddle [n_fft][2] = cos, -sin of 2 pi j / n_fft), |X|^2, the
 *                       filterbank given as chunks of 8 consecutive bins (chunk_start [128], chunk_vals [128][8] zero
 *                       padded, every start + 7 < 272; filt_chunks [n_mels][2] = first chunk, number of chunks: a
 *                       filter's chunks are consecutive and summed in order), log(mel + guard), out [B, n_mels, Tm]. */
int ia_feat_preemph(const float* audio, int B, int L, float preemph, float dither, unsigned seed, float* y,
                    ia_stream_t stream);
int ia_feat_logmel_fft_supported(int n_fft, int win, int n_mels, int n_chunks);
int ia_feat_logmel_fft(const float* y, int B, int L, int Tm, const float* window, int win, int n_fft, int hop,
                       const float* twiddle, const int* chunk_start, const float* chunk_vals, const int* filt_chunks,
                       int n_mels, int n_chunks, float guard, float* out, ia_stream_t stream);
int ia_feat_normalize(const float* x, const int64_t* seq_len, int B, int F, int T, float eps, const int* freq_starts,
                      const int* freq_widths, int n_freq_masks, const int* time_starts, const int* time_widths,
                      int n_time_masks, float mask_value, float* y, ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * CTC loss (ATen nn.CTCLoss semantics as used by A/losses/ctc.py:45-82: blank = num_classes, reduction 'none',
 * zero_infinity): log_probs [B,T,V] f32 batch-major (what ConvASRDecoder.forward returns, conv_asr.py:459-490 -- the
 * reference transposes to [T,B,V] first), targets [B,S] i64 padded, lens i64.
 *   ia_ctc_forward : nll [B] f32 (0 for infeasible alignments under zero_infinity); alpha/beta stay in `workspace`.
 *   ia_ctc_backward: grad [B,T,V] f32 = nll_grad[b] * (exp(lp) - exp(log sum_{s: l'_s = v} alpha_t(s) beta_t(s) + nll - lp)),
 *                    zero for t >= input_lens[b] and for infeasible alignments (nll_grad NULL = ones).
 *   workspace: ia_ctc_workspace_bytes(B,T,S), 256-byte aligned.  Limit: S <= 255 (extended length 2S+1 <= 512). */
size_t ia_ctc_workspace_bytes(int B, int T, int S);
int ia_ctc_forward(const float* log_probs, const int64_t* targets, const int64_t* input_lens, const int64_t* target_lens,
                   int B, int T, int V, int S, int blank, int zero_infinity, float* nll, void* workspace,
                   size_t workspace_bytes, ia_stream_t stream);
int ia_ctc_backward(const float* log_probs, const int64_t* targets, const int64_t* input_lens, const int64_t* target_lens,
                    int B, int T, int V, int S, int blank, const float* nll_grad, float* grad, void* workspace,
                    size_t workspace_bytes, ia_stream_t stream);

/* ia_gemm_tn_bf16: weight gradient of a projection, dW[n,k] = sum_m dY[m,n] X[m,k] (f32, written) and optionally
 * db[n] = sum_m dY[m,n] (NULL to skip): both operands row-major over the contracted frame axis m (bf16, row strides
 * ldy/ldx multiples of 8), tiles transposed on the fly by ds_read_b64_tr_b16, split-K partial tiles in `scratch`
 * (f32 x ia_gemm_tn_scratch_elems) + a finishing sum (one pass when db == dW + n*k).  n, k multiples of 8.  Replaces autograd's x^T @ dy of
 * nn.Linear / pointwise Conv1d in the trainable Conformer blocks. */
/* Grouped form: up to 8 weight gradients in one GEMM launch + one finishing launch (a trainable block's projections:
 * single launches are latency-bound).  n*k and n multiples of 4 per problem (n, k % 8 == 0); db may be NULL. */
typedef struct ia_tn_problem {
    const void* dY; const void* X; float* dW; float* db;
    int ldy, ldx, M, n, k;
} ia_tn_problem;
int64_t ia_gemm_tn_grouped_scratch_elems(const ia_tn_problem* problems, int count);
int ia_gemm_tn_bf16_grouped(const ia_tn_problem* problems, int count, float* scratch, ia_stream_t stream);
int64_t ia_gemm_tn_scratch_elems(int M, int n, int k);
int ia_gemm_tn_bf16(const void* dY, int ldy, const void* X, int ldx, int M, int n, int k, float* dW, float* db,
                    float* scratch, ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row-resident feed-forward module of a Conformer block (csrc/ffn_fused.hip): ONE launch for
 *     x <- [LN2]( x + alpha * dropout_res( dropout_ff(SiLU(LN(x) W1^T + b1)) W2^T + b2 ) )
 * replacing norm_feed_forward{1,2} + ConformerFeedForward.forward + the residual update (+ norm_out after the second
 * module) of ConformerLayer.forward, A/parts/submodules/conformer_modules.py:141-214,385-404.  The [N, d_ff] intermediate
 * stays in LDS (64-frame tiles); the LayerNorm'd frames stay in registers as MFMA operands.
 * x [N, d] f32 in/out (in place); W1 [d_ff, d] bf16, W2 [d, d_ff] bf16 (nn.Linear layout), biases / LayerNorm f32.
 * Dropout: after the activation a mask of the kernel's own (one cheap 32-bit word per frame and 4 hidden units, keyed by
 * seed_ff: this is the forward of no-autograd passes, nothing has to regenerate it); on the module output the mask of
 * ia_gemm_bf16 for (seed_res, row, column of [N, d]); p = 0 disables.  ln2_g/ln2_b NULL: no second LayerNorm.
 * y_out (optional, [N, d] bf16): a bf16 copy of the result.  ln2_to_y_only != 0: x receives the un-normalised residual and
 * LN2 goes to y_out only (the LayerNorm in front of the NEXT module, e.g. norm_self_att after the first feed-forward).  Limits: ia_ffn_fused_supported(d, d_ff) (d = 256,
 * d_ff % 128 == 0); IA_UNSUPPORTED otherwise. */
int ia_ffn_fused_supported(int d, int d_ff);
int ia_ffn_fused(float* x, int N, int d, int d_ff, const float* ln_g, const float* ln_b, float eps, const void* W1,
                 const float* b1, const void* W2, const float* b2, float alpha, float p_ff, unsigned seed_ff, float p_res,
                 unsigned seed_res, const float* ln2_g, const float* ln2_b, void* y_out, int ln2_to_y_only,
                 ia_stream_t stream);
/* ... followed, in the same launch, by a projection of the LN2 rows (the q|k|v projection behind feed_forward1 + norm_self_att of a
 * frozen block, multi_head_attention.py:77-96): t_out [N, nt] bf16 = LN2(x) Wt^T + bt, Wt [nt, d] bf16 row-major, nt % 64 == 0,
 * bt [nt] f32 or NULL; ln2_g / ln2_b required, y_out may be NULL. */
int ia_ffn_fused_tail_supported(int d, int d_ff, int nt);
int ia_ffn_fused_tail(float* x, int N, int d, int d_ff, const float* ln_g, const float* ln_b, float eps, const void* W1,
                      const float* b1, const void* W2, const float* b2, float alpha, float p_ff, unsigned seed_ff, float p_res,
                      unsigned seed_res, const float* ln2_g, const float* ln2_b, void* y_out, int ln2_to_y_only, const void* Wt,
                      const float* bt, void* t_out, int nt, ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Native executor of the no-autograd Conformer prefix (frozen blocks / teacher / eval): one call enqueues the 14
 * kernels of each of `n_layers` blocks (ConformerLayer.forward, conformer_modules.py:141-214) on `stream`.
 * x [B*T, d] f32 residual stream, updated in place to the last block's norm_out; pos_emb [pos_rows >= 2T-1, d] bf16;
 * block l uses dropout seeds seed_base + l*seed_stride + {1..7} (the sites of the Python path); training = 0 disables
 * dropout and uses the BatchNorm running statistics.  All pointers in ia_block_params are device pointers: bf16 weights
 * [out, in] (w_qkv = q|k|v rows concatenated, b_qkv likewise), f32 everything else.  Workspace from
 * ia_conformer_prefix_ws_bytes.  Same limits as the kernels it sequences (head dim <= 64, taps <= 31). */
typedef struct ia_block_params {
    const void *w_ff1a, *w_ff1b, *w_qkv, *w_pos, *w_out, *w_pw1, *w_pw2, *w_ff2a, *w_ff2b;
    const float *b_ff1a, *b_ff1b, *b_qkv, *b_out, *b_pw1, *b_pw2, *b_ff2a, *b_ff2b;
    const float *ln_ff1_g, *ln_ff1_b, *ln_att_g, *ln_att_b, *ln_conv_g, *ln_conv_b, *ln_ff2_g, *ln_ff2_b, *ln_out_g, *ln_out_b;
    const float *pos_u, *pos_v, *dw_w, *dw_b, *bn_g, *bn_b;
    float *bn_rm, *bn_rv;
    int64_t* bn_nbt;
    float ln_eps, bn_eps, bn_momentum, p_drop, p_ff, p_att, fc_factor;
    int d, d_ff, n_heads, ksz;
    const void* pl_cached;   /* optional: linear_pos(pos_emb) [pos_rows, d] bf16 computed earlier (frozen weights: it only
                                depends on T) -- the executor then skips that GEMM; NULL: computed per call */
    const void* w_pw1_glu;   /* optional (d % 64 == 0): pointwise_conv1 weight / bias with the rows regrouped so that every 128 */
    const float* b_pw1_glu;  /* consecutive outputs are 64 value channels followed by THEIR 64 gate channels (rows 128t+j = value
                                channel 64t+j, rows 128t+64+j = gate channel d+64t+j): the prefix executor then applies the GLU in
                                the GEMM epilogue (act 4) and the depthwise conv reads the gated [N,d] bf16 tensor; NULL: off */
} ia_block_params;
size_t ia_conformer_prefix_ws_bytes(int B, int T, int d, int d_ff, int H, int ksz, int pos_rows);
int ia_conformer_prefix_fwd(const ia_block_params* layers, int n_layers, float* x, const void* pos_emb, int pos_rows,
                            const int64_t* lens, int B, int T, unsigned seed_base, unsigned seed_stride, int training,
                            void* workspace, size_t workspace_bytes, ia_stream_t stream);
/* The same prefix in segments of half blocks, for SyncBatchNorm over several ranks: unit 2k = block k up to and including
 * the BatchNorm sums, unit 2k+1 = BatchNorm + SiLU onwards; runs units [seg_begin, seg_end).  Between the two halves of a
 * block the caller all-reduces [sum | sumsq | count] (2d+1 floats at workspace + ia_conformer_prefix_ws_sums_offset) and
 * calls ia_bn_sync_finish; bn_synced = 1 then keeps ia_bn_silu from touching the running statistics. */
size_t ia_conformer_prefix_ws_sums_offset(int B, int T, int d, int d_ff, int H, int ksz, int pos_rows);
int ia_conformer_prefix_fwd_seg(const ia_block_params* layers, int n_layers, float* x, const void* pos_emb, int pos_rows,
                                const int64_t* lens, int B, int T, unsigned seed_base, unsigned seed_stride, int training,
                                int seg_begin, int seg_end, int bn_synced, void* workspace, size_t workspace_bytes,
                                ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Native executors of ONE TRAINABLE Conformer block (csrc/block_train.hip): ConformerLayer.forward
 * (conformer_modules.py:141-214) keeping what its backward needs, and that backward as two calls around the attention
 * core's backward (ia_relpos_attention_flash_bwd + the caller's position-gradient GEMMs).  Replaces the per-op Python autograd node:
 * one C call enqueues the ~25 (forward) / ~35 + ~25 (backward) kernels of a block on `stream`.
 *   ia_block_saved  activations kept from forward to backward (caller-owned; bf16 unless noted): y1,y2,y3,y4 [N,d] =
 *                   LayerNorm outputs; h1p,h1,h4p,h4 [N,d_ff] = feed-forward pre-activation / dropout(SiLU(.));
 *                   x1..x4 [N,d] f32 = residual stream after each module; qkv [N,3d]; pl [pos_rows,d]; ctxv [N,d];
 *                   c2 [N,2d] (pointwise_conv1 output); z [N,d] f32 (depthwise conv output); sums [2,d] f32 (BatchNorm
 *                   sum / sum of squares); c3 [N,d]; lse [B*H,T] f32 (attention log-sum-exp).
 *   ia_block_grads  where the parameter gradients are WRITTEN (f32, caller-owned; the q|k|v weight / bias gradients are
 *                   one [3d,d] / [3d] block in that order).
 *   forward   x0 [N,d] f32 -> out [N,d] f32 (= norm_out(...)); train-mode BatchNorm (running statistics updated); dropout
 *             sites seed + {1..7} as in ia_conformer_prefix_fwd; vt_scratch = ia_attn_vt_elems bf16, dw_scratch =
 *             ia_dwconv_scratch_elems f32.  Limits: ia_conformer_block_supported (head dim 64, taps <= 31; any T: the
 *             attention core is the key-tiled ia_relpos_attention_flash_lse, vt_scratch is unused).
 *   bwd_a     dout [N,d] f32 -> gradients of norm_out, feed_forward2, conv module, linear_out; *dx2_out (f32 [N,d]) and
 *             *dctx_out (bf16 [N,d]) point INTO the workspace: d(residual in front of the attention branch), d(ctx).
 *   bwd_b     dqkv [N,3d], dpl [pos_rows,d] (bf16, from the attention backward) -> remaining gradients, dx0 [N,d] f32,
 *             then one launch adds the parameter gradients to their .grad buffers: add_table = n_add rows
 *             {float* dst, const float* src, int64 n} in device memory (NULL: skip).
 *   The workspace (ia_conformer_block_bwd_ws_bytes) must stay untouched between bwd_a and bwd_b. */
typedef struct ia_block_saved {
    void *y1, *h1p, *h1; float* x1;
    void *y2, *qkv, *pl, *ctxv; float* x2;
    void *y3, *c2; float *z, *sums; void* c3; float* x3;
    void *y4, *h4p, *h4; float* x4;
    float* lse;   /* [B*n_heads, T] f32: the attention core's per-query log-sum-exp (ia_relpos_attention_flash_lse) */
} ia_block_saved;
typedef struct ia_block_grads {
    float *w_ff1a, *b_ff1a, *w_ff1b, *b_ff1b, *w_qkv, *b_qkv, *w_pos, *w_out, *b_out, *w_pw1, *b_pw1, *w_pw2, *b_pw2;
    float *w_ff2a, *b_ff2a, *w_ff2b, *b_ff2b;
    float *ln_ff1_g, *ln_ff1_b, *ln_att_g, *ln_att_b, *ln_conv_g, *ln_conv_b, *ln_ff2_g, *ln_ff2_b, *ln_out_g, *ln_out_b;
    float *dw_w, *dw_b, *bn_g, *bn_b;
} ia_block_grads;
struct ia_block_params;
int ia_conformer_block_supported(int d, int d_ff, int H, int ksz, int T);
size_t ia_conformer_block_bwd_ws_bytes(int B, int T, int d, int d_ff, int ksz);
int ia_conformer_block_fwd(const struct ia_block_params* layer, const float* x0, const void* pos_emb, int pos_rows,
                           const int64_t* lens, int B, int T, unsigned seed, const ia_block_saved* saved, float* out,
                           void* vt_scratch, float* dw_scratch, ia_stream_t stream);
int ia_conformer_block_bwd_a(const struct ia_block_params* layer, const ia_block_saved* saved, const ia_block_grads* grads,
                             const float* dout, const int64_t* lens, int B, int T, unsigned seed, void* workspace,
                             size_t workspace_bytes, float** dx2_out, void** dctx_out, ia_stream_t stream);
/* SyncBatchNorm over several ranks: the same calls split at the BatchNorm exchanges (phase 0 = whole call; forward: 1 = up
 * to the BatchNorm sums, 2 = BatchNorm + SiLU onwards; backward part 1: 1 = up to the local S1 | S2 in grads->bn_b / bn_g,
 * 2 = from dz onwards with bn_S12 = the all-reduced sums rescaled by n_local / n_global). */
int ia_conformer_block_fwd_phase(const struct ia_block_params* layer, const float* x0, const void* pos_emb, int pos_rows,
                                 const int64_t* lens, int B, int T, unsigned seed, const ia_block_saved* saved, float* out,
                                 void* vt_scratch, float* dw_scratch, int phase, ia_stream_t stream);
int ia_conformer_block_bwd_a_phase(const struct ia_block_params* layer, const ia_block_saved* saved, const ia_block_grads* grads,
                                   const float* dout, const int64_t* lens, int B, int T, unsigned seed, void* workspace,
                                   size_t workspace_bytes, float** dx2_out, void** dctx_out, int phase, const float* bn_S12,
                                   ia_stream_t stream);
int ia_conformer_block_bwd_b(const struct ia_block_params* layer, const ia_block_saved* saved, const ia_block_grads* grads,
                             const float* x0, const void* pos_emb, int pos_rows, const void* dqkv, const void* dpl, int B,
                             int T, unsigned seed, void* workspace, size_t workspace_bytes, float* dx0,
                             const void* add_table, int n_add, ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backward-side kernels of the trainable Conformer blocks (autograd of ConformerLayer.forward,
 * A/parts/submodules/conformer_modules.py:141-214); the dense contractions in between are bf16 GEMMs.
 *   ia_layernorm_bwd      dx_out = (dx_in or 0) + dLN/dx; dgamma/dbeta written (block partials in `scratch`, f32 x
 *                         ia_layernorm_bwd_scratch_elems, then a column-sum pass).  dy as f32 OR bf16.
 *   ia_silu_dropout       out = dropout(SiLU(h_pre)) (bf16 [M,N]); mask = the GEMM epilogue's counter mask (seed,row,col/8)
 *   ia_silu_dropout_bwd   out = dh * keep*scale * SiLU'(h_pre)
 *   ia_scale_dropout_bf16 out = bf16(alpha * keep*scale * dy)   -- gradient entering a residual branch
 *   ia_bn_silu_bwd        SiLU' and train-mode BatchNorm backward from the forward's per-channel sums: S1 = dbeta,
 *                         S2 = dgamma (both written: block partial rows in `scratch`, f32 x ia_bn_silu_bwd_scratch_elems),
 *                         dz [n_rows,d] f32
 *   ia_glu_mask / ia_glu_bwd   G = mask(GLU(c2)) f32;  dc2 bf16 [rows,2d] from dG
 *   ia_attn_keepmask      attention-dropout keep mask of ia_relpos_attention as a bf16 [B,H,T,T] tensor (0 or 1/(1-p))
 */
int ia_layernorm_bwd(const float* x, int ldx, const float* dy_f32, const void* dy_bf16, int ldy, int N, int d,
                     const float* gamma, float eps, const float* dx_in, float* dx_out, int lddx, float* dgamma,
                     float* dbeta, float* scratch, ia_stream_t stream);
/* ... and, in the same pass, dx_bf16 = bf16(alpha * keep*scale * dx_out): the ia_scale_dropout_bf16 of the result (the operand of
 * the data- and weight-gradient GEMMs of the residual branch in front of this LayerNorm); d % 8 == 0. */
int ia_layernorm_bwd_drop(const float* x, int ldx, const float* dy_f32, const void* dy_bf16, int ldy, int N, int d,
                          const float* gamma, float eps, const float* dx_in, float* dx_out, int lddx, float* dgamma,
                          float* dbeta, float alpha, float dropout_p, unsigned seed, void* dx_bf16, int lddxh, float* scratch,
                          ia_stream_t stream);
/* Deferred column sums: ia_layernorm_bwd(_drop) with dgamma = dbeta = NULL leaves its ia_layernorm_bwd_partial_rows(N) partial
 * rows [rows][2d] (d gamma | d beta) in `scratch`; ia_partials_finish_multi adds up to eight such sets in ONE launch:
 * out0[c] = sum_g part[g*C + c] for c < C0, out1[c - C0] for the rest (a trainable block's five LayerNorm gradients: one
 * finishing launch instead of five). */
typedef struct ia_finish_job { const float* part; int G, C, C0; float* out0; float* out1; } ia_finish_job;
int ia_layernorm_bwd_partial_rows(int N);
int ia_partials_finish_multi(const ia_finish_job* jobs, int count, ia_stream_t stream);
int64_t ia_layernorm_bwd_scratch_elems(int N, int d);
int ia_silu_dropout(const void* h_pre, int64_t M, int N, float dropout_p, unsigned seed, void* out, ia_stream_t stream);
int ia_silu_dropout_bwd(const void* h_pre, const void* dh, int64_t M, int N, float dropout_p, unsigned seed, void* out,
                        ia_stream_t stream);
int ia_scale_dropout_bf16(const float* dy, int64_t M, int N, float alpha, float dropout_p, unsigned seed, void* out,
                          ia_stream_t stream);
/* dst [M, ldd] bf16 = src [M, N] f32 (row stride lds), columns N..ldd zero (ldd % 8 == 0): ragged-width gradients -> GEMM operand */
int ia_cast_pad_bf16(const float* src, int lds, int64_t M, int N, void* dst, int ldd, ia_stream_t stream);
int ia_bn_silu_bwd(const float* z, const void* dc3, int64_t n_rows, int d, const float* bn_sum, const float* bn_sumsq,
                   const float* gamma, const float* beta, float eps, float* S1, float* S2, float* dz, float* scratch,
                   ia_stream_t stream);
int64_t ia_bn_silu_bwd_scratch_elems(int64_t n_rows, int d);
/* SyncBatchNorm (torch.nn.SyncBatchNorm.convert_sync_batchnorm, R/cl_baseline.py:133; conformer_modules.py:322): the
 * per-channel sums of ia_glu_dwconv, extended by the rank's row count ([sum(d) | sumsq(d) | count]), are all-reduced by the
 * caller (torch.distributed / RCCL); ia_bn_sync_finish then updates the running statistics from the GLOBAL batch and
 * rescales the sums by n_local / count so that ia_bn_silu / ia_bn_silu_bwd (which form mean and rstd as sums / n_local)
 * see the global statistics.  The backward is split at its own exchange: ia_bn_silu_bwd_reduce leaves the LOCAL S1 | S2
 * (= d beta | d gamma of this rank), the caller all-reduces a copy and rescales it by n_local / n_global,
 * ia_bn_silu_bwd_apply forms dz from it.  (ia_bn_silu_bwd = reduce + apply on one rank.) */
int ia_bn_sync_finish(float* sums_and_count, int d, int64_t n_local_rows, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float momentum, ia_stream_t stream);
int ia_bn_silu_bwd_reduce(const float* z, const void* dc3, int64_t n_rows, int d, const float* bn_sum, const float* bn_sumsq,
                          const float* gamma, const float* beta, float eps, float* S1, float* S2, float* scratch,
                          ia_stream_t stream);
int ia_bn_silu_bwd_apply(const float* z, const void* dc3, int64_t n_rows, int d, const float* bn_sum, const float* bn_sumsq,
                         const float* gamma, const float* beta, float eps, const float* S1, const float* S2, float* dz,
                         ia_stream_t stream);
int ia_glu_mask(const void* c2, const int64_t* lens, int B, int T, int d, float* G, ia_stream_t stream);
int ia_glu_bwd(const void* c2, const float* dG, const int64_t* lens, int B, int T, int d, void* dc2, ia_stream_t stream);
int ia_attn_keepmask(int B, int H, int T, float dropout_p, unsigned seed, void* mask_bf16, ia_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Continual-learning regularisers and optimizer over ONE flat fp32 buffer holding every trainable parameter
 * (tensor k occupies [off_k, off_k + numel_k), off_k a multiple of 64 floats, gaps zero-filled).
 *
 * chunk_table: int32[nchunks][4] = {offset (multiple of 4), count (<= ia_cl_chunk_elems()), segment id, 0};
 *              chunks never straddle two tensors.  seg_inv_numel[k] = 1/numel_k.
 *
 * ia_cl_penalty        replaces get_penalty_grads R/cl_baseline_ewc.py:69-81 (+ set_grads R/utils.py:316-321) and
 *                      penalty R/cl_baseline_mas.py:70-75:
 *                         g = coef * weight * (theta - theta_star)      (EWC: coef = 2*e_lambda, weight = Fisher;
 *                                                                        MAS: coef = 2*mas_lambda, weight = omega)
 *                         grad = g (accumulate = 0: "pre-load .grad") or grad += g (accumulate = 1); grad may be NULL
 *                         seg_abs_mean[k] += mean_k |g|   (caller zeroes; NULL to skip; EWC monitor 'ewc_penalty')
 *                         penalty_sum[0]  += sum weight*(theta-theta_star)^2  (caller zeroes; NULL to skip; MAS 'mass_loss')
 * ia_cl_fisher_accumulate  R/cl_baseline_ewc.py:245-255:  fisher += loss_scalar[0] * grad^2  (scalar read on device)
 * ia_cl_abs_accumulate     R/cl_baseline_mas.py:267-270:  omega  += |grad|
 * ia_adamw_step            torch.optim.AdamW update (R/cl_baseline.py:137; lr 1e-4, betas .9/.999, eps 1e-8, wd 1e-2),
 *                          `step` counts from 1; grad is multiplied by grad_scale first (1.0, or 1/world for DP mean);
 *                          shadow_bf16 (optional, n x bf16) receives the updated weights rounded to bf16.
 * ia_adamw_step_segmented  the same update per TENSOR, with torch.optim.AdamW's treatment of `p.grad is None`
 *                          (optimizer.zero_grad() sets grads to None, R/cl_baseline.py:187: a tensor that received no
 *                          gradient -- the other 21 language heads, heads of finished tasks -- gets no weight decay, no
 *                          moment decay and keeps its own step counter).  seg_active[nseg] i32 (zero on entry, zero on
 *                          exit), seg_step[nseg] i32 per-tensor step counters (advanced here); a tensor is live when any
 *                          bit of its gradient is set, or always when all_active != 0 (a penalty was pre-loaded into
 *                          every .grad: set_grads R/utils.py:316-321).  Three launches on `stream`.
 */
int ia_cl_chunk_elems(void);
int ia_cl_penalty(const float* theta, const float* theta_star, const float* weight, float coef, float* grad,
                  int accumulate, const int32_t* chunk_table, int nchunks, const float* seg_inv_numel,
                  float* seg_abs_mean, float* penalty_sum, ia_stream_t stream);
int ia_cl_fisher_accumulate(float* fisher, const float* grad, const float* loss_scalar, int64_t n, ia_stream_t stream);
int ia_cl_abs_accumulate(float* omega, const float* grad, int64_t n, ia_stream_t stream);
int ia_adamw_step(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale, void* shadow_bf16,
                  ia_stream_t stream);
int ia_adamw_step_segmented(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, const int32_t* chunk_table,
                            int nchunks, int32_t* seg_active, int32_t* seg_step, int nseg, int all_active, float lr,
                            float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* shadow_bf16,
                            ia_stream_t stream);

/* ---- CTC head + loss on RAW logits (ConvASRDecoder.forward + CTCLoss.forward, A/modules/conv_asr.py:459-490 and
 * A/losses/ctc.py:68-82, without the [B,T,V] log-prob tensor and without a softmax backward pass): logits [B*T, ld] f32 with V
 * valid columns (the head GEMM's padded output), lse [B*T] = per-frame log-sum-exp over the V columns (ia_ctc_row_lse).
 * ia_ctc_backward_logits writes grad_scale * nll_grad[b] * (softmax - occupancy) as bf16 [B*T, ldg] with zero padding columns
 * (the A operand of the head's data / weight gradient GEMMs); same workspace as ia_ctc_forward. */
int ia_ctc_row_lse(const float* logits, int ld, int64_t M, int V, float* lse, ia_stream_t stream);
int ia_ctc_forward_logits(const float* logits, int ld, const float* lse, const int64_t* targets, const int64_t* input_lens,
                          const int64_t* target_lens, int B, int T, int V, int S, int blank, int zero_infinity, float* nll,
                          void* workspace, size_t workspace_bytes, ia_stream_t stream);
int ia_ctc_backward_logits(const float* logits, int ld, const float* lse, const int64_t* targets, const int64_t* input_lens,
                           const int64_t* target_lens, int B, int T, int V, int S, int blank, const float* nll_grad,
                           float grad_scale, void* grad_bf16, int ldg, void* workspace, size_t workspace_bytes, ia_stream_t stream);

/* ---- Glue of the step's tail (csrc/tail_ops.hip): each replaces a handful of small ATen launches.
 * ia_select_rows_cast  rows [row0, row0+nrows) (+ extra_row if >= 0) of W [*, K] f32 (row stride ldw), times `scale`, as a 16-bit
 *                      operand out [rows_out, K] (bf16, or f16 when out_f16; rows beyond the selection zero); optionally the
 *                      transpose outT [K, ldt] (ldt >= rows_out, padding zero) and the selected bias entries bias_out [rows_out]:
 *                      the language's 256 + blank rows of the 5633-wide CTC head (conv_asr.py:469-480) / the language head of the
 *                      joint with the dropout scale folded in (rnnt.py:1632-1640).
 * ia_rows_scatter_add  dst[row] += scale * src[i] over the same row selection (+ bias): the backward of that selection.
 * ia_loss_combine      out4 = [mean(costs), mean(nll), (1-w) mean(costs) + w mean(nll), #non-zero flag words]; `total` (optional,
 *                      one float) receives out4[2] as well (the autograd-visible scalar)
 *                      (hybrid_rnnt_ctc_models.py:899-913; flags: the persistent LSTM's sticky timeout words, may be NULL).
 * ia_loss_combine_bwd  g_costs[b] = gout (1-w)/B, g_nll[b] = gout w/B  (gout NULL = 1).
 * ia_embed_sos         prediction-network input (rnnt.py:734-751): out[(b,0)] = 0, out[(b,u)] = E[tokens[b,u-1]]; time_major
 *                      selects [U+1,B,H] (LSTM operand) or [B,U+1,H]; f32 or bf16.
 * ia_embed_sos_bwd     dE[row] += scale * sum of the dX (f32, or bf16 when dx_bf16) rows that read it (tokens scanned in order: deterministic; rows >=
 *                      n_rows never referenced; pad_row skipped as nn.Embedding(padding_idx) does).
 * ia_multi_axpy        dst_i += scale_i * src_i for a HOST array of ia_axpy_row (device pointers inside; the rows travel as kernel
 *                      arguments, 24 per launch): AccumulateGrad of several parameters in one launch. */
typedef struct ia_axpy_row {
    float* dst; const float* src; long long n; float scale; int pad;
} ia_axpy_row;
/* ia_transpose16_multi  out_i [cols, rows] = in_i [rows, cols]^T for up to 8 matrices of 16-bit elements (bf16 / f16; rows, cols
 *                      multiples of 8) in ONE launch: the weight images W^T the data-gradient GEMMs of the tail read.
 * ia_swap01_cast       out[j][i][:] = in[i][j][:] for in [n0, n1, H] f32 -> out [n1, n0, H] (bf16 when out_bf16 else f32), or
 *                      with in_bf16 a bf16 input: the prediction network's time-major <-> the joint's batch-major layout. */
typedef struct ia_tr_job { const void* in; void* out; int rows, cols; } ia_tr_job;
int ia_transpose16_multi(const ia_tr_job* jobs_host, int njobs, ia_stream_t stream);
int ia_swap01_cast(const void* in, int in_bf16, int n0, int n1, int H, void* out, int out_bf16, ia_stream_t stream);
int ia_select_rows_cast(const float* W, int ldw, const float* bias, int row0, int nrows, int extra_row, int K, int rows_out,
                        float scale, int out_f16, void* out, void* outT, int ldt, float* bias_out, ia_stream_t stream);
int ia_rows_scatter_add(float* dst, int ldd, const float* src, int lds, int row0, int nrows, int extra_row, int K, float scale,
                        float* bias_dst, const float* bias_src, ia_stream_t stream);
int ia_loss_combine(const float* costs, const float* nll, int B, float ctc_weight, const int* flag0, const int* flag1,
                    const int* flag2, const int* flag3, float* out4, float* total, ia_stream_t stream);
int ia_loss_combine_bwd(const float* gout, int B, float ctc_weight, float* g_costs, float* g_nll, ia_stream_t stream);
int ia_embed_sos(const float* E, const int64_t* tokens, int B, int U, int H, int n_rows, int time_major, int out_bf16, void* out,
                 ia_stream_t stream);
int ia_embed_sos_bwd(const void* dX, int dx_bf16, const int64_t* tokens, int B, int U, int H, int n_rows, int pad_row,
                     int time_major, float scale, float* dE, ia_stream_t stream);
int ia_multi_axpy(const ia_axpy_row* rows_host, int nrows, int max_blocks_per_row, ia_stream_t stream);

/* ---- Peaks measured on the box ("peaks measured on the box with a streaming-copy and an MFMA
 * microbenchmark"); no reference counterpart -- bench.py prices its roofline fractions against these and the datasheet.
 * ia_peak_stream_copy   16-byte-per-lane copy src -> dst (bytes % 16 == 0): 2 x bytes of HBM traffic per launch.
 * ia_peak_mfma_bf16     `workgroups` x 4 waves, each `iters` x 8 back-to-back v_mfma_f32_16x16x32_bf16 on register operands
 *                       (non-trivial values); sink: >= workgroups * 256 floats (never written in practice).
 * ia_peak_mfma_bf16_flops  FLOPs one such launch performs (pure host function).
 */
int ia_peak_stream_copy(const void* src, void* dst, size_t bytes, ia_stream_t stream);
int ia_peak_mfma_bf16(float* sink, int workgroups, int iters, ia_stream_t stream);
double ia_peak_mfma_bf16_flops(int workgroups, int iters);

#ifdef __cplusplus
}
#endif
#endif /* INDICASR_H */
