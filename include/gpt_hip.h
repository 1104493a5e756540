/* gpt_hip.h - C ABI of the MI355X (gfx950) G.pt denoiser hot path.
 *
 * The reference (wpeebles/G.pt) has no FFI: its hot path sits behind a Python call boundary
 * (SURVEY.md section 8b).  Each entry point below replaces the ATen dispatches of the cited
 * reference lines; the Python host in g.pt_amd/ binds them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; sizes are element counts;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - `dtype` selects the activation/weight type of the MFMA contractions:
 *        GPT_F32  = exact fp32 (v_mfma_f32_32x32x2_f32), the parity mode,
 *        GPT_BF16 = bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16);
 *     residual stream, LayerNorm/softmax statistics, biases and all DDPM arithmetic are fp32 in both;
 *   - return value: 0 = ok; negative = rejected before any launch (bad shape / alignment / enum);
 *     positive = a hipError_t from the launch.  gpt_last_error() gives the message.
 *   - no entry point allocates, frees or synchronises; all are stream-ordered and graph-capturable.
 */
#ifndef GPT_HIP_H
#define GPT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPT_F32 0
#define GPT_BF16 1

/* epilogues of gpt_gemm_nt */
#define GPT_EPI_STORE 0     /* C(dtype)  = acc + bias                                   */
#define GPT_EPI_GELU 1      /* C(dtype)  = gelu_erf(acc + bias)          nn.GELU()      */
#define GPT_EPI_RESADD 2    /* C(fp32)   = R + acc + bias, R = aux (fp32, ldaux) if aux != NULL else C   x = x + f(x) */
#define GPT_EPI_STORE_F32 3 /* C(fp32)   = acc + bias                                   */
#define GPT_EPI_GELU_AUX 4  /* aux(dtype) = acc + bias ; C(dtype) = gelu(aux)  (training: keeps pre-activation) */
#define GPT_EPI_DGELU 5     /* C(dtype)  = acc * gelu'(aux)              (backward of GELU) */
#define GPT_EPI_ACCUM_F32 6 /* C(fp32)  += acc                           (gradient accumulation) */

const char* gpt_last_error(void);
int gpt_abi_version(void);

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]).  A, W: `dtype`, row-major, leading dims lda/ldw in
 * elements, K % 64 == 0 (bf16) or K % 32 == 0 (fp32), 16-byte aligned rows; N % 8 == 0, ldc % 4 == 0, C/aux/bias
 * 16-byte aligned (vector epilogue).  bias: fp32 or NULL.  GELU: fp32 mode uses libm erff (parity mode); bf16 mode uses
 * x * sigmoid(a x + b x^3 + c x^5) fitted to the exact erf GELU, |error| <= 2.6e-5 absolute (below the bf16 rounding of the
 * output; csrc/gemm_common.h gelu_fast2), and GPT_EPI_DGELU multiplies by the derivative of that same fit, |error| <= 1.1e-4
 * absolute against Phi(x) + x phi(x) (gelu_fast_grad2).
 * Replaces nn.Linear on the path: Gpt/models/transformer.py:38-40 (q,k,v), :50 (proj), :63-65 (mlp). */
int gpt_gemm_nt(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                void* C, int64_t ldc, void* aux, int64_t ldaux, int M, int N, int K, int epilogue,
                void* stream);

/* Optional scratch for gpt_gemm_nt (bf16): with a device buffer registered, problems of few 256 x 256 output tiles and a
 * long K (small batches) are contracted in K spans - fp32 partial tiles into this buffer, then one streaming pass that
 * sums them and applies the epilogue - instead of leaving most CUs idle.  The buffer stays the caller's (the library
 * never allocates); it must outlive the calls, and calls that use it must not overlap on different streams.  The
 * registration is per device: it applies to the device that is current at the call.
 * 64 MiB covers the G.pt shapes.  (NULL, 0) unregisters.  No counterpart in the reference (an ATen/cuBLAS internal). */
int gpt_gemm_set_workspace(void* ptr, size_t bytes);

/* gpt_gemm_nt for callers that cannot allocate device memory or hand over a workspace (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same kernels and -
 * against gpt_gemm_nt with a registered workspace of GPT_GEMM_STATIC_SCRATCH_BYTES - bit-identical results; the fp32 partial
 * tiles of the K-span path go to a scratch array that is part of the library's code object (one copy per GPU, held until the
 * process ends).  Calls no hipMalloc / hipFree or any allocating routine.  Limit: M * N * 8 <= GPT_GEMM_STATIC_SCRATCH_BYTES
 * (larger problems are rejected with a negative status: they belong to gpt_gemm_nt, whose single-pass kernels need no
 * scratch at that size).  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different
 * streams, different threads) race on it; nothing in it has to be cleared between calls (every partial tile that is read
 * was written by the same call). */
#define GPT_GEMM_STATIC_SCRATCH_BYTES (64u << 20)
int gpt_gemm_nt_static(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       void* C, int64_t ldc, void* aux, int64_t ldaux, int M, int N, int K, int epilogue,
                       void* stream);

/* The residual Linear of a Block together with the LayerNorm that reads its result (Gpt/models/transformer.py:70-71:
 * x = x + attn(ln1(x)); then ln2(x) - and x = x + mlp(ln2(x)); then the next block's ln1 or ln_f, :311):
 *   x[M,N] (fp32, dense: ldx == N) = resid[M,N] (fp32, leading dim ldr; may be x itself) + A[M,K] . W[N,K]^T + bias
 *   y[M,N] (dtype)                 = LayerNorm(x; gamma, beta, eps);   mean / rstd (fp32 per row) optional
 * Exactly gpt_gemm_nt(..., GPT_EPI_RESADD) followed by gpt_layernorm, and for large problems run as those two kernels.
 * Small bf16 problems (few output tiles) with a workspace registered are contracted in K spans, and the pass that sums the
 * spans is the LayerNorm: one kernel less per residual Linear and no re-read of x.  N must be 768, 1024, 1536, 2048 or
 * 3072 for that path; other shapes take the two-kernel route. */
int gpt_gemm_resadd_layernorm(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                              const float* resid, int64_t ldr, float* x, int64_t ldx, const float* gamma,
                              const float* beta, void* y, float* mean, float* rstd, int M, int N, int K, float eps,
                              void* stream);

/* C[I, J] (+)= P[R, I]^T . Q[R, J]: bf16 operands contracted over their ROW index, fp32 output (accumulate != 0:
 * C += ...).  The weight gradient dW = dY^T . X that autograd derives for every nn.Linear of the trunk
 * (Gpt/models/transformer.py:38-40, 50, 63-65), reading dY [B*T, n_out] and X [B*T, k_in] as they lie in memory
 * (transposing LDS reads, ds_read_b64_tr_b16).  I % 256 == 0, J % 256 == 0, any R > 0 (a ragged last 64-row K-tile is
 * zero-filled); ldp, ldq % 8 == 0. */
int gpt_gemm_tn(const void* P, int64_t ldp, const void* Q, int64_t ldq, float* C, int64_t ldc, int I, int J, int R,
                int accumulate, void* stream);

/* n <= 4 independent gpt_gemm_tn problems with the same contraction length R in ONE launch (arrays of n entries): the
 * four weight gradients of a Block (query|key|value, proj, mlp.0, mlp.2; the backward of transformer.py:69-72) have too few
 * 256 x 256 tiles each to fill 256 CUs but fill them together.  No counterpart in the reference (autograd launches one
 * cuBLAS GEMM per nn.Linear). */
int gpt_gemm_tn_grouped(int n, const void* const* P, const int64_t* ldp, const void* const* Q, const int64_t* ldq,
                        float* const* C, const int64_t* ldc, const int* I, const int* J, int R, int accumulate,
                        void* stream);

/* Measurement aid (bench.py roofline pass; no counterpart in the reference): while enabled, every gpt_gemm_nt
 * launch is bracketed by two HIP events recorded on its own stream (at most 4096 launches per enable);
 * collect waits for them and returns the summed kernel time, the summed algorithmic FLOPs (2 M N K) and the
 * launch count.  Not for use under graph capture. */
int gpt_gemm_prof_enable(int on);
int gpt_gemm_prof_collect(double* total_ms, double* total_flops, int* launches);

/* y[rows, d](out_dtype) = LayerNorm(x[rows, d] fp32; gamma, beta fp32, eps) (+ add[(row % add_rows), d] fp32).
 * out_dtype: GPT_F32 or GPT_BF16.  Optionally writes mean/rstd (fp32, per row) for the backward.
 * Replaces nn.LayerNorm at transformer.py:70-71 (ln1, ln2), :226 (ln_in, fused with the pos_emb add of
 * :308-309 through add/add_rows), :311 (ln_f). */
int gpt_layernorm(const float* x, const float* gamma, const float* beta, const float* add, int add_rows,
                  void* y, int out_dtype, float* mean, float* rstd, int rows, int d, float eps, void* stream);

/* Non-causal multi-head attention over T <= 32 tokens.  qkv: [B*T, 3*d] (`dtype`), columns
 * [0,d) = query, [d,2d) = key, [2d,3d) = value, head h at columns h*hd..; y: [B*T, d] (`dtype`).
 * hd = d / n_head must be a multiple of 32 and <= 128.  Replaces transformer.py:43-47. */
int gpt_attention(int dtype, const void* qkv, void* y, int B, int T, int n_head, int hd, void* stream);

/* Scalar tokens: out[b, j*W .. (j+1)*W) = [sin(f*s_j), cos(f*s_j), s_j],  W = 2*nfreq+1,
 * j = 0: s = t[b] (int64 timestep); j = 1: s = fp32(lt - lp); j = 2: fp32(lt); j = 3: fp32(lp), where the
 * subtraction is done in float64 when loss_is_f64 (inputs are float64) and in fp32 otherwise, exactly as
 * transformer.py:431 subtracts before the cast of :329.  n_scalar = 2 (delta only) or 4 (absolute).
 * out: fp32 [B, n_scalar*W].  Replaces FrequencyEmbedder.forward (:324-334) x4 and embed_loss (:427-438). */
int gpt_scalar_tokens(const int64_t* t, const void* loss_target, const void* loss_prev, int loss_is_f64,
                      const float* freqs, int nfreq, int n_scalar, float* out, int B, void* stream);

/* Ragged per-token encoder: for token i, out[b, i, :] = src_i[b, 0:k_i] . W_i[d, k_i]^T.
 * Token descriptors are host arrays (copied into kernel arguments, <= 64 tokens): src pointer (fp32,
 * row stride src_ld[i]), weight pointer (`dtype`, row stride w_ld[i] >= k_i), k_i.
 * out: fp32 [B, n_tok, d].  Replaces GPT.encode_parameters' split + 26 Linears + stack
 * (transformer.py:221-225) without materialising the (B,16948) concatenation of :458-459. */
int gpt_encode_tokens(int dtype, int n_tok, const float* const* host_src, const int64_t* host_src_ld,
                      const void* const* host_w, const int64_t* host_w_ld, const int* host_k,
                      float* out, int B, int d, void* stream);

/* The same for n_sel of the n_tok tokens: descriptor i writes out[b, slot[i], :] (slot == NULL: identity).  Inside
 * p_sample_loop only x and t change from step to step (gaussian_diffusion.py:492-505): the tokens of x_prev and of the
 * two losses are encoded once per loop. */
int gpt_encode_tokens_subset(int dtype, int n_sel, const float* const* host_src, const int64_t* host_src_ld,
                             const void* const* host_w, const int64_t* host_w_ld, const int* host_k,
                             const int* host_slot, int n_tok, float* out, int B, int d, void* stream);

/* Ragged per-token decoder + global residual: for j < n_tok,
 * out[b, off_j .. off_j+n_j) = feat[b, j, :] . W_j[n_j, d]^T + resid[b, off_j ..].
 * feat: `dtype` [B, T, d]; out/resid: fp32 [B, D] (resid may be NULL).
 * Replaces GPT.decode_parameters (transformer.py:230-242) and the residual of :462-463. */
int gpt_decode_tokens(int dtype, int n_tok, const void* feat, int T, const void* const* host_w,
                      const int* host_n, const int* host_off, const float* resid, float* out,
                      int B, int d, int D, void* stream);

/* DDPM elementwise kernels.  `coef` is a device fp32 table [n_steps, GPT_DDPM_NCOEF] built on the host in
 * float64 and cast once (the reference casts per gather, gaussian_diffusion.py:864); t: int64 [B]. */
#define GPT_DDPM_NCOEF 8
#define GPT_C_SQRT_AC 0      /* sqrt_alphas_cumprod            */
#define GPT_C_SQRT_1MAC 1    /* sqrt_one_minus_alphas_cumprod  */
#define GPT_C_POST1 2        /* posterior_mean_coef1           */
#define GPT_C_POST2 3        /* posterior_mean_coef2           */
#define GPT_C_LOGVAR 4       /* model log variance (FIXED_LARGE or FIXED_SMALL) */
#define GPT_C_RECIP 5        /* sqrt_recip_alphas_cumprod      */
#define GPT_C_RECIPM1 6      /* sqrt_recipm1_alphas_cumprod    */
#define GPT_C_SPARE 7

/* x_t = sqrt_ac[t] x0 + sqrt_1mac[t] noise            gaussian_diffusion.py:209-224 */
int gpt_q_sample(const float* x0, const float* noise, const int64_t* t, const float* coef, float* x_t,
                 int B, int D, void* stream);

/* One reverse step, fused (gaussian_diffusion.py:301-317 + :404-410):
 *   x0 = model_out                       (predict_xstart)   or  recip[t] x_t - recipm1[t] model_out
 *   x0 = clamp(x0, lo, hi) if use_clamp  (denoised_fn 'static' thresholding / clip_denoised)
 *   x_prev = post1[t] x0 + post2[t] x_t + (t != 0) exp(0.5 logvar[t]) noise
 * Writes x_out (may alias x_t) and optionally pred_xstart.  noise == NULL: the mean only (p_mean_variance). */
int gpt_posterior_step(const float* model_out, const float* x_t, const float* noise, const int64_t* t,
                       const float* coef, int predict_xstart, int use_clamp, float lo, float hi,
                       float* x_out, float* pred_xstart, int B, int D, void* stream);

/* One DDIM step, fused (gaussian_diffusion.py:507-552): x0 as in gpt_posterior_step, then
 *   eps    = (recip[t] x_t - x0) / recipm1[t]
 *   sigma  = eta sqrt((1 - abp) / (1 - ab)) sqrt(1 - ab / abp),  ab = alphas_cumprod[t], abp = alphas_cumprod[t-1] (1 at t = 0)
 *   x_prev = x0 sqrt(abp) + sqrt(1 - abp - sigma^2) eps + (t != 0) sigma noise
 * alphas_cumprod: device fp32 [n_steps] (the float64 table rounded once).  noise may be NULL when eta == 0. */
int gpt_ddim_step(const float* model_out, const float* x_t, const float* noise, const int64_t* t, const float* coef,
                  const float* alphas_cumprod, int predict_xstart, int use_clamp, float lo, float hi, float eta,
                  float* x_out, float* pred_xstart, int B, int D, void* stream);

/* Deterministic reverse ODE step x_t -> x_{t+1} (ddim_reverse_sample, gaussian_diffusion.py:559-592): the gpt_ddim_step
 * update towards alphas_cumprod_next[t] (= alphas_cumprod[t + 1], 0 for t = n_steps - 1) with eta = 0 and no noise. */
int gpt_ddim_reverse_step(const float* model_out, const float* x_t, const int64_t* t, const float* coef,
                          const float* alphas_cumprod, int n_steps, int predict_xstart, int use_clamp, float lo, float hi,
                          float* x_out, float* pred_xstart, int B, int D, void* stream);

/* out[b] = mean_i KL(N(s * x[b,i], exp(lv)) || N(0, 1)) / ln 2  with s = sqrt_alphas_cumprod[T-1], lv =
 * log_one_minus_alphas_cumprod[T-1] (_prior_bpd, gaussian_diffusion.py:783-797, normal_kl of diffusion_utils.py:15-41;
 * Gpt/data/prepare_checkpoints.py:98 reports it when choosing openai_coeff). */
int gpt_prior_bpd(const float* x_start, float sqrt_ac_last, float log_one_minus_ac_last, float* out, int B, int D, void* stream);

/* mse[b] = mean_d (target - out)^2   (diffusion_utils.py:8-12 over gaussian_diffusion.py:773) */
int gpt_mse_rows(const float* target, const float* out, float* mse, int B, int D, void* stream);
/* grad_out[b, :] = scale[b] * 2 (out - target) / D     (backward of the above; scale = dL/dmse) */
int gpt_mse_rows_bwd(const float* target, const float* out, const float* scale, float* grad_out,
                     int B, int D, void* stream);

/* Philox4x32-10 standard normal fill (Box-Muller), counter = element index, key = (seed, stream_id). */
int gpt_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);
/* out[i] = value, int64 (timestep vectors of the sampling loop, gaussian_diffusion.py:493) */
int gpt_fill_i64(int64_t* out, int64_t n, int64_t value, void* stream);

/* dtype conversion helpers (packed bf16 weight copies; round-to-nearest-even). */
int gpt_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int gpt_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* dst[c, r] = src[r, c]  for a row-major [rows, cols] matrix of `dtype`; dst rows have leading dimension
 * ld_dst >= rows and columns [rows, ld_dst) are written as zero (K padding of the wgrad GEMM operands). */
int gpt_transpose(int dtype, const void* src, void* dst, int rows, int cols, int64_t ld_dst, void* stream);

/* ---- backward kernels (what autograd derives for transformer.py:34-72, 215-242 in the reference) ----
 * LayerNorm backward: gout[rows,d] (fp32) = (resid ? resid : 0) + dLN/dx(dy); gout_t: optional copy in gout_t_dtype;
 * dgamma/dbeta (fp32 [d]) are ACCUMULATED (atomics).  dy: dy_dtype [rows,d]; x fp32; mean/rstd from the forward. */
int gpt_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                      const float* gamma, const float* resid, float* gout, void* gout_t, int gout_t_dtype,
                      float* dgamma, float* dbeta, int rows, int d, void* stream);
/* Same, and gcol (fp32 [d], optional) ACCUMULATES the column sums of gout: gout is the output gradient of the Linear that
 * feeds the residual stream there (mlp.2 / attn.proj, transformer.py:66-70), so gcol is that Linear's bias gradient and the
 * separate gpt_colsum pass over gout_t is not needed. */
int gpt_layernorm_bwd_colsum(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                             const float* gamma, const float* resid, float* gout, void* gout_t, int gout_t_dtype,
                             float* dgamma, float* dbeta, float* gcol, int rows, int d, void* stream);
/* dqkv[B*T,3d] from qkv[B*T,3d] and dy[B*T,d] (gradient of the attention output), all `dtype`. */
int gpt_attention_bwd(int dtype, const void* qkv, const void* dy, void* dqkv, int B, int T, int n_head, int hd,
                      void* stream);
/* out[c] += sum_r X[r, c]  (bias and position-embedding gradients; atomics). */
int gpt_colsum(int dtype, const void* X, int64_t ld, float* out, int rows, int cols, void* stream);
/* C[m,n] (+)= sum_k A[m*a_sm + k*a_sk] * B[k*b_sk + n*b_sn], fp32 FMA, arbitrary strides (ragged encoder /
 * decoder gradients). */
int gpt_small_gemm(int a_dtype, const void* A, int64_t a_sm, int64_t a_sk, int b_dtype, const void* B,
                   int64_t b_sk, int64_t b_sn, float* C, int64_t ldc, int M, int N, int K, int accumulate,
                   void* stream);

/* gpt_small_gemm over a ragged batch of problems in ONE launch (<= 40; all tokens of the encoder or the decoder).
 * The pointer / size arrays are host arrays (copied into the kernel arguments); element strides and dtypes are
 * shared, every problem has its own M, N, K and ldc. */
int gpt_small_gemm_batched(int n_problems, int a_dtype, const void* const* host_A, int64_t a_sm, int64_t a_sk,
                           int b_dtype, const void* const* host_B, int64_t b_sk, int64_t b_sn, float* const* host_C,
                           const int* host_ldc, const int* host_M, const int* host_N, const int* host_K,
                           int accumulate, void* stream);

/* Permutation augmentation (Gpt/data/augment.py:25-75 with data_gen/train_mnist.py:28-50 and
 * data_gen/train_cifar10.py:27-47).  perms: int32 device array [n_samples, perm_len] holding, per sample,
 * the hidden-unit permutations back to back (MLP: p[H]; CNN: p1[C1] then p2[C2]).
 * index: int64 [n_samples, D]. */
int gpt_perm_index_mlp(const int32_t* perms, int64_t* index, int n_samples, int n_in, int hidden, int n_out,
                       void* stream);
int gpt_perm_index_cnn(const int32_t* perms, int64_t* index, int n_samples, int c_in, int c1, int c2, int ksz,
                       int n_out, void* stream);
/* The Cartpole policy (data_gen/train_rl.py:51-74 random_permute_actor on VisActorCritic, train_rl.py:95-121; Gpt/tasks.py:73-81):
 * `prefix` leading scalars kept in place (log_std), then n_dims - 1 Linears of widths host_dims[0..n_dims) (weight [out, in]
 * then bias, state_dict order).  perms: per sample the permutations of the HIDDEN widths host_dims[1..n_dims-1) back to back,
 * in the order the reference draws them (input layer first).  index: int64 [n_samples, D]. */
int gpt_perm_index_chain(const int32_t* perms, int64_t* index, int n_samples, int prefix, int n_dims, const int* host_dims,
                         void* stream);
/* dst[s, i] = src[s, index[s, i]] for n_nets tensors sharing the index (both checkpoints of a pair). */
int gpt_perm_gather(const float* const* host_src, float* const* host_dst, int n_nets, const int64_t* index,
                    int n_samples, int D, void* stream);

/* Training-pair fetch from a checkpoint store held in HBM (ParameterDataset.__getitem__, Gpt/data/dataset_lmdb.py:218-225:
 * database[key_0], database[key_1] -> normalize (Gpt/data/normalization.py:23-25) -> aug_fn on both):
 *   out0[s, i] = store[row0[s] * ld + j] * scale,  out1[s, i] = store[row1[s] * ld + j] * scale,  j = index ? index[s, i] : i
 * store: fp32 [n_rows, ld] (one checkpoint per row), row0/row1: int64 device [n_samples] (the caller keeps them < n_rows),
 * index: int64 device [n_samples, D] from gpt_perm_index_* or NULL, scale = openai_coeff (1 for the identity normaliser). */
int gpt_pair_gather(const float* store, int64_t ld, int64_t n_rows, const int64_t* row0, const int64_t* row1,
                    const int64_t* index, float scale, float* out0, float* out1, int n_samples, int D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-path entry points: one host call launches every kernel of a denoiser forward / of the whole
 * reverse-diffusion loop on `stream` (no host synchronisation, no allocation), so the Python caller pays
 * one FFI crossing per forward instead of ~90.  All pointers inside the structs are device pointers,
 * the structs themselves live in host memory.
 * ------------------------------------------------------------------------------------------------ */
#define GPT_MAX_TOKENS 40
#define GPT_SRC_X 0       /* token is a slice of the noised vector x          */
#define GPT_SRC_XPREV 1   /* token is a slice of the starting vector x_prev   */
#define GPT_SRC_SCALAR 2  /* token is one frequency-embedded scalar           */

typedef struct {
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
  const void* wqkv;  const float* bqkv;    /* [3d, d] rows = query | key | value, (dtype); [3d] fp32 */
  const void* wproj; const float* bproj;   /* [d, d]   */
  const void* w1;    const float* b1;      /* [4d, d]  */
  const void* w2;    const float* b2;      /* [d, 4d]  */
  /* hipEvent_t or NULL.  Non-NULL: the forward makes its stream wait for this event before the layer's first kernel - the
   * layer's operands are being delivered asynchronously (the sharded optimizer step all-gathers them on another stream while
   * the next forward already runs its first layers: g.pt_amd/trainstep.py).  The event must have been recorded. */
  const void* ready;
} gpt_layer_t;

typedef struct {
  int dtype, d, n_head, n_layer, T, D, nfreq, n_scalar, n_in_tok, n_out_tok;
  const float* freqs;                      /* [nfreq] */
  const float* pos_emb;                    /* [T, d] */
  const float *ln_in_w, *ln_in_b, *ln_f_w, *ln_f_b;
  const void* enc_w[GPT_MAX_TOKENS];       /* (dtype) [d, enc_w_ld], zero padded to a multiple of 32 */
  int64_t enc_w_ld[GPT_MAX_TOKENS];
  int enc_k[GPT_MAX_TOKENS], enc_src[GPT_MAX_TOKENS], enc_off[GPT_MAX_TOKENS];
  const void* dec_w[GPT_MAX_TOKENS];       /* (dtype) [dec_n, d] */
  int dec_n[GPT_MAX_TOKENS], dec_off[GPT_MAX_TOKENS];
  const gpt_layer_t* layers;               /* host array [n_layer] */
  const void* dec_ready;                   /* hipEvent_t or NULL: as gpt_layer_t.ready, for the decoder Linears (dec_w) */
} gpt_model_t;

/* Activations.  With layer_stride == 0 every layer reuses the same buffers (inference); with
 * layer_stride == 1 buffer `p` of layer l lives at p + l * (its per-layer size), which keeps what the
 * backward needs.  M = B_max * T rows. */
typedef struct {
  int B_max, layer_stride;
  float* scal;     /* [B, n_scalar * (2 nfreq + 1)]                                  */
  float* enc;      /* [M, d]   encoder output before ln_in                           */
  float* x;        /* [(2 n_layer + 1) if stride else 1][M, d]  residual stream      */
  void* h1;        /* (dtype) [L?][M, d]   ln1 output                                */
  void* qkv;       /* (dtype) [L?][M, 3d]                                            */
  void* att;       /* (dtype) [L?][M, d]   attention output                          */
  void* h2;        /* (dtype) [L?][M, d]   ln2 output                                */
  void* pre;       /* (dtype) [L][M, 4d]   fc1 pre-activation (stride mode only)     */
  void* act;       /* (dtype) [L?][M, 4d]  gelu output                               */
  void* hf;        /* (dtype) [M, d]       ln_f output                               */
  float* stats;    /* fp32 [(2 L + 2)][2][M]  LayerNorm mean / rstd (stride mode only, else NULL) */
} gpt_workspace_t;

/* Gradient outputs and scratch of gpt_denoiser_backward.  Every g_* is fp32 in the parameter's own layout and
 * is ACCUMULATED into (the caller zeroes them): g_wqkv is [3d, d] = the query | key | value weight gradients
 * back to back (the host makes the three .grad tensors views of it). */
typedef struct {
  const void *wqkv_t, *wproj_t, *w1_t, *w2_t;   /* (dtype) transposed weights [d,3d] [d,d] [d,4d] [4d,d] for dgrad */
  float *g_ln1_w, *g_ln1_b, *g_ln2_w, *g_ln2_b, *g_wqkv, *g_bqkv, *g_wproj, *g_bproj, *g_w1, *g_b1, *g_w2, *g_b2;
} gpt_layer_grad_t;

typedef struct {
  const gpt_layer_grad_t* layers;               /* host array [n_layer] */
  float *g_pos_emb, *g_ln_in_w, *g_ln_in_b, *g_ln_f_w, *g_ln_f_b;
  float* g_enc_w[GPT_MAX_TOKENS];               /* fp32 [d, k_i] */
  float* g_dec_w[GPT_MAX_TOKENS];               /* fp32 [n_j, d] */
  float* g;        /* scratch fp32 [M, d]   residual-stream gradient               */
  void* gt;        /* scratch (dtype) [M, d]                                         */
  void* d_h;       /* scratch (dtype) [M, d]                                         */
  void* d_qkv;     /* scratch (dtype) [M, 3d]                                        */
  void* d_act;     /* scratch (dtype) [M, 4d]                                        */
  void* tA;        /* scratch (dtype) [4d, Mp]  transposed operand, Mp = M rounded up to 64 */
  void* tB;        /* scratch (dtype) [4d, Mp]                                       */
  float* d_tok;    /* scratch fp32 [M, d]   decoder / encoder token gradients        */
  /* != 0: the four weight gradients of every Block (g_wqkv, g_wproj, g_w1, g_w2) are STORED, not accumulated: they need no
   * zeroing before the call and are not read (no read-modify-write of 12 d^2 floats per Block).  Only valid where the grouped
   * TN launch computes them (GPT_BF16, n_embd % 256 == 0); rejected otherwise.  Every other gradient is
   * accumulated as before. */
  int wgrad_overwrite;
  /* G = 2..4: the grouped TN launches of up to G consecutive Blocks (inside one gpt_denoiser_backward_stages call) are merged
   * into one grid, so that their tiles share rounds of CUs (n_embd 1536: 432 tiles per Block = 2 rounds of 256 CUs, 1728 per
   * four Blocks = 7 rounds instead of 8).  The operands of a deferred launch must survive the Blocks in between: gt, d_act and
   * d_qkv then hold G slots of [M, d], [M, 4d], [M, 3d] back to back (M = B * T of the call).  0 / 1: one launch per Block. */
  int wgrad_defer;
  /* device fp32 scalar or NULL.  Non-NULL (needs wgrad_overwrite): every weight-gradient tile the grouped TN launch stores also
   * adds the sum of its squares here (the caller zeroes it before the backward): the Blocks' share of the squared gradient norm
   * that torch.nn.utils.clip_grad_norm_ (main.py:70) needs, taken from the accumulators instead of a second pass over 1.4 GB. */
  float* wgrad_sumsq;
  /* ABI 4.  Optional scratch (NULL / 0: none) of ln_partials_bytes bytes: the LayerNorm backward launches of a call then STORE
   * their per-workgroup dgamma / dbeta / bias-gradient sums there (256 x 3 x n_embd floats per launch) and ONE launch at the end of
   * the call adds them to the gradients in a fixed order, instead of 2 - 3 n_embd float atomics per workgroup on the same
   * addresses (a third of the kernel's time at 3328 rows).  Launches the buffer has no room for keep the atomics.  Full size:
   * (2 n_layer + 2) x 256 x 3 x n_embd x 4 bytes. */
  float* ln_partials;
  int64_t ln_partials_bytes;
} gpt_backward_t;

/* Backward of gpt_denoiser_forward run on a layer_stride == 1 workspace: d_out[B, D] = dLoss/d(out).
 * In GPT_BF16 mode every product of the backward, the ragged encoder / decoder gradients included, multiplies bf16-rounded
 * operands with fp32 accumulation (GPT_F32: exact fp32 throughout).
 * x / x_prev / t inputs are not differentiated (training_losses needs parameter gradients only). */
int gpt_denoiser_backward(const gpt_model_t* m, const gpt_workspace_t* ws, const gpt_backward_t* bw, const float* x,
                          const float* x_prev, const float* d_out, int B, void* stream);

/* The same backward in stages, so that the host can start the gradient all-reduce of finished layers while the
 * rest still runs (what DistributedDataParallel's bucketed reducer does for main.py:229-240).  Stage 0 = decoder
 * Linears + ln_f; stage s in 1..n_layer = block n_layer - s; stage n_layer + 1 = pos_emb, ln_in, encoder Linears.
 * Runs stages [stage_begin, stage_end) - consecutive calls must cover 0 .. n_layer + 2 in order on one stream. */
int gpt_denoiser_backward_stages(const gpt_model_t* m, const gpt_workspace_t* ws, const gpt_backward_t* bw,
                                 const float* x, const float* x_prev, const float* d_out, int B, int stage_begin,
                                 int stage_end, void* stream);

/* out[B, D] = Gpt.forward(x, t, loss_target, loss_prev, x_prev)   (transformer.py:440-464) */
int gpt_denoiser_forward(const gpt_model_t* m, const gpt_workspace_t* ws, const float* x, const int64_t* t,
                         const void* loss_target, const void* loss_prev, int loss_is_f64, const float* x_prev,
                         float* out, int B, void* stream);

/* The whole p_sample_loop (gaussian_diffusion.py:458-505) for one batch, in place on x[B, D]:
 * for i = n_steps-1 .. 0:  out = denoiser(x, timestep_map[i]);  x = posterior_step(out, x, noise_i, i).
 * noise: NULL -> in-kernel Philox with (seed, i);  else fp32 [n_steps, B, D] holding the noise drawn at index i
 * in slab i (injected-noise parity mode).  t_model / t_index: int64 [B] scratch; model_out: fp32 [B, D] scratch;
 * noise_buf: fp32 [B, D] scratch (Philox mode).  host_timestep_map: int64 host array [n_steps]. */
int gpt_sample_loop(const gpt_model_t* m, const gpt_workspace_t* ws, float* x, const void* loss_target,
                    const void* loss_prev, int loss_is_f64, const float* x_prev, const float* coef,
                    const int64_t* host_timestep_map, int n_steps, int first_step, int last_step,
                    int predict_xstart, int use_clamp, float lo, float hi,
                    const float* noise, uint64_t seed, int64_t* t_model, int64_t* t_index, float* model_out,
                    float* noise_buf, int B, void* stream);

/* The whole ddim_sample_loop (gaussian_diffusion.py:594-674, the sampler playground.py:49-56 calls), same contract as
 * gpt_sample_loop with gpt_ddim_step as the update.  eta == 0: deterministic, noise / noise_buf may be NULL. */
int gpt_ddim_sample_loop(const gpt_model_t* m, const gpt_workspace_t* ws, float* x, const void* loss_target,
                         const void* loss_prev, int loss_is_f64, const float* x_prev, const float* coef,
                         const float* alphas_cumprod, float eta, const int64_t* host_timestep_map, int n_steps,
                         int first_step, int last_step, int predict_xstart, int use_clamp, float lo, float hi,
                         const float* noise, uint64_t seed, int64_t* t_model, int64_t* t_index, float* model_out,
                         float* noise_buf, int B, void* stream);

/* Quality metric of sampled MNIST-MLP parameter vectors, all B nets at once (replaces moduleify + the per-net loop
 * of batch_test_fn, Gpt/vis.py:398-413, 539-570, around test_epoch, data_gen/train_mnist.py:324-332).
 * H: fp32 [B * hidden, ldh] = fc1 pre-activations WITHOUT bias, image index contiguous (one gpt_gemm_nt over the
 * fc1 weights of every net); nets: fp32 [B, net_stride] flat vectors with fc1.bias / fc2.weight / fc2.bias at the
 * given element offsets; labels: int64 [n_img].  loss[b] = mean cross-entropy, err[b] = 100 * mean(argmax != label).
 * hidden <= 64, classes <= 16. */
int gpt_mlp_head(const float* H, int64_t ldh, const float* nets, int64_t net_stride, int off_b1, int off_w2, int off_b2,
                 const int64_t* labels, float* loss, float* err, int B, int hidden, int classes, int n_img, void* stream);

/* Quality metric for the CIFAR-10 ConvNet (replaces moduleify + the per-net loop of batch_test_fn, Gpt/vis.py:398-413,
 * 539-570, around data_gen/train_cifar10.py:53-68 and test_epoch :426-434): mean cross-entropy loss[b] and top-1 error
 * err[b] (percent) of B flat parameter vectors nets[b * net_stride + ...] = conv1.weight [16,3,3,3] | conv2.weight
 * [32,16,3,3] | fc.weight [10,32] | fc.bias [10] on images [n_img,3,32,32] fp32 / labels int64 [n_img], in one fused
 * kernel.  The images are cut into `chunks` ranges (grid = ceil(B/4) x chunks); partial = caller scratch of
 * B * chunks * 2 floats. */
int gpt_cnn_eval(const float* images, const int64_t* labels, const float* nets, int64_t net_stride, float* partial, int chunks,
                 float* loss, float* err, int B, int n_img, void* stream);

/* ---- optimizer step on one flat fp32 parameter arena (training-step harness, main.py:66-74) ----------------------
 * A segment is one named parameter: `count` elements at element offset `start` of the arenas (starts are multiples
 * of 4 elements; the gap up to the next start is zero padding).  decay: 1 = AdamW weight-decay group (Linear
 * weights), 0 = no-decay group (biases, LayerNorm, pos_emb) - Gpt/models/transformer.py:163-213.
 * packed_off >= 0: the updated values are also written, converted to the packed dtype, to element offset
 * packed_off + (i / k) * ld + (i % k) of `packed` (k == ld: a plain cast copy; ld > k: rows zero-padded to ld,
 * the encoder-weight operand layout of gpt_encode_tokens); packed_off < 0: no copy. */
typedef struct {
  int64_t start, count, packed_off;
  int32_t k, ld, decay, reserved;
} gpt_opt_segment_t;

/* out[0] = sum(grads^2)   (the squared global norm of torch.nn.utils.clip_grad_norm_, main.py:70).
 * scratch: fp32 [1024] device buffer.  n % 4 == 0. */
int gpt_grad_sumsq(const float* grads, int64_t n, float* scratch, float* out, void* stream);

/* One fused pass over the arenas (all fp32 [n], n % 4 == 0, 16-byte aligned; segments: DEVICE array sorted by start):
 *   g' = grads * grad_scale * min(1, clip / (sqrt(*grad_norm_sq) * grad_scale + 1e-6))      (clip <= 0: no clipping)
 *   torch.optim.AdamW single-tensor update of params / exp_avg / exp_avg_sq at step `step` (>= 1)
 *   ema = ema * ema_decay + params * (1 - ema_decay)      (Gpt/utils.py:146-152; ema NULL: skipped)
 *   packed copies per segment (packed NULL: skipped);  grads = 0 when zero_grad (optimizer.zero_grad, main.py:66). */
int gpt_adamw_ema_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema, void* packed,
                       int packed_dtype, const gpt_opt_segment_t* segments, int n_segments, int64_t n,
                       const float* grad_norm_sq, float clip, float grad_scale, double lr, double beta1, double beta2,
                       double eps, double weight_decay, int step, double ema_decay, int zero_grad, void* stream);

/* The packed copies alone (after load_state_dict or any other write to the parameters that bypasses the step). */
int gpt_pack_segments(const float* params, void* packed, int packed_dtype, const gpt_opt_segment_t* segments,
                      int n_segments, int64_t n, void* stream);

/* ---- data-parallel collectives on RCCL over xGMI (dp_init / dp_destroy / dp_allreduce_grads / dp_allgather /
 * dp_broadcast / dp_barrier) -------------------------------------------------------------------------------------------------
 * For callers that do not use torch.distributed (the Python host's default route, whose "nccl" backend is the same RCCL).
 * One communicator per process, one process per GPU (main.py:344-354); librccl is loaded on first use.  Rank 0 obtains an id
 * with gpt_dp_unique_id and hands its GPT_DP_ID_BYTES bytes to the other ranks by any channel (the reference uses a TCP store,
 * main.py:328-333); every rank then calls gpt_dp_init with the device it will use current.  All calls are stream-ordered and
 * in place where one buffer is given; counts are elements of `dtype` (GPT_F32, GPT_BF16, GPT_DP_I64).  Status: 0 ok, negative =
 * rejected, >= 1000 = 1000 + ncclResult_t. */
#define GPT_DP_ID_BYTES 128
#define GPT_DP_I64 2
int gpt_dp_unique_id(void* host_id_out);
int gpt_dp_init(const void* host_id, int rank, int world);
int gpt_dp_destroy(void);
int gpt_dp_rank(void);
int gpt_dp_world(void);
/* grads[n] (fp32) = SUM over ranks, in place: the gradient exchange of DistributedDataParallel (main.py:229-240); the 1/world
 * scale is folded into gpt_adamw_ema_step's grad_scale. */
int gpt_dp_allreduce_grads(float* grads, int64_t n, void* stream);
/* The two halves of that exchange for the sharded optimizer step (g.pt_amd/trainstep.py): recv[recv_count] = SUM over ranks of
 * send[rank * recv_count ...] (send holds world * recv_count elements); recv[world * send_count] = every rank's send[send_count]
 * (send may be recv + rank * send_count: in place).  gpt_dp_allgather is also Gpt/distributed.py:170-183 (all_gather). */
int gpt_dp_reduce_scatter(const void* send, void* recv, int64_t recv_count, int dtype, void* stream);
int gpt_dp_allgather(const void* send, void* recv, int64_t send_count, int dtype, void* stream);
/* buf[count] = root's buf (initial parameter broadcast of DDP, main.py:231-236; rank0_to_all, Gpt/distributed.py:186-189). */
int gpt_dp_broadcast(void* buf, int64_t count, int dtype, int root, void* stream);
/* synchronize() (Gpt/distributed.py:152-157): a one-word all-reduce, then the host waits for `stream` - the one entry point
 * that synchronises. */
int gpt_dp_barrier(void* stream);

/* The same copies for parameters held as SEPARATE tensors (the reference's nn.Module keeps 235 of them): one
 * launch over a DEVICE table of items.  Item i: `count` fp32 elements at src (16-byte aligned) are written to dst (16-byte
 * aligned) as dst_dtype (GPT_F32 / GPT_BF16), element j at dst[(j / k) * ld + j % k] (k == ld: a plain cast copy; ld > k: rows
 * zero-padded by the caller, the K-padded encoder operands).  first4 = the sum of ceil(count / 4) over the items before it;
 * n4 = that sum over all items.  Used by the Python engine to re-derive its packed operands (fused query|key|value weight and
 * bias, bf16 copies) from the live parameters before every forward / sampling loop, because writers such as the reference's
 * EMA update (Gpt/utils.py:146-152: par.data.mul_().add_()) are invisible to torch's version counters. */
typedef struct {
  const float* src;
  void* dst;
  int64_t first4, count;
  int32_t k, ld, dst_dtype, reserved;
} gpt_pack_item_t;
int gpt_pack_tensors(const gpt_pack_item_t* items, int n_items, int64_t n4, void* stream);
/* The same, skipped ON THE DEVICE when no source element changed since the last call with this `state` (4 x uint64 of device
 * memory, zero before the first call; no host synchronisation): a streaming 64-bit position-sensitive checksum of all sources
 * (read only: a third of the pack's bytes) decides whether the pack kernel does anything.  For callers that run the model in a
 * host loop (the reference's per-step p_sample_loop_progressive, gaussian_diffusion.py:458-505: 1000 forwards over weights that
 * do not change).  The destinations must not have been written by anyone else in between. */
int gpt_pack_tensors_if_changed(const gpt_pack_item_t* items, int n_items, int64_t n4, uint64_t* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif
