// Whole-path host drivers: one call launches every kernel of a denoiser forward, or of the whole
// reverse-diffusion loop, on the caller's stream.  No allocation, no synchronisation, graph-capturable.
//   Gpt.forward                     Gpt/models/transformer.py:440-464 (+ GPT.forward :302-314, Block :69-72)
//   p_sample_loop_progressive       Gpt/diffusion/gaussian_diffusion.py:458-505
#include "common.h"
#include <stdlib.h>

#define CHECK(call)             \
  do {                          \
    int _st = (call);           \
    if (_st != 0) return _st;   \
  } while (0)

static inline size_t esize(int dtype) { return dtype == GPT_BF16 ? 2 : 4; }

// operands delivered asynchronously (gpt_layer_t.ready / gpt_model_t.dec_ready): everything launched on `stream` after this
// call runs after the event
static int wait_ready(const void* ev, void* stream) {
  if (!ev) return 0;
  const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
  GPT_REQUIRE(e == hipSuccess, "gpt_denoiser_forward: hipStreamWaitEvent failed: %s", hipGetErrorString(e));
  return 0;
}

static int validate(const gpt_model_t* m, const gpt_workspace_t* ws, int B, const char* who) {
  GPT_REQUIRE(m && ws, "%s: null model/workspace", who);
  GPT_REQUIRE(m->dtype == GPT_F32 || m->dtype == GPT_BF16, "%s: bad dtype %d", who, m->dtype);
  GPT_REQUIRE(B > 0 && B <= ws->B_max, "%s: batch %d exceeds workspace capacity %d", who, B, ws->B_max);
  GPT_REQUIRE(m->n_in_tok == m->T && m->T <= 32 && m->T <= GPT_MAX_TOKENS, "%s: %d tokens unsupported (max 32)", who, m->T);
  GPT_REQUIRE(m->n_out_tok > 0 && m->n_out_tok <= m->T, "%s: bad n_out_tok", who);
  GPT_REQUIRE(m->d % m->n_head == 0, "%s: n_embd %% n_head != 0", who);
  const int hd = m->d / m->n_head;
  GPT_REQUIRE(hd % 32 == 0 && hd <= 128, "%s: head dim %d unsupported (multiple of 32, <= 128)", who, hd);
  GPT_REQUIRE(m->d % 64 == 0, "%s: n_embd must be a multiple of 64", who);
  GPT_REQUIRE(m->layers && m->n_layer > 0, "%s: no layers", who);
  GPT_REQUIRE(!ws->layer_stride || (ws->pre && ws->stats), "%s: training workspace needs pre/stats", who);
  return 0;
}

// enc_cached != 0: the encoder output of the tokens that do not depend on x or t (x_prev, the loss scalars) is already in
// ws->enc from an earlier call with the same x_prev / losses (the sampling loops): only the other tokens are encoded.
// Training (kept activations): the last Block behind its attention, the final LayerNorm and the decoder touch only the rows of the
// tokens the decoder reads (the loss does not depend on the others: their rows of the last Block's projection / MLP are dead
// work in the forward and all-zero gradients in the backward).  Forward and backward drivers must agree, so both ask this:
// bf16 on the grouped TN weight-gradient route (any row count: the TN kernel zero-fills a ragged last K-tile).
static bool prune_last_train(const gpt_model_t* m, int B) {
  static const bool on = [] {
    const char* e = getenv("GPT_PRUNE_LAST_TRAIN");   // 0: all rows (cross-check, A/B)
    const char* tn = getenv("GPT_WGRAD_TN");
    return (!e || atoi(e) != 0) && (!tn || atoi(tn) != 0);
  }();
  const int To = m->n_out_tok, T = m->T;
  (void)B;
  return on && m->dtype == GPT_BF16 && m->n_layer > 0 && To < T && m->d % 256 == 0;
}

static int denoiser_forward(const gpt_model_t* m, const gpt_workspace_t* ws, const float* x, const int64_t* t,
                            const void* loss_target, const void* loss_prev, int loss_is_f64, const float* x_prev, float* out,
                            int B, int enc_cached, void* stream) {
  CHECK(validate(m, ws, B, "gpt_denoiser_forward"));
  GPT_REQUIRE(x && t && loss_target && loss_prev && x_prev && out, "gpt_denoiser_forward: null operand");
  const int d = m->d, T = m->T, M = B * T, L = m->n_layer, dt = m->dtype;
  const size_t es = esize(dt);
  const int W = 2 * m->nfreq + 1;
  const bool keep = ws->layer_stride != 0;
  const size_t Mmax = (size_t)ws->B_max * T;

  CHECK(gpt_scalar_tokens(t, loss_target, loss_prev, loss_is_f64, m->freqs, m->nfreq, m->n_scalar, ws->scal, B, stream));

  const float* src[GPT_MAX_TOKENS];
  const void* wsel[GPT_MAX_TOKENS];
  int64_t src_ld[GPT_MAX_TOKENS], w_ld[GPT_MAX_TOKENS];
  int ksel[GPT_MAX_TOKENS], slot[GPT_MAX_TOKENS];
  int n_sel = 0;
  for (int i = 0; i < m->T; ++i) {
    const float* sp;
    int64_t sld;
    bool varies;                       // with x or t
    switch (m->enc_src[i]) {
      case GPT_SRC_X: sp = x + m->enc_off[i]; sld = m->D; varies = true; break;
      case GPT_SRC_XPREV: sp = x_prev + m->enc_off[i]; sld = m->D; varies = false; break;
      case GPT_SRC_SCALAR: sp = ws->scal + m->enc_off[i]; sld = (int64_t)m->n_scalar * W; varies = m->enc_off[i] == 0; break;
      default: gpt_set_error("gpt_denoiser_forward: token %d: bad source kind %d", i, m->enc_src[i]); return -1;
    }
    if (enc_cached && !varies) continue;
    src[n_sel] = sp; src_ld[n_sel] = sld; wsel[n_sel] = m->enc_w[i]; w_ld[n_sel] = m->enc_w_ld[i]; ksel[n_sel] = m->enc_k[i];
    slot[n_sel++] = i;
  }
  CHECK(gpt_encode_tokens_subset(dt, n_sel, src, src_ld, wsel, w_ld, ksel, slot, T, ws->enc, B, d, stream));

  auto xbuf = [&](int idx) { return ws->x + (keep ? (size_t)idx * Mmax * d : 0); };
  auto lay = [&](void* base, int l, size_t width) {
    return (void*)((char*)base + (keep ? (size_t)l * Mmax * width * es : 0));
  };
  auto stat = [&](int idx, int which) { return keep ? ws->stats + ((size_t)idx * 2 + which) * Mmax : (float*)nullptr; };
  // inference only: the last Block behind its attention on the decoded tokens' rows alone (see there); needs the compact fp32
  // residual rows to fit behind the compact activations in `act`: (T - To) * 4 * es >= To * 4
  static const bool prune_on = [] {
    const char* e = getenv("GPT_PRUNE_LAST");   // 0: all rows through the last Block (cross-check, A/B)
    return !e || atoi(e) != 0;
  }();
  const bool prune = prune_on && !keep && L > 0 && m->n_out_tok < T && (size_t)(T - m->n_out_tok) * 4 * es >= (size_t)m->n_out_tok * 4;

  // ln_in + position embedding (transformer.py:226, 308-309), then the first Block's ln1: one launch in the inference forward
  // (no statistics kept), two otherwise
  static const bool ln_pair = [] { const char* e = getenv("GPT_LN_FUSE0"); return !e || atoi(e) != 0; }();   // 0: two launches (A/B)
  if (ln_pair && !keep && L > 0 && gpt_layernorm_pair_has(d)) {
    CHECK(gpt_layernorm_pair(ws->enc, m->ln_in_w, m->ln_in_b, m->pos_emb, T, xbuf(0), m->layers[0].ln1_w, m->layers[0].ln1_b,
                             lay(ws->h1, 0, d), dt, M, d, 1e-5f, stream));
  } else {
    CHECK(gpt_layernorm(ws->enc, m->ln_in_w, m->ln_in_b, m->pos_emb, T, xbuf(0), GPT_F32, stat(0, 0), stat(0, 1), M, d, 1e-5f, stream));
    // Every residual Linear is launched together with the LayerNorm that reads its result (gpt_gemm_resadd_layernorm: one
    // fused pass at small batches, the two kernels otherwise): proj + ln2, mlp.2 + the next block's ln1 (ln_f after the last).
    CHECK(gpt_layernorm(xbuf(0), m->layers[0].ln1_w, m->layers[0].ln1_b, nullptr, 0, lay(ws->h1, 0, d), dt, stat(1, 0), stat(1, 1), M, d,
                        1e-5f, stream));
  }
  for (int l = 0; l < L; ++l) {
    const gpt_layer_t& ly = m->layers[l];
    float* xin = xbuf(2 * l);
    float* xmid = xbuf(2 * l + 1);
    float* xout = xbuf(2 * l + 2);
    void* h1 = lay(ws->h1, l, d);
    void* qkv = lay(ws->qkv, l, 3 * (size_t)d);
    void* att = lay(ws->att, l, d);
    void* h2 = lay(ws->h2, l, d);
    void* act = lay(ws->act, l, 4 * (size_t)d);
    CHECK(wait_ready(ly.ready, stream));
    CHECK(gpt_gemm_nt(dt, h1, d, ly.wqkv, d, ly.bqkv, qkv, 3 * d, nullptr, 0, M, 3 * d, d, GPT_EPI_STORE, stream));
    const bool last = l + 1 == L;
    if (last && keep && prune_last_train(m, B)) {
      const int To = m->n_out_tok, Mc = B * To;
      void* pre = lay(ws->pre, l, 4 * (size_t)d);
      CHECK(gpt_attention_rows(dt, qkv, att, B, T, To, m->n_head, d / m->n_head, stream));
      CHECK(gpt_gather_token_rows(xin, xmid, B, T, To, d, stream));                 // compact residual rows: [B * To, d]
      CHECK(gpt_gemm_resadd_layernorm(dt, att, d, ly.wproj, d, ly.bproj, xmid, d, xmid, d, ly.ln2_w, ly.ln2_b, h2, stat(2 + 2 * l, 0),
                                      stat(2 + 2 * l, 1), Mc, d, d, 1e-5f, stream));
      CHECK(gpt_gemm_nt(dt, h2, d, ly.w1, d, ly.b1, act, 4 * d, pre, 4 * d, Mc, 4 * d, d, GPT_EPI_GELU_AUX, stream));
      CHECK(gpt_gemm_resadd_layernorm(dt, act, 4 * d, ly.w2, 4 * d, ly.b2, xmid, d, xout, d, m->ln_f_w, m->ln_f_b, ws->hf, stat(2 * L + 1, 0),
                                      stat(2 * L + 1, 1), Mc, d, 4 * d, 1e-5f, stream));
      CHECK(wait_ready(m->dec_ready, stream));
      return gpt_decode_tokens(dt, To, ws->hf, To, m->dec_w, m->dec_n, m->dec_off, x_prev, out, B, d, m->D, stream);
    }
    if (last && prune) {
      // The decoder reads the first n_out_tok tokens only (transformer.py:237-240), and behind the last Block's attention rows
      // do not mix any more: its projection, LayerNorms and MLP run on those rows alone, compacted to [B * To, d] (the attention
      // kernel stores them so; the residual rows are gathered) - 15 of 26 tokens at the shipped plans = 58 % of 9/12 of the
      // Block's FLOPs.  Same arithmetic per row; what changes is which kernel instance a GEMM of that height takes.
      // (r01 measured +-0 for this at B = 320 on the kernels of the time; with gemm128.hip and the K-span passes: section 4.)
      const int To = m->n_out_tok, Mc = B * To;
      float* xc = reinterpret_cast<float*>((char*)ws->act + (size_t)Mc * 4 * d * es);   // the part of `act` the compact rows leave free
      CHECK(gpt_attention_rows(dt, qkv, att, B, T, To, m->n_head, d / m->n_head, stream));
      CHECK(gpt_gather_token_rows(xin, xc, B, T, To, d, stream));
      CHECK(gpt_gemm_resadd_layernorm(dt, att, d, ly.wproj, d, ly.bproj, xc, d, xc, d, ly.ln2_w, ly.ln2_b, h2, nullptr, nullptr, Mc, d, d,
                                      1e-5f, stream));
      CHECK(gpt_gemm_nt(dt, h2, d, ly.w1, d, ly.b1, act, 4 * d, nullptr, 0, Mc, 4 * d, d, GPT_EPI_GELU, stream));
      CHECK(gpt_gemm_resadd_layernorm(dt, act, 4 * d, ly.w2, 4 * d, ly.b2, xc, d, xc, d, m->ln_f_w, m->ln_f_b, ws->hf, nullptr, nullptr, Mc, d,
                                      4 * d, 1e-5f, stream));
      CHECK(wait_ready(m->dec_ready, stream));
      return gpt_decode_tokens(dt, To, ws->hf, To, m->dec_w, m->dec_n, m->dec_off, x_prev, out, B, d, m->D, stream);
    }
    CHECK(gpt_attention(dt, qkv, att, B, T, m->n_head, d / m->n_head, stream));
    CHECK(gpt_gemm_resadd_layernorm(dt, att, d, ly.wproj, d, ly.bproj, xin, d, xmid, d, ly.ln2_w, ly.ln2_b, h2, stat(2 + 2 * l, 0),
                                    stat(2 + 2 * l, 1), M, d, d, 1e-5f, stream));
    if (keep) {
      void* pre = lay(ws->pre, l, 4 * (size_t)d);
      CHECK(gpt_gemm_nt(dt, h2, d, ly.w1, d, ly.b1, act, 4 * d, pre, 4 * d, M, 4 * d, d, GPT_EPI_GELU_AUX, stream));
    } else {
      CHECK(gpt_gemm_nt(dt, h2, d, ly.w1, d, ly.b1, act, 4 * d, nullptr, 0, M, 4 * d, d, GPT_EPI_GELU, stream));
    }
    const float* nw = last ? m->ln_f_w : m->layers[l + 1].ln1_w;
    const float* nb = last ? m->ln_f_b : m->layers[l + 1].ln1_b;
    void* nh = last ? ws->hf : lay(ws->h1, l + 1, d);
    const int ns = last ? 2 * L + 1 : 1 + 2 * (l + 1);
    CHECK(gpt_gemm_resadd_layernorm(dt, act, 4 * d, ly.w2, 4 * d, ly.b2, xmid, d, xout, d, nw, nb, nh, stat(ns, 0), stat(ns, 1), M, d,
                                    4 * d, 1e-5f, stream));
  }
  CHECK(wait_ready(m->dec_ready, stream));
  CHECK(gpt_decode_tokens(dt, m->n_out_tok, ws->hf, T, m->dec_w, m->dec_n, m->dec_off, x_prev, out, B, d, m->D, stream));
  return 0;
}

extern "C" int gpt_denoiser_forward(const gpt_model_t* m, const gpt_workspace_t* ws, const float* x, const int64_t* t,
                                    const void* loss_target, const void* loss_prev, int loss_is_f64,
                                    const float* x_prev, float* out, int B, void* stream) {
  return denoiser_forward(m, ws, x, t, loss_target, loss_prev, loss_is_f64, x_prev, out, B, 0, stream);
}

// ddim: 0 = ancestral DDPM step (gpt_posterior_step), 1 = DDIM step with `eta` (gpt_ddim_step, needs alphas_cumprod)
static int sample_loop(const char* who, int ddim, float eta, const float* alphas_cumprod, const gpt_model_t* m,
                       const gpt_workspace_t* ws, float* x, const void* loss_target, const void* loss_prev, int loss_is_f64,
                       const float* x_prev, const float* coef, const int64_t* host_timestep_map, int n_steps, int first_step,
                       int last_step, int predict_xstart, int use_clamp, float lo, float hi, const float* noise,
                       uint64_t seed, int64_t* t_model, int64_t* t_index, float* model_out, float* noise_buf, int B,
                       void* stream) {
  CHECK(validate(m, ws, B, who));
  GPT_REQUIRE(x && coef && host_timestep_map && t_model && t_index && model_out, "%s: null operand", who);
  const bool need_noise = !ddim || eta != 0.f;
  GPT_REQUIRE(!need_noise || noise || noise_buf, "%s: need injected noise or a noise scratch buffer", who);
  GPT_REQUIRE(!ddim || alphas_cumprod, "%s: null alphas_cumprod", who);
  GPT_REQUIRE(n_steps > 0 && first_step < n_steps && last_step >= 0 && first_step >= last_step,
              "%s: bad step range [%d..%d] of %d", who, first_step, last_step, n_steps);
  GPT_REQUIRE(ws->layer_stride == 0, "%s: needs an inference workspace", who);
  const int64_t BD = (int64_t)B * m->D;
  static const bool fuse_noise = [] {
    const char* e = getenv("GPT_POST_PHILOX");   // 0: gpt_randn + gpt_posterior_step as separate launches (cross-check, A/B)
    return !e || atoi(e) != 0;
  }();
  const bool fused_post = !ddim && !noise && fuse_noise;
  for (int i = first_step; i >= last_step; --i) {
    // (the fused posterior kernel of step i + 1 has written this step's model timesteps already)
    if (!fused_post || i == first_step) CHECK(gpt_fill_i64(t_model, B, host_timestep_map[i], stream));
    // x_prev and the losses are the loop's constants: their tokens are encoded in the first step only
    CHECK(denoiser_forward(m, ws, x, t_model, loss_target, loss_prev, loss_is_f64, x_prev, model_out, B, i != first_step, stream));
    if (fused_post) {
      // DDPM with in-kernel noise: Philox + posterior update in one kernel, the timestep as a scalar
      CHECK(gpt_posterior_step_philox(model_out, x, i, coef, predict_xstart, use_clamp, lo, hi, seed, (uint64_t)i, x, B, m->D,
                                      i > last_step ? t_model : nullptr, i > last_step ? host_timestep_map[i - 1] : 0, stream));
      continue;
    }
    CHECK(gpt_fill_i64(t_index, B, i, stream));
    const float* nz = nullptr;
    if (need_noise) {
      nz = noise ? noise + (int64_t)i * BD : noise_buf;
      if (!noise) CHECK(gpt_randn(noise_buf, BD, seed, (uint64_t)i, stream));
    }
    if (ddim)
      CHECK(gpt_ddim_step(model_out, x, nz, t_index, coef, alphas_cumprod, predict_xstart, use_clamp, lo, hi, eta, x, nullptr,
                          B, m->D, stream));
    else
      CHECK(gpt_posterior_step(model_out, x, nz, t_index, coef, predict_xstart, use_clamp, lo, hi, x, nullptr, B, m->D, stream));
  }
  return 0;
}

extern "C" int gpt_sample_loop(const gpt_model_t* m, const gpt_workspace_t* ws, float* x, const void* loss_target,
                               const void* loss_prev, int loss_is_f64, const float* x_prev, const float* coef,
                               const int64_t* host_timestep_map, int n_steps, int first_step, int last_step,
                               int predict_xstart, int use_clamp, float lo, float hi, const float* noise, uint64_t seed,
                               int64_t* t_model, int64_t* t_index, float* model_out, float* noise_buf, int B,
                               void* stream) {
  return sample_loop("gpt_sample_loop", 0, 0.f, nullptr, m, ws, x, loss_target, loss_prev, loss_is_f64, x_prev, coef,
                     host_timestep_map, n_steps, first_step, last_step, predict_xstart, use_clamp, lo, hi, noise, seed, t_model,
                     t_index, model_out, noise_buf, B, stream);
}

extern "C" int gpt_ddim_sample_loop(const gpt_model_t* m, const gpt_workspace_t* ws, float* x, const void* loss_target,
                                    const void* loss_prev, int loss_is_f64, const float* x_prev, const float* coef,
                                    const float* alphas_cumprod, float eta, const int64_t* host_timestep_map, int n_steps,
                                    int first_step, int last_step, int predict_xstart, int use_clamp, float lo, float hi,
                                    const float* noise, uint64_t seed, int64_t* t_model, int64_t* t_index, float* model_out,
                                    float* noise_buf, int B, void* stream) {
  return sample_loop("gpt_ddim_sample_loop", 1, eta, alphas_cumprod, m, ws, x, loss_target, loss_prev, loss_is_f64, x_prev, coef,
                     host_timestep_map, n_steps, first_step, last_step, predict_xstart, use_clamp, lo, hi, noise, seed, t_model,
                     t_index, model_out, noise_buf, B, stream);
}

extern "C" int gpt_denoiser_backward(const gpt_model_t* m, const gpt_workspace_t* ws, const gpt_backward_t* bw,
                                     const float* x, const float* x_prev, const float* d_out, int B, void* stream) {
  GPT_REQUIRE(m, "gpt_denoiser_backward: null model");
  return gpt_denoiser_backward_stages(m, ws, bw, x, x_prev, d_out, B, 0, m->n_layer + 2, stream);
}

extern "C" int gpt_denoiser_backward_stages(const gpt_model_t* m, const gpt_workspace_t* ws, const gpt_backward_t* bw,
                                            const float* x, const float* x_prev, const float* d_out, int B, int stage_begin,
                                            int stage_end, void* stream) {
  CHECK(validate(m, ws, B, "gpt_denoiser_backward"));
  GPT_REQUIRE(0 <= stage_begin && stage_begin < stage_end && stage_end <= m->n_layer + 2,
              "gpt_denoiser_backward_stages: bad stage range [%d, %d) of %d", stage_begin, stage_end, m->n_layer + 2);
  GPT_REQUIRE(bw && bw->layers && x && x_prev && d_out, "gpt_denoiser_backward: null operand");
  GPT_REQUIRE(ws->layer_stride == 1, "gpt_denoiser_backward: needs the workspace of a training forward");
  const int d = m->d, T = m->T, M = B * T, L = m->n_layer, dt = m->dtype;
  const size_t es = esize(dt);
  const size_t Mmax = (size_t)ws->B_max * T;
  const int64_t Mp = ((int64_t)M + 63) / 64 * 64;
  const int W = 2 * m->nfreq + 1;
  hipStream_t hs = (hipStream_t)stream;
  const int mma16 = dt == GPT_BF16;   // the ragged encoder / decoder gradients on the bf16 matrix pipe in bf16 compute mode
  auto xbuf = [&](int idx) { return ws->x + (size_t)idx * Mmax * d; };
  auto lay = [&](void* base, int l, size_t width) { return (void*)((char*)base + (size_t)l * Mmax * width * es); };
  auto stat = [&](int idx, int which) { return ws->stats + ((size_t)idx * 2 + which) * Mmax; };
  // dW[n_out, k_in] += dY^T . X through the NT kernel: both operands transposed to [*, Mp] (K = rows, zero padded)
  static const bool use_tn = [] {
    const char* e = getenv("GPT_WGRAD_TN");   // 0: always transpose + NT kernel (cross-check of the TN kernel)
    return !e || atoi(e) != 0;
  }();
  static const bool fuse_b1 = [] {
    const char* e = getenv("GPT_DGELU_COLSUM");   // 0: mlp.0's bias gradient by a separate gpt_colsum launch (cross-check, A/B)
    return !e || atoi(e) != 0;
  }();
  static const bool fuse_col = [] {
    const char* e = getenv("GPT_LNB_COLSUM");   // 0: bias gradients of mlp.2 / attn.proj by separate gpt_colsum launches (cross-check, A/B)
    return !e || atoi(e) != 0;
  }();
  // LayerNorm backward: per-workgroup sums into bw->ln_partials where there is room (one reduce launch at the end of this call),
  // float atomics otherwise
  LnPartCalls ln_calls;
  const int64_t ln_slot = 256 * 3 * (int64_t)d;                       // floats per launch
  static const bool ln_part_on = [] {
    const char* e = getenv("GPT_LNB_PARTIALS");   // 0: float atomics (cross-check, A/B)
    return !e || atoi(e) != 0;
  }();
  auto ln_bwd = [&](const void* dy, int dy_dt, const float* xin, const float* mean, const float* rstd, const float* gamma,
                    const float* resid, float* gout, void* gout_t, int gout_t_dt, float* dgamma, float* dbeta, float* gcol,
                    int rows) -> int {
    float* part = nullptr;
    if (ln_part_on && bw->ln_partials && ln_calls.n < GPT_LN_PART_MAX_CALLS &&
        (ln_calls.n + 1) * ln_slot * (int64_t)sizeof(float) <= bw->ln_partials_bytes)
      part = bw->ln_partials + ln_calls.n * ln_slot;
    int nb = 0;
    CHECK(gpt_layernorm_bwd_part(dy, dy_dt, xin, mean, rstd, gamma, resid, gout, gout_t, gout_t_dt, dgamma, dbeta, gcol, part, &nb, rows, d,
                                 stream));
    if (part) {
      auto& c = ln_calls.c[ln_calls.n++];
      c.part = part; c.out[0] = dgamma; c.out[1] = dbeta; c.out[2] = gcol; c.nblocks = nb;
    }
    return 0;
  };
  auto wgrad = [&](const void* dY, int n_out, const void* X, int k_in, float* gW) -> int {
    // the TN kernel cuts the row range into spans when the weight has few 256 x 256 tiles (gemm256_tn.hip)
    if (use_tn && dt == GPT_BF16 && n_out % 256 == 0 && k_in % 256 == 0)
      return gpt_gemm_tn(dY, n_out, X, k_in, gW, k_in, n_out, k_in, M, 1, stream);
    CHECK(gpt_transpose(dt, dY, bw->tA, M, n_out, Mp, stream));
    CHECK(gpt_transpose(dt, X, bw->tB, M, k_in, Mp, stream));
    return gpt_gemm_nt(dt, bw->tA, Mp, bw->tB, Mp, nullptr, gW, k_in, nullptr, 0, n_out, k_in, (int)Mp, GPT_EPI_ACCUM_F32, stream);
  };

  // stage 0 - decoder: d_tok[b, j, :] = d_out[b, off_j..] . W_j (zero for the tokens that are not decoded);  dW_j += d_out_j^T . hf_j
  // pruned last Block (prune_last_train): everything up to that Block's attention backward runs on the compact rows [B * To, d]
  const bool pruned = prune_last_train(m, B);
  const int To = m->n_out_tok, Mc = pruned ? B * To : M, Tc = pruned ? To : T;
  if (stage_begin == 0) {
  if (hipMemsetAsync(bw->d_tok, 0, (size_t)Mc * d * sizeof(float), hs) != hipSuccess) { gpt_set_error("memset failed"); return -1; }
  {
    const void *pa[GPT_MAX_TOKENS], *pw[GPT_MAX_TOKENS], *ph[GPT_MAX_TOKENS];
    float *pt[GPT_MAX_TOKENS], *pg[GPT_MAX_TOKENS];
    int vB[GPT_MAX_TOKENS], vd[GPT_MAX_TOKENS], vn[GPT_MAX_TOKENS], vTd[GPT_MAX_TOKENS];
    for (int j = 0; j < m->n_out_tok; ++j) {
      pa[j] = d_out + m->dec_off[j];
      pw[j] = m->dec_w[j];
      ph[j] = (const char*)ws->hf + (size_t)j * d * es;
      pt[j] = bw->d_tok + (size_t)j * d;
      pg[j] = bw->g_dec_w[j];
      vB[j] = B; vd[j] = d; vn[j] = m->dec_n[j]; vTd[j] = Tc * d;
    }
    // d_tok[b, j, :] = d_out[b, off_j ..] . W_j            (M = B, N = d, K = n_j)
    CHECK(gpt_small_gemm_batched_mode(m->n_out_tok, GPT_F32, pa, m->D, 1, dt, pw, d, 1, pt, vTd, vB, vd, vn, 0, mma16, stream));
    // dW_j[n, :] += sum_b d_out[b, off_j + n] hf[b, j, :]   (M = n_j, N = d, K = B)
    CHECK(gpt_small_gemm_batched_mode(m->n_out_tok, GPT_F32, pa, 1, m->D, dt, ph, (int64_t)Tc * d, 1, pg, vd, vn, vd, vB, 1, mma16, stream));
  }
  // bw->g / bw->gt = gradient of the residual stream behind the last Block = dY of its mlp.2: the bias gradient g_b2 is its column sum,
  // taken by the LayerNorm backward that produces it (and so for every Block below: ln1's backward of Block l + 1 feeds Block l's mlp.2,
  // ln2's backward feeds the attention projection of its own Block)
  CHECK(ln_bwd(bw->d_tok, GPT_F32, xbuf(2 * L), stat(2 * L + 1, 0), stat(2 * L + 1, 1), m->ln_f_w, nullptr, bw->g, bw->gt,
               dt, bw->g_ln_f_w, bw->g_ln_f_b, fuse_col && L > 0 ? bw->layers[L - 1].g_b2 : nullptr, Mc));
  }

  // stages 1 .. L - the blocks, last to first (stage s = layer L - s)
  // When the TN kernel takes all four weight gradients of a Block they are launched TOGETHER (gpt_gemm_tn_grouped) once the last
  // of their operands (d_qkv) exists: 432 tiles at d = 1536 = two nearly full rounds of whole contractions instead of four
  // launches of 36 ... 144 tiles.  Until then mlp.2's dY must survive the LayerNorm backward that would overwrite it: that
  // LayerNorm writes its operand copy to bw->tA (the NT fallback's transpose scratch, [4d, Mp] >= [M, d], unused on this path).
  const bool grouped = use_tn && dt == GPT_BF16 && d % 256 == 0;
  GPT_REQUIRE(!bw->wgrad_overwrite || grouped,
              "gpt_denoiser_backward: wgrad_overwrite needs the grouped TN launch (bf16, n_embd %% 256 == 0)");
  // Deferred weight gradients (gpt_backward_t.wgrad_defer): Block l works in slot (L - 1 - l) % G of the dY buffers and its four
  // TN problems wait in `pend` until G Blocks are collected (or the call ends): one grid for all of them.
  GPT_REQUIRE(!bw->wgrad_sumsq || bw->wgrad_overwrite, "gpt_denoiser_backward: wgrad_sumsq needs wgrad_overwrite");
  GPT_REQUIRE(bw->wgrad_defer >= 0 && bw->wgrad_defer <= 4, "gpt_denoiser_backward: wgrad_defer %d out of range (0..4)", bw->wgrad_defer);
  const int G = grouped && bw->wgrad_defer > 1 ? bw->wgrad_defer : 1;
  auto slot = [&](void* base, int sl, size_t width) { return (void*)((char*)base + (size_t)sl * M * width * es); };
  struct {
    const void *P[16], *Q[16];
    float* C[16];
    int I[16], J[16], R[16];
    int64_t ldp[16], ldq[16], ldc[16];
    int n = 0;
  } pend;
  auto flush = [&]() -> int {
    if (pend.n == 0) return 0;
    const int rc = gpt_tn_grouped_sumsq_r(pend.n, pend.P, pend.ldp, pend.Q, pend.ldq, pend.C, pend.ldc, pend.I, pend.J, pend.R,
                                          bw->wgrad_overwrite ? 0 : 1, bw->wgrad_sumsq, stream);
    pend.n = 0;
    return rc;
  };
  for (int l = L - 1; l >= 0; --l) {
    if (L - l < stage_begin || L - l >= stage_end) continue;
    const gpt_layer_t& ly = m->layers[l];
    const gpt_layer_grad_t& gl = bw->layers[l];
    const int sl = (L - 1 - l) % G;
    void* const gt_l = slot(bw->gt, sl, d);                 // dY of this Block's mlp.2 (written by the LayerNorm backward before it)
    void* const gt_next = slot(bw->gt, (sl + 1) % G, d);    // ... of the Block below
    void* const d_act_l = slot(bw->d_act, sl, 4 * (size_t)d);
    void* const d_qkv_l = slot(bw->d_qkv, sl, 3 * (size_t)d);
    const void* h1 = lay(ws->h1, l, d);
    const void* qkv = lay(ws->qkv, l, 3 * (size_t)d);
    const void* att = lay(ws->att, l, d);
    const void* h2 = lay(ws->h2, l, d);
    const void* act = lay(ws->act, l, 4 * (size_t)d);
    void* pre = lay(ws->pre, l, 4 * (size_t)d);
    void* gt2 = grouped ? slot(bw->tA, sl, d) : bw->gt;  // dY of the attention projection ([4d, Mp] holds four [M, d] slots)
    // rows of this Block's projection / MLP half: the decoded tokens' compact rows in the pruned last Block, else all
    const bool cmp = pruned && l == L - 1;
    const int Ml = cmp ? Mc : M;
    // mlp.2
    if (!fuse_col) CHECK(gpt_colsum(dt, gt_l, d, gl.g_b2, Ml, d, stream));
    if (!grouped) CHECK(wgrad(gt_l, d, act, 4 * d, gl.g_w2));
    // ... and mlp.0's bias gradient = the column sums of that output, taken in the same epilogue
    CHECK(gpt_gemm_nt_colsum(dt, gt_l, d, gl.w2_t, d, nullptr, d_act_l, 4 * d, pre, 4 * d, Ml, 4 * d, d, GPT_EPI_DGELU,
                             fuse_b1 ? gl.g_b1 : nullptr, stream));
    // mlp.0
    if (!fuse_b1) CHECK(gpt_colsum(dt, d_act_l, 4 * d, gl.g_b1, Ml, 4 * d, stream));
    if (!grouped) CHECK(wgrad(d_act_l, 4 * d, h2, d, gl.g_w1));
    CHECK(gpt_gemm_nt(dt, d_act_l, 4 * d, gl.w1_t, 4 * d, nullptr, bw->d_h, d, nullptr, 0, Ml, d, 4 * d, GPT_EPI_STORE, stream));
    // (compact rows: the residual gradient leaves this LayerNorm backward in bw->d_tok, free between stage 0 and stage L + 1, and
    // is scattered back to all rows below)
    CHECK(ln_bwd(bw->d_h, dt, xbuf(2 * l + 1), stat(2 + 2 * l, 0), stat(2 + 2 * l, 1), ly.ln2_w, bw->g, cmp ? bw->d_tok : bw->g, gt2, dt,
                 gl.g_ln2_w, gl.g_ln2_b, fuse_col ? gl.g_bproj : nullptr, Ml));
    // attention output projection
    if (!fuse_col) CHECK(gpt_colsum(dt, gt2, d, gl.g_bproj, Ml, d, stream));
    if (!grouped) CHECK(wgrad(gt2, d, att, d, gl.g_wproj));
    CHECK(gpt_gemm_nt(dt, gt2, d, gl.wproj_t, d, nullptr, bw->d_h, d, nullptr, 0, Ml, d, d, GPT_EPI_STORE, stream));
    const void* d_att = bw->d_h;
    if (cmp) {
      // back to all rows: the attention mixes tokens.  Gradient of the attention output and of the residual stream are zero in
      // the rows of the tokens the decoder does not read (bw->tB: the NT fallback's transpose scratch, unused on this route)
      CHECK(gpt_scatter_token_rows(bw->d_h, bw->tB, B, T, To, d, (int)es, stream));
      CHECK(gpt_scatter_token_rows(bw->d_tok, bw->g, B, T, To, d, 4, stream));
      d_att = bw->tB;
    }
    CHECK(gpt_attention_bwd(dt, qkv, d_att, d_qkv_l, B, T, m->n_head, d / m->n_head, stream));
    // query | key | value
    CHECK(gpt_colsum(dt, d_qkv_l, 3 * d, gl.g_bqkv, M, 3 * d, stream));
    if (grouped) {
      // mlp.2, mlp.0 first: they are the whole first round (2 x 144 tiles at d = 1536)
      const void* P[4] = {gt_l, d_act_l, d_qkv_l, gt2};
      const void* Q[4] = {act, h2, h1, att};
      float* Cg[4] = {gl.g_w2, gl.g_w1, gl.g_wqkv, gl.g_wproj};
      const int I[4] = {d, 4 * d, 3 * d, d}, J[4] = {4 * d, d, d, d};
      const int Rk[4] = {Ml, Ml, M, Ml};                      // contraction lengths: only q|k|v sees all rows in the pruned Block
      for (int k = 0; k < 4; ++k) {
        const int n = pend.n++;
        pend.P[n] = P[k]; pend.Q[n] = Q[k]; pend.C[n] = Cg[k]; pend.I[n] = I[k]; pend.J[n] = J[k]; pend.R[n] = Rk[k];
        pend.ldp[n] = I[k]; pend.ldq[n] = J[k]; pend.ldc[n] = J[k];
      }
      if (pend.n == 4 * G) CHECK(flush());
    } else {
      CHECK(wgrad(d_qkv_l, 3 * d, h1, d, gl.g_wqkv));
    }
    CHECK(gpt_gemm_nt(dt, d_qkv_l, 3 * d, gl.wqkv_t, 3 * d, nullptr, bw->d_h, d, nullptr, 0, M, d, 3 * d, GPT_EPI_STORE, stream));
    CHECK(ln_bwd(bw->d_h, dt, xbuf(2 * l), stat(1 + 2 * l, 0), stat(1 + 2 * l, 1), ly.ln1_w, bw->g, bw->g, gt_next, dt,
                 gl.g_ln1_w, gl.g_ln1_b, fuse_col && l > 0 ? bw->layers[l - 1].g_b2 : nullptr, M));
  }
  CHECK(flush());                                        // what this call collected (gradients are complete when it returns)
  if (stage_end < L + 2) return gpt_ln_partials_reduce(ln_calls, d, stream);
  // stage L + 1 - position embedding: sum over the batch of the gradient of x0 = ln_in(enc) + pos_emb
  CHECK(gpt_colsum(GPT_F32, bw->g, (int64_t)T * d, bw->g_pos_emb, B, T * d, stream));
  CHECK(ln_bwd(bw->g, GPT_F32, ws->enc, stat(0, 0), stat(0, 1), m->ln_in_w, nullptr, bw->d_tok, nullptr, GPT_F32,
               bw->g_ln_in_w, bw->g_ln_in_b, nullptr, M));
  CHECK(gpt_ln_partials_reduce(ln_calls, d, stream));
  {
    // dW_i[:, k] += sum_b d_tok[b, i, :] src_i[b, k]        (M = d, N = k_i, K = B); sources grouped by their row stride
    const int64_t ld_scal = (int64_t)m->n_scalar * W;
    for (int pass = 0; pass < 2; ++pass) {
      const void *pa[GPT_MAX_TOKENS], *pb[GPT_MAX_TOKENS];
      float* pc[GPT_MAX_TOKENS];
      int vM[GPT_MAX_TOKENS], vN[GPT_MAX_TOKENS], vK[GPT_MAX_TOKENS], vld[GPT_MAX_TOKENS];
      int n = 0;
      for (int i = 0; i < T; ++i) {
        const bool scalar = m->enc_src[i] == GPT_SRC_SCALAR;
        if (scalar != (pass == 1)) continue;
        pa[n] = bw->d_tok + (size_t)i * d;
        pb[n] = scalar ? ws->scal + m->enc_off[i] : (m->enc_src[i] == GPT_SRC_X ? x : x_prev) + m->enc_off[i];
        pc[n] = bw->g_enc_w[i];
        vM[n] = d; vN[n] = m->enc_k[i]; vK[n] = B; vld[n] = m->enc_k[i];
        ++n;
      }
      if (n) CHECK(gpt_small_gemm_batched_mode(n, GPT_F32, pa, 1, (int64_t)T * d, GPT_F32, pb, pass ? ld_scal : (int64_t)m->D, 1, pc, vld,
                                               vM, vN, vK, 1, mma16, stream));
    }
  }
  return 0;
}
