// Fused DDPM elementwise kernels (HBM-bound) + Philox normal generator + dtype/transposition helpers.
//   q_sample                     Gpt/diffusion/gaussian_diffusion.py:209-224   12*D bytes + 4*D written per sample
//   posterior step (p_sample)    :289-317 + :404-410                           16*D read + 4*D written per sample-step
//   mean_flat((target-out)^2)    diffusion_utils.py:8-12, gaussian_diffusion.py:773
// The reference materialises four to five (B,D) broadcast tensors per step through _extract_into_tensor
// (:855-867); here the per-timestep scalars come from one small device table indexed by t[b].
#include "common.h"

// The reference rounds every product and sum separately (separate ATen kernels); hipcc contracts a*b+c into an
// FMA by default, __fmul_rn/__fadd_rn included, which changes last bits.  Keep IEEE op-by-op rounding in this file.
#pragma clang fp contract(off)

namespace {

constexpr int NC = GPT_DDPM_NCOEF;

template <int V>
struct vec {
  float v[V];
};
template <int V>
__device__ __forceinline__ vec<V> vload(const float* p) {
  vec<V> r;
  if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
  else if constexpr (V == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
  else { r.v[0] = *p; }
  return r;
}
template <int V>
__device__ __forceinline__ void vstore(float* p, const vec<V>& r) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  else if constexpr (V == 2) *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
  else *p = r.v[0];
}

// D = V * DV: rows of CIFAR-10 CNN vectors (D = 5370) are only 8-byte aligned, MNIST (7960) 16-byte
template <int V>
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const int64_t* __restrict__ t, const float* __restrict__ coef,
                                                       float* __restrict__ xt, int B, int DV) {
  const int b = blockIdx.y;
  const float a = coef[t[b] * NC + GPT_C_SQRT_AC], s = coef[t[b] * NC + GPT_C_SQRT_1MAC];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < DV; i += gridDim.x * blockDim.x) {
    const int64_t o = ((int64_t)b * DV + i) * V;
    const vec<V> x = vload<V>(x0 + o), n = vload<V>(noise + o);
    vec<V> r;
    // two roundings per term then one add, as the reference's a*x0 + s*noise (no FMA contraction)
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = a * x.v[e] + s * n.v[e];
    vstore<V>(xt + o, r);
  }
}

struct PostScal {
  float c1, c2, sigma, recip, recipm1;
};

__device__ __forceinline__ float post_one(float mo, float xt, float nz, const PostScal& p, int predict_xstart,
                                          int use_clamp, float lo, float hi, float* x0_out) {
  float x0 = predict_xstart ? mo : p.recip * xt - p.recipm1 * mo;
  if (use_clamp) x0 = fminf(fmaxf(x0, lo), hi);
  *x0_out = x0;
  const float mean = p.c1 * x0 + p.c2 * xt;
  return mean + p.sigma * nz;
}

// Philox4x32-10 -> four N(0,1) by Box-Muller: block q of stream (seed, sid) = elements [4 q, 4 q + 4) of gpt_randn's output
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox_normals4(int64_t q, uint64_t seed, uint64_t sid, float (&z)[4]) {
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    z[2 * h] = rad * cs;
    z[2 * h + 1] = rad * sn;
  }
}

template <int V>
__global__ __launch_bounds__(256) void posterior_kernel(const float* __restrict__ mo, const float* __restrict__ xt,
                                                        const float* __restrict__ noise, const int64_t* __restrict__ t,
                                                        const float* __restrict__ coef, int predict_xstart, int use_clamp,
                                                        float lo, float hi, float* __restrict__ xout,
                                                        float* __restrict__ x0out, int B, int DV) {
  const int b = blockIdx.y;
  const int64_t tb = t[b];
  PostScal p;
  p.c1 = coef[tb * NC + GPT_C_POST1];
  p.c2 = coef[tb * NC + GPT_C_POST2];
  // (t != 0) * exp(0.5 * logvar): the mask multiplies the already-rounded exp, as the reference does
  p.sigma = tb != 0 ? expf(0.5f * coef[tb * NC + GPT_C_LOGVAR]) : 0.f;
  p.recip = coef[tb * NC + GPT_C_RECIP];
  p.recipm1 = coef[tb * NC + GPT_C_RECIPM1];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < DV; i += gridDim.x * blockDim.x) {
    const int64_t o = ((int64_t)b * DV + i) * V;
    const vec<V> m = vload<V>(mo + o), x = vload<V>(xt + o);
    vec<V> n;
    if (noise) n = vload<V>(noise + o);
    else {
#pragma unroll
      for (int e = 0; e < V; ++e) n.v[e] = 0.f;
    }
    vec<V> r, z;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = post_one(m.v[e], x.v[e], n.v[e], p, predict_xstart, use_clamp, lo, hi, &z.v[e]);
    vstore<V>(xout + o, r);
    if (x0out) vstore<V>(x0out + o, z);
  }
}

// The sampling loop's step: every row is at the same timestep and the noise is drawn in the kernel - the Philox block of flat
// element index e is e >> 2 of stream (seed, sid), i.e. exactly what gpt_randn would have written to a [B, D] buffer, so the
// result equals gpt_randn + gpt_posterior_step bit for bit, without the 2 x 4 B D bytes of the noise round trip, the randn
// launch and the timestep fill.  V = 2 / 1 (D % 4 != 0) recompute the block for every V elements.
template <int V>
__global__ __launch_bounds__(256) void posterior_philox_kernel(const float* __restrict__ mo, const float* __restrict__ xt, int64_t tb,
                                                               const float* __restrict__ coef, int predict_xstart, int use_clamp,
                                                               float lo, float hi, uint64_t seed, uint64_t sid,
                                                               float* __restrict__ xout, int B, int DV, int64_t* __restrict__ t_next,
                                                               int64_t t_next_value) {
  const int b = blockIdx.y;
  // the NEXT step's model timestep for this row (the loop's forward reads it from a [B] array): saves the loop a fill launch
  if (t_next && blockIdx.x == 0 && threadIdx.x == 0) t_next[b] = t_next_value;
  PostScal p;
  p.c1 = coef[tb * NC + GPT_C_POST1];
  p.c2 = coef[tb * NC + GPT_C_POST2];
  p.sigma = tb != 0 ? expf(0.5f * coef[tb * NC + GPT_C_LOGVAR]) : 0.f;
  p.recip = coef[tb * NC + GPT_C_RECIP];
  p.recipm1 = coef[tb * NC + GPT_C_RECIPM1];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < DV; i += gridDim.x * blockDim.x) {
    const int64_t o = ((int64_t)b * DV + i) * V;
    const vec<V> m = vload<V>(mo + o), x = vload<V>(xt + o);
    float z[4];
    philox_normals4(o >> 2, seed, sid, z);
    vec<V> r;
    float x0;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = post_one(m.v[e], x.v[e], z[(V == 4 ? 0 : (int)(o & 3)) + e], p, predict_xstart, use_clamp, lo, hi, &x0);
    vstore<V>(xout + o, r);
  }
}

// DDIM step (gaussian_diffusion.py:507-552).  Every scalar is formed in fp32 from the fp32-rounded tables with the
// reference's own operation order (sqrt and division are IEEE here; contraction is off in this file), so the update is
// bit-exact against the same arithmetic on the CPU:
//   eps    = (recip[t] x_t - x0) / recipm1[t]
//   sigma  = eta * sqrt((1 - abp) / (1 - ab)) * sqrt(1 - ab / abp)         ab = alphas_cumprod[t], abp = alphas_cumprod[t - 1] | 1
//   x_prev = x0 sqrt(abp) + sqrt(1 - abp - sigma^2) eps + (t != 0) sigma noise
struct DdimScal {
  float recip, recipm1, sq_abp, dir, sigma, mask;
};

template <int V>
__global__ __launch_bounds__(256) void ddim_kernel(const float* __restrict__ mo, const float* __restrict__ xt,
                                                   const float* __restrict__ noise, const int64_t* __restrict__ t,
                                                   const float* __restrict__ coef, const float* __restrict__ acp,
                                                   int predict_xstart, int use_clamp, float lo, float hi, float eta,
                                                   float* __restrict__ xout, float* __restrict__ x0out, int B, int DV,
                                                   int reverse_steps) {
  const int b = blockIdx.y;
  const int64_t tb = t[b];
  DdimScal p;
  p.recip = coef[tb * NC + GPT_C_RECIP];
  p.recipm1 = coef[tb * NC + GPT_C_RECIPM1];
  // reverse_steps > 0: the deterministic reverse ODE (ddim_reverse_sample, gaussian_diffusion.py:559-592) = the same
  // update towards alphas_cumprod_next (0 after the last step) with sigma = 0
  const float ab = acp[tb];
  const float abp = reverse_steps > 0 ? (tb + 1 < reverse_steps ? acp[tb + 1] : 0.0f) : (tb > 0 ? acp[tb - 1] : 1.0f);
  p.sigma = reverse_steps > 0 ? 0.0f : (eta * sqrtf((1.0f - abp) / (1.0f - ab))) * sqrtf(1.0f - ab / abp);
  p.sq_abp = sqrtf(abp);
  p.dir = sqrtf((1.0f - abp) - p.sigma * p.sigma);
  p.mask = tb != 0 ? 1.0f : 0.0f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < DV; i += gridDim.x * blockDim.x) {
    const int64_t o = ((int64_t)b * DV + i) * V;
    const vec<V> m = vload<V>(mo + o), x = vload<V>(xt + o);
    vec<V> n;
    if (noise) n = vload<V>(noise + o);
    else {
#pragma unroll
      for (int e = 0; e < V; ++e) n.v[e] = 0.f;
    }
    vec<V> r, z;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float x0 = predict_xstart ? m.v[e] : p.recip * x.v[e] - p.recipm1 * m.v[e];
      if (use_clamp) x0 = fminf(fmaxf(x0, lo), hi);
      z.v[e] = x0;
      const float eps = (p.recip * x.v[e] - x0) / p.recipm1;
      const float mean = x0 * p.sq_abp + p.dir * eps;
      r.v[e] = mean + (p.mask * p.sigma) * n.v[e];
    }
    vstore<V>(xout + o, r);
    if (x0out) vstore<V>(x0out + o, z);
  }
}

// one 256-thread block per sample row
__global__ __launch_bounds__(256) void mse_rows_kernel(const float* __restrict__ target, const float* __restrict__ out,
                                                       float* __restrict__ mse, int D) {
  __shared__ float part[4];
  const int b = blockIdx.x;
  const float* tr = target + (int64_t)b * D;
  const float* orow = out + (int64_t)b * D;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float e = tr[i] - orow[i];
    s += e * e;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) mse[b] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)D;
}

// _prior_bpd (gaussian_diffusion.py:783-797): KL(q(x_T | x_0) || N(0, I)) per row in bits per dimension, with normal_kl's
// term order (diffusion_utils.py:35-41: 0.5 * (-1 + logvar2 - logvar1 + exp(logvar1 - logvar2) + (mean1 - mean2)^2 *
// exp(-logvar2)), mean2 = logvar2 = 0).  One 256-thread block per row.
__global__ __launch_bounds__(256) void prior_bpd_kernel(const float* __restrict__ x, float s, float lv1, float* __restrict__ out,
                                                        int D) {
  __shared__ float part[4];
  const float* row = x + (int64_t)blockIdx.x * D;
  const float base = ((-1.0f + 0.0f) - lv1) + expf(lv1 - 0.0f);
  float acc = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float m = s * row[i];
    acc += 0.5f * (base + (m * m) * 1.0f);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (((part[0] + part[1]) + (part[2] + part[3])) / (float)D) / 0.6931471805599453f;
}

__global__ __launch_bounds__(256) void mse_rows_bwd_kernel(const float* __restrict__ target, const float* __restrict__ out,
                                                           const float* __restrict__ scale, float* __restrict__ grad, int D) {
  const int b = blockIdx.y;
  const float c = 2.0f * scale[b] / (float)D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) {
    const int64_t o = (int64_t)b * D + i;
    grad[o] = c * (out[o] - target[o]);
  }
}

// ---------------------------------------------------------------- Philox4x32-10 -> N(0,1) by Box-Muller
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t sid) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one Philox block -> 4 normals
  if (q * 4 >= n) return;
  float z[4];
  philox_normals4(q, seed, sid, z);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (q * 4 + e < n) out[q * 4 + e] = z[e];
}

__global__ __launch_bounds__(256) void fill_i64_kernel(int64_t* __restrict__ out, int64_t n, int64_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = v;
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = f32_to_bf16(src[i]);
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = bf16_to_f32(src[i]);
}

// 64 x 64 tile transpose through padded LDS; dst[c][r], r in [rows, ld_dst) zero-filled
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int rows, int cols,
                                                        int64_t ld_dst) {
  __shared__ T tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : (T)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < ld_dst && c < cols) dst[(int64_t)c * ld_dst + r] = tile[tx][i];
  }
}

// bf16, 16-byte global accesses on both sides (the kernel above moves 2 bytes per lane and instruction: 1.7 TB/s on the weight
// transposes of the training step): a 64 x 64 tile is scattered TRANSPOSED into LDS (pitch 66 elements: the 8 x 8 (chunk, row)
// lanes of a wave hit distinct banks), then read back as rows of the output.  Needs cols % 8 == 0, ld_dst % 8 == 0.
__global__ __launch_bounds__(256) void transpose16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int rows, int cols,
                                                          int64_t ld_dst) {
  constexpr int PT = 66;
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * PT];     // tile[c][r]
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int q = threadIdx.x & 7, row8 = threadIdx.x >> 3;          // 8 chunks of 8 elements x 32 rows per pass
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = row8 + 32 * i, c = c0 + 8 * q;
    u32x4 v = u32x4{0, 0, 0, 0};
    if (r0 + r < rows && c < cols) v = *reinterpret_cast<const u32x4*>(src + (int64_t)(r0 + r) * cols + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[(8 * q + k) * PT + r] = (bf16_t)(v[k >> 1] >> (16 * (k & 1)));
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = row8 + 32 * i, r = r0 + 8 * q;
    if (c0 + c < cols && r < ld_dst) {
      const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile + c * PT + 8 * q);   // (c * 66 + 8 q) * 2 bytes: 4-byte aligned
      *reinterpret_cast<u32x4*>(dst + (int64_t)(c0 + c) * ld_dst + r) = u32x4{t32[0], t32[1], t32[2], t32[3]};
    }
  }
}

inline int blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

static inline int vec_width(int D, const void* a, const void* b, const void* c, const void* d, const void* e) {
  const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e;
  if (D % 4 == 0 && m % 16 == 0) return 4;
  if (D % 2 == 0 && m % 8 == 0) return 2;
  return 1;
}

extern "C" int gpt_q_sample(const float* x0, const float* noise, const int64_t* t, const float* coef, float* x_t, int B,
                            int D, void* stream) {
  GPT_REQUIRE(x0 && noise && t && coef && x_t, "gpt_q_sample: null operand");
  GPT_REQUIRE(B > 0 && D > 0, "gpt_q_sample: empty problem (B=%d D=%d)", B, D);
  const int V = vec_width(D, x0, noise, x_t, nullptr, nullptr);
  const int gx = blocks_for(D / V) > 8 ? 8 : blocks_for(D / V);
  const dim3 grid(gx, B);
  hipStream_t s = (hipStream_t)stream;
  if (V == 4) hipLaunchKernelGGL(q_sample_kernel<4>, grid, dim3(256), 0, s, x0, noise, t, coef, x_t, B, D / 4);
  else if (V == 2) hipLaunchKernelGGL(q_sample_kernel<2>, grid, dim3(256), 0, s, x0, noise, t, coef, x_t, B, D / 2);
  else hipLaunchKernelGGL(q_sample_kernel<1>, grid, dim3(256), 0, s, x0, noise, t, coef, x_t, B, D);
  return gpt_launch_status("gpt_q_sample");
}

extern "C" int gpt_posterior_step(const float* model_out, const float* x_t, const float* noise, const int64_t* t,
                                  const float* coef, int predict_xstart, int use_clamp, float lo, float hi, float* x_out,
                                  float* pred_xstart, int B, int D, void* stream) {
  GPT_REQUIRE(model_out && x_t && t && coef && x_out, "gpt_posterior_step: null operand");
  GPT_REQUIRE(B > 0 && D > 0, "gpt_posterior_step: empty problem (B=%d D=%d)", B, D);
  const int V = vec_width(D, model_out, x_t, noise, x_out, pred_xstart);
  const int gx = blocks_for(D / V) > 8 ? 8 : blocks_for(D / V);
  const dim3 grid(gx, B);
  hipStream_t s = (hipStream_t)stream;
#define POST(VV)                                                                                                       \
  hipLaunchKernelGGL(posterior_kernel<VV>, grid, dim3(256), 0, s, model_out, x_t, noise, t, coef, predict_xstart,        \
                     use_clamp, lo, hi, x_out, pred_xstart, B, D / VV)
  if (V == 4) POST(4);
  else if (V == 2) POST(2);
  else POST(1);
#undef POST
  return gpt_launch_status("gpt_posterior_step");
}

int gpt_posterior_step_philox(const float* model_out, const float* x_t, int64_t t, const float* coef, int predict_xstart, int use_clamp,
                              float lo, float hi, uint64_t seed, uint64_t sid, float* x_out, int B, int D, int64_t* t_next,
                              int64_t t_next_value, void* stream) {
  GPT_REQUIRE(model_out && x_t && coef && x_out && t >= 0, "gpt_sample_loop: posterior step: null operand");
  GPT_REQUIRE(B > 0 && D > 0, "gpt_sample_loop: posterior step: empty problem (B=%d D=%d)", B, D);
  const int V = vec_width(D, model_out, x_t, x_out, nullptr, nullptr);
  const int gx = blocks_for(D / V) > 8 ? 8 : blocks_for(D / V);
  const dim3 grid(gx, B);
  hipStream_t s = (hipStream_t)stream;
#define POSTP(VV)                                                                                                          \
  hipLaunchKernelGGL(posterior_philox_kernel<VV>, grid, dim3(256), 0, s, model_out, x_t, t, coef, predict_xstart, use_clamp, lo, hi, \
                     seed, sid, x_out, B, D / VV, t_next, t_next_value)
  if (V == 4) POSTP(4);
  else if (V == 2) POSTP(2);
  else POSTP(1);
#undef POSTP
  return gpt_launch_status("gpt_sample_loop: posterior step");
}

extern "C" int gpt_ddim_step(const float* model_out, const float* x_t, const float* noise, const int64_t* t,
                             const float* coef, const float* alphas_cumprod, int predict_xstart, int use_clamp, float lo,
                             float hi, float eta, float* x_out, float* pred_xstart, int B, int D, void* stream) {
  GPT_REQUIRE(model_out && x_t && t && coef && alphas_cumprod && x_out, "gpt_ddim_step: null operand");
  GPT_REQUIRE(B > 0 && D > 0, "gpt_ddim_step: empty problem (B=%d D=%d)", B, D);
  GPT_REQUIRE(eta >= 0.f, "gpt_ddim_step: eta must be >= 0");
  GPT_REQUIRE(noise || eta == 0.f, "gpt_ddim_step: eta > 0 needs noise");
  const int V = vec_width(D, model_out, x_t, noise, x_out, pred_xstart);
  const int gx = blocks_for(D / V) > 8 ? 8 : blocks_for(D / V);
  const dim3 grid(gx, B);
  hipStream_t s = (hipStream_t)stream;
#define DDIM(VV)                                                                                                       \
  hipLaunchKernelGGL(ddim_kernel<VV>, grid, dim3(256), 0, s, model_out, x_t, noise, t, coef, alphas_cumprod,             \
                     predict_xstart, use_clamp, lo, hi, eta, x_out, pred_xstart, B, D / VV, 0)
  if (V == 4) DDIM(4);
  else if (V == 2) DDIM(2);
  else DDIM(1);
#undef DDIM
  return gpt_launch_status("gpt_ddim_step");
}

extern "C" int gpt_ddim_reverse_step(const float* model_out, const float* x_t, const int64_t* t, const float* coef,
                                     const float* alphas_cumprod, int n_steps, int predict_xstart, int use_clamp, float lo,
                                     float hi, float* x_out, float* pred_xstart, int B, int D, void* stream) {
  GPT_REQUIRE(model_out && x_t && t && coef && alphas_cumprod && x_out && B > 0 && D > 0 && n_steps > 0,
              "gpt_ddim_reverse_step: bad arguments");
  GPT_REQUIRE(B <= 65535, "gpt_ddim_reverse_step: B > 65535");
  const int V = vec_width(D, model_out, x_t, nullptr, x_out, pred_xstart);
  const int gx = blocks_for(D / V) > 8 ? 8 : blocks_for(D / V);
  const dim3 grid(gx, B);
  hipStream_t s = (hipStream_t)stream;
#define DDIMR(VV)                                                                                                      \
  hipLaunchKernelGGL(ddim_kernel<VV>, grid, dim3(256), 0, s, model_out, x_t, (const float*)nullptr, t, coef,          \
                     alphas_cumprod, predict_xstart, use_clamp, lo, hi, 0.0f, x_out, pred_xstart, B, D / VV, n_steps)
  if (V == 4) DDIMR(4);
  else if (V == 2) DDIMR(2);
  else DDIMR(1);
#undef DDIMR
  return gpt_launch_status("gpt_ddim_reverse_step");
}

extern "C" int gpt_prior_bpd(const float* x_start, float sqrt_ac_last, float log_one_minus_ac_last, float* out, int B, int D,
                             void* stream) {
  GPT_REQUIRE(x_start && out && B > 0 && D > 0, "gpt_prior_bpd: bad arguments");
  hipLaunchKernelGGL(prior_bpd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x_start, sqrt_ac_last, log_one_minus_ac_last,
                     out, D);
  return gpt_launch_status("gpt_prior_bpd");
}

extern "C" int gpt_mse_rows(const float* target, const float* out, float* mse, int B, int D, void* stream) {
  GPT_REQUIRE(target && out && mse && B > 0 && D > 0, "gpt_mse_rows: bad arguments");
  hipLaunchKernelGGL(mse_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, target, out, mse, D);
  return gpt_launch_status("gpt_mse_rows");
}

extern "C" int gpt_mse_rows_bwd(const float* target, const float* out, const float* scale, float* grad_out, int B, int D,
                                void* stream) {
  GPT_REQUIRE(target && out && scale && grad_out && B > 0 && D > 0, "gpt_mse_rows_bwd: bad arguments");
  const dim3 grid(blocks_for(D) > 8 ? 8 : blocks_for(D), B);
  hipLaunchKernelGGL(mse_rows_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, target, out, scale, grad_out, D);
  return gpt_launch_status("gpt_mse_rows_bwd");
}

extern "C" int gpt_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  GPT_REQUIRE(out && n > 0, "gpt_randn: bad arguments");
  const int64_t quads = (n + 3) / 4;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                     stream_id);
  return gpt_launch_status("gpt_randn");
}

extern "C" int gpt_fill_i64(int64_t* out, int64_t n, int64_t value, void* stream) {
  GPT_REQUIRE(out && n > 0, "gpt_fill_i64: bad arguments");
  hipLaunchKernelGGL(fill_i64_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, out, n, value);
  return gpt_launch_status("gpt_fill_i64");
}

extern "C" int gpt_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  GPT_REQUIRE(src && dst && n > 0, "gpt_f32_to_bf16: bad arguments");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  return gpt_launch_status("gpt_f32_to_bf16");
}

extern "C" int gpt_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  GPT_REQUIRE(src && dst && n > 0, "gpt_bf16_to_f32: bad arguments");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n);
  return gpt_launch_status("gpt_bf16_to_f32");
}

extern "C" int gpt_transpose(int dtype, const void* src, void* dst, int rows, int cols, int64_t ld_dst, void* stream) {
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_transpose: bad dtype %d", dtype);
  GPT_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_dst >= rows, "gpt_transpose: bad arguments");
  const dim3 grid((cols + 63) / 64, (int)((ld_dst + 63) / 64));
  if (dtype == GPT_BF16 && cols % 8 == 0 && ld_dst % 8 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0)
    hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, rows, cols, ld_dst);
  else if (dtype == GPT_BF16)
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, rows, cols, ld_dst);
  else
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, rows, cols, ld_dst);
  return gpt_launch_status("gpt_transpose");
}
