// Backward kernels of the denoiser that are not plain GEMMs (those reuse gpt_gemm_nt on transposed operands):
// LayerNorm backward, short-sequence attention backward, column sums (bias / position-embedding gradients)
// and a small strided fp32 GEMM for the ragged encoder / decoder gradients.
// They implement what autograd derives for Gpt/models/transformer.py:34-51 (attention), :69-72 (Block),
// :215-242 (encoder / decoder) in the reference; parity is pinned by the gradient fixtures g6_training_*.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------ LayerNorm backward
// One wave per row.  dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); the row's contribution to
// dgamma / dbeta is accumulated per workgroup in registers and flushed with one fp32 atomic per column per
// workgroup (gamma/beta gradients are sums over all B*T rows).
//   resid != NULL: gout = resid + dx (the residual-stream gradient passes the LayerNorm branch point)
//   gout_t != NULL: also emits gout in the compute dtype (operand of the next dgrad / wgrad GEMMs)
template <typename DyT, typename OutT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const DyT* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ resid,
                                                            float* __restrict__ gout, OutT* __restrict__ gout_t,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                            int d, int rows_per_block) {
  extern __shared__ float red[];  // [2][d] per-workgroup partial dgamma / dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < 2 * d; c += 256) red[c] = 0.f;
  __syncthreads();
  const int row_begin = blockIdx.x * rows_per_block;
  const int row_end = min(rows, row_begin + rows_per_block);
  for (int row = row_begin + wave; row < row_end; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + (int64_t)row * d;
    const DyT* dyr = dy + (int64_t)row * d;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float gdy = gamma[c] * elem<DyT>::load(dyr + c);
      const float xh = (xr[c] - mu) * rs;
      s1 += gdy;
      s2 += gdy * xh;
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
    for (int c = lane; c < d; c += 64) {
      const float dyv = elem<DyT>::load(dyr + c);
      const float xh = (xr[c] - mu) * rs;
      float dx = rs * (gamma[c] * dyv - s1 - xh * s2);
      if (resid) dx += resid[(int64_t)row * d + c];
      gout[(int64_t)row * d + c] = dx;
      if (gout_t) elem<OutT>::store(gout_t + (int64_t)row * d + c, dx);
      atomicAdd(&red[c], dyv * xh);       // LDS atomics: 4 waves share the partial
      atomicAdd(&red[d + c], dyv);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) {
    atomicAdd(dgamma + c, red[c]);
    atomicAdd(dbeta + c, red[d + c]);
  }
}

// ------------------------------------------------------------------------------------------ column sums
// out[c] += sum_r X[r, c]   (X: rows x cols, leading dimension ld)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t ld, float* __restrict__ out, int rows,
                                                     int cols, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += elem<T>::load(X + (int64_t)r * ld + c);
  atomicAdd(out + c, s);
}

// ------------------------------------------------------------------------------------------ small strided GEMM
// C[m, n] (+)= sum_k A(m, k) * B(k, n), fp32 FMA, arbitrary element strides: used for the ragged encoder /
// decoder gradients (0.4 % of the FLOPs; K = batch or a token width).  64 x 64 tile, 16-deep LDS chunks.
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void small_gemm_kernel(const TA* __restrict__ A, int64_t a_sm, int64_t a_sk,
                                                         const TB* __restrict__ B, int64_t b_sk, int64_t b_sn,
                                                         float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                         int accumulate) {
  __shared__ float As[16][65], Bs[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      // lay the loads out along whichever index is contiguous in memory
      int kk, mm;
      if (a_sk == 1) { kk = i & 15; mm = i >> 4; } else { mm = i & 63; kk = i >> 6; }
      const int m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < M && k < K) ? elem<TA>::load(A + m * a_sm + k * a_sk) : 0.f;
      int kb, nn;
      if (b_sk == 1) { kb = i & 15; nn = i >> 4; } else { nn = i & 63; kb = i >> 6; }
      const int n = n0 + nn, k2 = k0 + kb;
      Bs[kb][nn] = (n < N && k2 < K) ? elem<TB>::load(B + k2 * b_sk + n * b_sn) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float* c = C + (int64_t)m * ldc + n;
      *c = accumulate ? *c + acc[i][j] : acc[i][j];
    }
  }
}

// ------------------------------------------------------------------------------------------ attention backward
// One wave per (batch, head), T <= 32.  Both orientations of the score matrix are recomputed with MFMA so
// that every product that follows contracts over an accumulator's ROW index (no lane movement, see
// attention.hip):   P^T, dP^T (keys on rows) -> dS^T -> dQ^T = K^T . dS^T
//                   P,   dP   (queries on rows) -> dS  -> dK^T = Q^T . dS,   dV^T = dO^T . P
__device__ __forceinline__ int row_of(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

template <typename T, int HD>
__device__ __forceinline__ f32x16 rows_product(const T* __restrict__ a_rows, const T* __restrict__ b_rows, int64_t ld,
                                               int r, int hh, bool valid) {
  // acc[row = a index][col = b index] = sum_k a_rows[row][k] * b_rows[col][k]
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int64_t off = (int64_t)(valid ? r : 0) * ld;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      u32x4 a = *reinterpret_cast<const u32x4*>(a_rows + off + ks * 16 + hh * 8);
      u32x4 b = *reinterpret_cast<const u32x4*>(b_rows + off + ks * 16 + hh * 8);
      if (!valid) { a = u32x4{0, 0, 0, 0}; b = u32x4{0, 0, 0, 0}; }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < HD / 8; ++ks) {
      float4 a = *reinterpret_cast<const float4*>(a_rows + off + ks * 8 + hh * 4);
      float4 b = *reinterpret_cast<const float4*>(b_rows + off + ks * 8 + hh * 4);
      if (!valid) { a = make_float4(0, 0, 0, 0); b = make_float4(0, 0, 0, 0); }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  }
  return acc;
}

// out^T[dd][col] = scale * sum_row src[row][dd] * X[row][col]   (X = accumulator tile, contraction over its rows);
// written to dst[col][dd] (row-major [T][ld_out] slice of one head)
template <typename T, int HD>
__device__ __forceinline__ void cols_product_store(const T* __restrict__ src, int64_t ld, const f32x16& X, float scale,
                                                   T* __restrict__ dst, int64_t ld_out, int r, int hh, int Ttok) {
  constexpr int ND = HD / 32;
  bf16x8 xp[2];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) xp[s][j] = (short)f32_to_bf16(X[8 * s + j]);
  }
#pragma unroll
  for (int t = 0; t < ND; ++t) {
    f32x16 O;
#pragma unroll
    for (int e = 0; e < 16; ++e) O[e] = 0.f;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = row_of(8 * s + j, hh);
          v[j] = row < Ttok ? (short)src[(int64_t)row * ld + t * 32 + r] : (short)0;
        }
        O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, xp[s], O, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row_of(e, hh);
        const float v = row < Ttok ? src[(int64_t)row * ld + t * 32 + r] : 0.f;
        O = __builtin_amdgcn_mfma_f32_32x32x2f32(v, X[e], O, 0, 0, 0);
      }
    }
    if (r < Ttok) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        T* p = dst + (int64_t)r * ld_out + t * 32 + 8 * g + 4 * hh;
        if constexpr (sizeof(T) == 2) {
          uint2 o;
          o.x = pack_bf16x2(O[4 * g] * scale, O[4 * g + 1] * scale);
          o.y = pack_bf16x2(O[4 * g + 2] * scale, O[4 * g + 3] * scale);
          *reinterpret_cast<uint2*>(p) = o;
        } else {
          *reinterpret_cast<float4*>(p) = make_float4(O[4 * g] * scale, O[4 * g + 1] * scale, O[4 * g + 2] * scale, O[4 * g + 3] * scale);
        }
      }
    }
  }
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dy,
                                                            T* __restrict__ dqkv, int B, int Ttok, int n_head) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, hh = lane >> 5;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= B * n_head) return;
  const int b = pair / n_head, h = pair % n_head;
  const int d = n_head * HD;
  const int64_t ld = 3 * (int64_t)d;
  const T* qb = qkv + (int64_t)b * Ttok * ld + h * HD;
  const T* kb = qb + d;
  const T* vb = qb + 2 * d;
  const T* dob = dy + (int64_t)b * Ttok * d + h * HD;
  T* dqb = dqkv + (int64_t)b * Ttok * ld + h * HD;
  T* dkb = dqb + d;
  T* dvb = dqb + 2 * d;
  const bool valid = r < Ttok;
  const float scale = 1.0f / sqrtf((float)HD);

  // keys on rows, query r on this lane
  f32x16 St = rows_product<T, HD>(kb, qb, ld, r, hh, valid);
  float m = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    St[e] = row_of(e, hh) < Ttok ? St[e] * scale : -INFINITY;
    m = fmaxf(m, St[e]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    St[e] = expf(St[e] - m);
    sum += St[e];
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const float lse = m + logf(sum);   // per query (lane & 31)
#pragma unroll
  for (int e = 0; e < 16; ++e) St[e] *= inv;  // P^T[key][query]

  // dP^T[key][query] = sum_d V[key][d] dO[query][d];  rows of dO for r >= T are zero-loaded (dy row stride = d)
  f32x16 dPt;
  {
#pragma unroll
    for (int e = 0; e < 16; ++e) dPt[e] = 0.f;
    const int64_t offv = (int64_t)(valid ? r : 0) * ld, offo = (int64_t)(valid ? r : 0) * d;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        u32x4 a = *reinterpret_cast<const u32x4*>(vb + offv + ks * 16 + hh * 8);
        u32x4 o = *reinterpret_cast<const u32x4*>(dob + offo + ks * 16 + hh * 8);
        if (!valid) { a = u32x4{0, 0, 0, 0}; o = u32x4{0, 0, 0, 0}; }
        dPt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, o), dPt, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < HD / 8; ++ks) {
        float4 a = *reinterpret_cast<const float4*>(vb + offv + ks * 8 + hh * 4);
        float4 o = *reinterpret_cast<const float4*>(dob + offo + ks * 8 + hh * 4);
        if (!valid) { a = make_float4(0, 0, 0, 0); o = make_float4(0, 0, 0, 0); }
        dPt = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, o.x, dPt, 0, 0, 0);
        dPt = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, o.y, dPt, 0, 0, 0);
        dPt = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, o.z, dPt, 0, 0, 0);
        dPt = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, o.w, dPt, 0, 0, 0);
      }
    }
  }
  float delta = 0.f;  // delta_query = sum_key P dP
#pragma unroll
  for (int e = 0; e < 16; ++e) delta += St[e] * dPt[e];
  delta += __shfl_xor(delta, 32, 64);
  f32x16 dSt;
#pragma unroll
  for (int e = 0; e < 16; ++e) dSt[e] = St[e] * (dPt[e] - delta);
  // dQ[query][dd] = scale * sum_key dS[query][key] K[key][dd]
  cols_product_store<T, HD>(kb, ld, dSt, scale, dqb, ld, r, hh, Ttok);

  // queries on rows, key r on this lane: P[query][key] = exp(scale * S - lse_query)
  f32x16 S = rows_product<T, HD>(qb, kb, ld, r, hh, valid);
  f32x16 dP;
  {
#pragma unroll
    for (int e = 0; e < 16; ++e) dP[e] = 0.f;
    const int64_t offv = (int64_t)(valid ? r : 0) * ld, offo = (int64_t)(valid ? r : 0) * d;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        u32x4 o = *reinterpret_cast<const u32x4*>(dob + offo + ks * 16 + hh * 8);
        u32x4 a = *reinterpret_cast<const u32x4*>(vb + offv + ks * 16 + hh * 8);
        if (!valid) { a = u32x4{0, 0, 0, 0}; o = u32x4{0, 0, 0, 0}; }
        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o), __builtin_bit_cast(bf16x8, a), dP, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < HD / 8; ++ks) {
        float4 o = *reinterpret_cast<const float4*>(dob + offo + ks * 8 + hh * 4);
        float4 a = *reinterpret_cast<const float4*>(vb + offv + ks * 8 + hh * 4);
        if (!valid) { a = make_float4(0, 0, 0, 0); o = make_float4(0, 0, 0, 0); }
        dP = __builtin_amdgcn_mfma_f32_32x32x2f32(o.x, a.x, dP, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_32x32x2f32(o.y, a.y, dP, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_32x32x2f32(o.z, a.z, dP, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_32x32x2f32(o.w, a.w, dP, 0, 0, 0);
      }
    }
  }
  f32x16 P, dS;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int q = row_of(e, hh);                       // query index of this register
    const float lse_q = __shfl(lse, q & 31, 64);       // lanes 0..31 hold per-query values
    const float delta_q = __shfl(delta, q & 31, 64);
    const bool ok = q < Ttok && valid;                 // valid: key r < T
    P[e] = ok ? expf(S[e] * scale - lse_q) : 0.f;
    dS[e] = P[e] * (dP[e] - delta_q);
  }
  // dK[key][dd] = scale * sum_query dS[query][key] Q[query][dd];   dV[key][dd] = sum_query P[query][key] dO[query][dd]
  cols_product_store<T, HD>(qb, ld, dS, scale, dkb, ld, r, hh, Ttok);
  cols_product_store<T, HD>(dob, d, P, 1.0f, dvb, ld, r, hh, Ttok);
}

}  // namespace

extern "C" int gpt_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* resid, float* gout, void* gout_t, int gout_t_dtype,
                                 float* dgamma, float* dbeta, int rows, int d, void* stream) {
  GPT_REQUIRE(dy && x && mean && rstd && gamma && gout && dgamma && dbeta, "gpt_layernorm_bwd: null operand");
  GPT_REQUIRE(rows > 0 && d > 0 && 2 * d * 4 <= 60000, "gpt_layernorm_bwd: bad sizes (d <= 7500)");
  const int blocks = rows < 1024 ? (rows + 3) / 4 : 512;
  const int rpb = (rows + blocks - 1) / blocks;
  const dim3 grid((rows + rpb - 1) / rpb), block(256);
  const size_t lds = 2 * (size_t)d * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define LNB(DY, OUT)                                                                                              \
  hipLaunchKernelGGL((layernorm_bwd_kernel<DY, OUT>), grid, block, lds, s, (const DY*)dy, x, mean, rstd, gamma, resid, \
                     gout, (OUT*)gout_t, dgamma, dbeta, rows, d, rpb)
  const bool dyb = dy_dtype == GPT_BF16, ob = gout_t_dtype == GPT_BF16;
  if (dyb && ob) LNB(bf16_t, bf16_t);
  else if (dyb) LNB(bf16_t, float);
  else if (ob) LNB(float, bf16_t);
  else LNB(float, float);
#undef LNB
  return gpt_launch_status("gpt_layernorm_bwd");
}

extern "C" int gpt_colsum(int dtype, const void* X, int64_t ld, float* out, int rows, int cols, void* stream) {
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_colsum: bad dtype %d", dtype);
  GPT_REQUIRE(X && out && rows > 0 && cols > 0 && ld >= cols, "gpt_colsum: bad arguments");
  const int rpb = rows < 64 ? rows : (rows + 63) / 64;
  const dim3 grid((cols + 255) / 256, (rows + rpb - 1) / rpb), block(256);
  if (dtype == GPT_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)X, ld, out, rows, cols, rpb);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)X, ld, out, rows, cols, rpb);
  return gpt_launch_status("gpt_colsum");
}

extern "C" int gpt_small_gemm(int a_dtype, const void* A, int64_t a_sm, int64_t a_sk, int b_dtype, const void* Bm,
                              int64_t b_sk, int64_t b_sn, float* Cm, int64_t ldc, int M, int N, int K, int accumulate,
                              void* stream) {
  GPT_REQUIRE(A && Bm && Cm && M > 0 && N > 0 && K > 0 && ldc >= N, "gpt_small_gemm: bad arguments");
  const dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
  hipStream_t s = (hipStream_t)stream;
#define SG(TA, TB)                                                                                                  \
  hipLaunchKernelGGL((small_gemm_kernel<TA, TB>), grid, block, 0, s, (const TA*)A, a_sm, a_sk, (const TB*)Bm, b_sk, b_sn, \
                     Cm, ldc, M, N, K, accumulate)
  const bool ab = a_dtype == GPT_BF16, bb = b_dtype == GPT_BF16;
  if (ab && bb) SG(bf16_t, bf16_t);
  else if (ab) SG(bf16_t, float);
  else if (bb) SG(float, bf16_t);
  else SG(float, float);
#undef SG
  return gpt_launch_status("gpt_small_gemm");
}

extern "C" int gpt_attention_bwd(int dtype, const void* qkv, const void* dy, void* dqkv, int B, int T, int n_head, int hd,
                                 void* stream) {
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_attention_bwd: bad dtype %d", dtype);
  GPT_REQUIRE(qkv && dy && dqkv, "gpt_attention_bwd: null operand");
  GPT_REQUIRE(B > 0 && T > 0 && T <= 32 && n_head > 0, "gpt_attention_bwd: need 0 < T <= 32");
  const dim3 grid((B * n_head + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
#define AB(TT, HD) \
  hipLaunchKernelGGL((attention_bwd_kernel<TT, HD>), grid, block, 0, s, (const TT*)qkv, (const TT*)dy, (TT*)dqkv, B, T, n_head)
  if (dtype == GPT_BF16) {
    switch (hd) {
      case 32: AB(bf16_t, 32); break;
      case 64: AB(bf16_t, 64); break;
      case 96: AB(bf16_t, 96); break;
      case 128: AB(bf16_t, 128); break;
      default: gpt_set_error("gpt_attention_bwd: unsupported head dim %d", hd); return -1;
    }
  } else {
    switch (hd) {
      case 32: AB(float, 32); break;
      case 64: AB(float, 64); break;
      case 96: AB(float, 96); break;
      case 128: AB(float, 128); break;
      default: gpt_set_error("gpt_attention_bwd: unsupported head dim %d", hd); return -1;
    }
  }
#undef AB
  return gpt_launch_status("gpt_attention_bwd");
}
