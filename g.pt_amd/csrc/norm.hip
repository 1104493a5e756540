// Fused LayerNorm for the residual stream (fp32 in, fp32 or bf16 out), one wavefront per row.
// HBM-bound: algorithmic bytes per row = 4*d read + sizeof(out)*d written.
// Replaces nn.LayerNorm at Gpt/models/transformer.py:70-71 (ln1/ln2), :226 (ln_in, with the
// position-embedding add of :308-309 folded in through `add`), :311 (ln_f).
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;

// NV = float4 per lane held in registers (d == NV * 256); NV == 0: generic d % 4 == 0, re-reads the row.
// The element-wise steps the LayerNorm kernels of this file share.  The fused multiply-adds are written out and contraction is
// off around them, so the kernels agree bit for bit with one another whatever the compiler would have contracted in each
// (left to it, the one-row, two-row and paired kernels came out with different fma placements: results 1 ulp apart).
__device__ __forceinline__ float ln_sum4(const float4& t) { return (t.x + t.y) + (t.z + t.w); }
__device__ __forceinline__ float ln_sq4(const float4& t, float mean) {
#pragma clang fp contract(off)
  const float a = t.x - mean, b = t.y - mean, c = t.z - mean, e = t.w - mean;
  return __builtin_fmaf(a, a, b * b) + __builtin_fmaf(c, c, e * e);
}
__device__ __forceinline__ float4 ln_affine4(const float4& t, float mean, float rstd, const float4& g, const float4& b) {
#pragma clang fp contract(off)
  float4 o;
  o.x = __builtin_fmaf((t.x - mean) * rstd, g.x, b.x);
  o.y = __builtin_fmaf((t.y - mean) * rstd, g.y, b.y);
  o.z = __builtin_fmaf((t.z - mean) * rstd, g.z, b.z);
  o.w = __builtin_fmaf((t.w - mean) * rstd, g.w, b.w);
  return o;
}

template <int NV, typename OutT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ add,
                                                        int add_rows, OutT* __restrict__ y, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out, int rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * d);
  const int nv = d >> 2;
  float4 v[NV > 0 ? NV : 1];
  float s = 0.f;
  if constexpr (NV > 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = xr[lane + i * 64];
      s += ln_sum4(v[i]);
    }
  } else {
    for (int c = lane; c < nv; c += 64) s += ln_sum4(xr[c]);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  if constexpr (NV > 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) q += ln_sq4(v[i], mean);
  } else {
    for (int c = lane; c < nv; c += 64) q += ln_sq4(xr[c], mean);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  const float4* a4 = add ? reinterpret_cast<const float4*>(add + (int64_t)(row % add_rows) * d) : nullptr;
  OutT* yr = y + (int64_t)row * d;
  auto emit = [&](int c, const float4& t) {
    float4 o = ln_affine4(t, mean, rstd, g4[c], b4[c]);
    if (a4) {
      const float4 a = a4[c];
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    if constexpr (sizeof(OutT) == 4) {
      reinterpret_cast<float4*>(yr)[c] = o;
    } else {
      uint2 p;
      p.x = pack_bf16x2(o.x, o.y);
      p.y = pack_bf16x2(o.z, o.w);
      reinterpret_cast<uint2*>(yr)[c] = p;
    }
  };
  if constexpr (NV > 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) emit(lane + i * 64, v[i]);
  } else {
    for (int c = lane; c < nv; c += 64) emit(c, xr[c]);
  }
}

// RPW (two) rows per wave (d == NV * 256, no `add`): both rows' loads are in flight together and gamma / beta are read once for the
// pair - at M = 8320 they were 12 KiB of L1 traffic per 6-KiB row.  Per row the arithmetic (and its order) is that of
// layernorm_kernel: bit-identical results.
template <int NV, int RPW, typename OutT>
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, OutT* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows, int d,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = RPW * (blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6));
  if (row0 >= rows) return;
  float4 v[RPW][NV];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)min(row0 + rr, rows - 1) * d);   // past the end: the last row again
#pragma unroll
    for (int i = 0; i < NV; ++i) v[rr][i] = xr[lane + i * 64];
  }
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float4 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gm[i] = g4[lane + i * 64];
    bt[i] = b4[lane + i * 64];
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = row0 + rr;
    if (row >= rows) break;                                 // wave-uniform
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += ln_sum4(v[rr][i]);
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) q += ln_sq4(v[rr][i], mean);
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
    OutT* yr = y + (int64_t)row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 o = ln_affine4(v[rr][i], mean, rstd, gm[i], bt[i]);
      if constexpr (sizeof(OutT) == 4) {
        reinterpret_cast<float4*>(yr)[lane + i * 64] = o;
      } else {
        uint2 p;
        p.x = pack_bf16x2(o.x, o.y);
        p.y = pack_bf16x2(o.z, o.w);
        reinterpret_cast<uint2*>(yr)[lane + i * 64] = p;
      }
    }
  }
}

// ln_in + position embedding and the first Block's ln1 in one pass (inference forward): x0 = LN(x; g0, b0) + add -> fp32, then
// y = LN(x0; g1, b1) -> OutT from the registers.  Per row the arithmetic and its order are those of two layernorm_kernel launches
// (bit-identical); saves a launch and the re-read of x0.
template <int NV, typename OutT>
__global__ __launch_bounds__(256) void layernorm_pair_kernel(const float* __restrict__ x, const float* __restrict__ g0, const float* __restrict__ b0,
                                                             const float* __restrict__ add, int add_rows, float* __restrict__ x0,
                                                             const float* __restrict__ g1, const float* __restrict__ b1,
                                                             OutT* __restrict__ y, int rows, float eps) {
  constexpr int d = NV * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * d);
  float4 v[NV];
  auto norm = [&](const float* __restrict__ gamma, const float* __restrict__ beta, const float4* a4) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += ln_sum4(v[i]);
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) q += ln_sq4(v[i], mean);
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      float4 o = ln_affine4(v[i], mean, rstd, reinterpret_cast<const float4*>(gamma)[c], reinterpret_cast<const float4*>(beta)[c]);
      if (a4) {
        const float4 a = a4[c];
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      }
      v[i] = o;
    }
  };
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = xr[lane + i * 64];
  norm(g0, b0, reinterpret_cast<const float4*>(add + (int64_t)(row % add_rows) * d));
  float4* x0r = reinterpret_cast<float4*>(x0 + (int64_t)row * d);
#pragma unroll
  for (int i = 0; i < NV; ++i) x0r[lane + i * 64] = v[i];
  norm(g1, b1, nullptr);
  OutT* yr = y + (int64_t)row * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if constexpr (sizeof(OutT) == 4) {
      reinterpret_cast<float4*>(yr)[lane + i * 64] = v[i];
    } else {
      uint2 p;
      p.x = pack_bf16x2(v[i].x, v[i].y);
      p.y = pack_bf16x2(v[i].z, v[i].w);
      reinterpret_cast<uint2*>(yr)[lane + i * 64] = p;
    }
  }
}

template <typename OutT>
int launch(const float* x, const float* gamma, const float* beta, const float* add, int add_rows, void* y,
           float* mean, float* rstd, int rows, int d, float eps, hipStream_t s) {
  const dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  static const bool pair = [] { const char* e = getenv("GPT_LN_PAIR"); return !e || atoi(e) != 0; }();   // 0: one row per wave (A/B)
  if (pair && !add && rows >= 4096 && (d == 1536 || d == 2048)) {
    constexpr int R = 2;                    // 3 rows per wave measure the same (5.766 against 5.770 ms per B = 320 sampling step), 4 worse (5.81)
    const dim3 grid2((rows + R * ROWS_PER_BLOCK - 1) / (R * ROWS_PER_BLOCK));
    if (d == 1536) hipLaunchKernelGGL((layernorm2_kernel<6, R, OutT>), grid2, block, 0, s, x, gamma, beta, (OutT*)y, mean, rstd, rows, d, eps);
    else hipLaunchKernelGGL((layernorm2_kernel<8, R, OutT>), grid2, block, 0, s, x, gamma, beta, (OutT*)y, mean, rstd, rows, d, eps);
    return gpt_launch_status("gpt_layernorm");
  }
#define LN_LAUNCH(NV)                                                                                          \
  hipLaunchKernelGGL((layernorm_kernel<NV, OutT>), grid, block, 0, s, x, gamma, beta, add, add_rows, (OutT*)y, \
                     mean, rstd, rows, d, eps)
  if (d == 1536) LN_LAUNCH(6);
  else if (d == 2048) LN_LAUNCH(8);
  else if (d == 1024) LN_LAUNCH(4);
  else if (d == 768) LN_LAUNCH(3);
  else if (d == 3072) LN_LAUNCH(12);
  else LN_LAUNCH(0);
#undef LN_LAUNCH
  return gpt_launch_status("gpt_layernorm");
}

}  // namespace

extern "C" int gpt_layernorm(const float* x, const float* gamma, const float* beta, const float* add, int add_rows,
                             void* y, int out_dtype, float* mean, float* rstd, int rows, int d, float eps,
                             void* stream) {
  GPT_REQUIRE(x && gamma && beta && y, "gpt_layernorm: null operand");
  GPT_REQUIRE(rows > 0 && d > 0 && d % 4 == 0, "gpt_layernorm: need rows > 0 and d %% 4 == 0 (rows=%d d=%d)", rows, d);
  GPT_REQUIRE(out_dtype == GPT_F32 || out_dtype == GPT_BF16, "gpt_layernorm: bad out_dtype %d", out_dtype);
  GPT_REQUIRE(!add || add_rows > 0, "gpt_layernorm: add needs add_rows > 0");
  GPT_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)gamma % 16) == 0 &&
                  ((uintptr_t)beta % 16) == 0 && ((uintptr_t)add % 16) == 0,
              "gpt_layernorm: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == GPT_F32) return launch<float>(x, gamma, beta, add, add_rows, y, mean, rstd, rows, d, eps, s);
  return launch<bf16_t>(x, gamma, beta, add, add_rows, y, mean, rstd, rows, d, eps, s);
}

// x0 = LN(x; g0, b0) + add[row % add_rows] (fp32), y = LN(x0; g1, b1) (out_dtype), one launch.  Widths 1536, 2048 and 3072 only
// (gpt_layernorm_pair_has): for any other the caller makes the two gpt_layernorm calls
bool gpt_layernorm_pair_has(int d) { return d == 1536 || d == 2048 || d == 3072; }

int gpt_layernorm_pair(const float* x, const float* g0, const float* b0, const float* add, int add_rows, float* x0, const float* g1,
                       const float* b1, void* y, int out_dtype, int rows, int d, float eps, void* stream) {
  GPT_REQUIRE(x && g0 && b0 && add && add_rows > 0 && x0 && g1 && b1 && y && rows > 0, "gpt_layernorm_pair: bad operands");
  GPT_REQUIRE(gpt_layernorm_pair_has(d), "gpt_layernorm_pair: no instance for d=%d", d);
  GPT_REQUIRE(out_dtype == GPT_F32 || out_dtype == GPT_BF16, "gpt_layernorm_pair: bad out_dtype %d", out_dtype);
  GPT_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)x0 % 16) == 0 && ((uintptr_t)y % 8) == 0 && ((uintptr_t)g0 % 16) == 0 &&
                  ((uintptr_t)b0 % 16) == 0 && ((uintptr_t)g1 % 16) == 0 && ((uintptr_t)b1 % 16) == 0 && ((uintptr_t)add % 16) == 0,
              "gpt_layernorm_pair: operands must be 16-byte aligned");
  const dim3 grid((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LNP(NV)                                                                                                                   \
  do {                                                                                                                            \
    if (out_dtype == GPT_F32)                                                                                                     \
      hipLaunchKernelGGL((layernorm_pair_kernel<NV, float>), grid, block, 0, s, x, g0, b0, add, add_rows, x0, g1, b1, (float*)y, rows, eps); \
    else                                                                                                                          \
      hipLaunchKernelGGL((layernorm_pair_kernel<NV, bf16_t>), grid, block, 0, s, x, g0, b0, add, add_rows, x0, g1, b1, (bf16_t*)y, rows, eps); \
  } while (0)
  if (d == 1536) LNP(6);
  else if (d == 2048) LNP(8);
  else LNP(12);
#undef LNP
  return gpt_launch_status("gpt_layernorm_pair");
}

// dst[b * t_out + r, :] = src[b * T + r, :] for r < t_out: the residual-stream rows of the tokens the decoder reads, compacted for
// the pruned last Block of an inference forward (model.cpp).  fp32, d % 4 == 0, 16-byte aligned.
namespace {
__global__ __launch_bounds__(256) void gather_token_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int t_out,
                                                                int d4, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / d4;
    const int c = (int)(i - row * d4);
    const int64_t b = row / t_out;
    const int r = (int)(row - b * t_out);
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[(b * T + r) * d4 + c];
  }
}
}  // namespace

int gpt_gather_token_rows(const float* src, float* dst, int B, int T, int t_out, int d, void* stream) {
  GPT_REQUIRE(src && dst && B > 0 && t_out > 0 && t_out <= T && d > 0 && d % 4 == 0, "gpt_gather_token_rows: bad arguments");
  GPT_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "gpt_gather_token_rows: buffers must be 16-byte aligned");
  const int64_t n4 = (int64_t)B * t_out * (d / 4);
  const int64_t blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(gather_token_rows_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, src, dst, T,
                     t_out, d / 4, n4);
  return gpt_launch_status("gpt_gather_token_rows");
}


namespace {
__global__ __launch_bounds__(256) void scatter_token_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int T, int t_out,
                                                                 int v_per_row, int64_t n_vec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / v_per_row;
    const int c = (int)(i - row * v_per_row);
    const int64_t b = row / T;
    const int r = (int)(row - b * T);
    dst[i] = r < t_out ? src[(b * t_out + r) * v_per_row + c] : make_uint4(0u, 0u, 0u, 0u);
  }
}
}  // namespace

int gpt_scatter_token_rows(const void* src, void* dst, int B, int T, int t_out, int d, int elem_bytes, void* stream) {
  GPT_REQUIRE(src && dst && src != dst && B > 0 && t_out > 0 && t_out <= T && d > 0 && (elem_bytes == 2 || elem_bytes == 4) &&
                  (d * elem_bytes) % 16 == 0,
              "gpt_scatter_token_rows: bad arguments");
  GPT_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "gpt_scatter_token_rows: buffers must be 16-byte aligned");
  const int vpr = d * elem_bytes / 16;
  const int64_t n_vec = (int64_t)B * T * vpr, blocks = (n_vec + 255) / 256;
  hipLaunchKernelGGL(scatter_token_rows_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, (uint4*)dst, T, t_out, vpr, n_vec);
  return gpt_launch_status("gpt_scatter_token_rows");
}
