// Optimizer step of the G.pt training loop on one flat parameter arena (HBM-bound, one pass):
//   global-norm clip     main.py:68-70 (torch.nn.utils.clip_grad_norm_)
//   AdamW, 2 groups      Gpt/models/transformer.py:163-213 (decay: Linear weights; no decay: biases, LayerNorm, pos_emb)
//   EMA accumulate       Gpt/utils.py:146-152, main.py:73-74
//   low-precision / K-padded operand copies of the updated weights for the next forward (replaces a separate cast pass)
// Per element: reads p, g, m, v, ema (20 B), writes p, m, v, ema (16 B) [+ g = 0 (4 B)] [+ packed copy (2 or 4 B)].
#include "common.h"
#include <math.h>

#pragma clang fp contract(off)

namespace {

constexpr int OPT_THREADS = 256;
constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(OPT_THREADS) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n4, float* __restrict__ partial) {
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
    const f32x4 v = g4[i];
    a0 += v.x * v.x;
    a1 += v.y * v.y;
    a2 += v.z * v.z;
    a3 += v.w * v.w;
  }
  float s = wave_sum((a0 + a1) + (a2 + a3));
  __shared__ float red[OPT_THREADS / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(OPT_THREADS) void sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += OPT_THREADS) s += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ double red[OPT_THREADS / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// segment of a 16-byte group (segments start on multiples of 4 elements): the last one with start <= e
__device__ __forceinline__ int find_segment(const gpt_opt_segment_t* segs, int n_seg, int64_t e) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].start <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ void write_packed(void* packed, int packed_dtype, const gpt_opt_segment_t& sg, int64_t e, f32x4 p) {
  const int64_t r = e - sg.start;
  if (sg.k == sg.ld) {
    const int64_t o = sg.packed_off + r;
    if (packed_dtype == GPT_BF16) {
      uint2 w;
      w.x = pack_bf16x2(p[0], p[1]);
      w.y = pack_bf16x2(p[2], p[3]);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(packed) + o) = w;
    } else {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(packed) + o) = p;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t rj = r + j;
      if (rj >= sg.count) break;
      const int64_t o = sg.packed_off + (rj / sg.k) * sg.ld + (rj % sg.k);
      if (packed_dtype == GPT_BF16) reinterpret_cast<bf16_t*>(packed)[o] = f32_to_bf16(p[j]);
      else reinterpret_cast<float*>(packed)[o] = p[j];
    }
  }
}

struct OptArgs {
  float* p;
  float* g;
  float* m;
  float* v;
  float* ema;
  void* packed;
  const gpt_opt_segment_t* segs;
  const float* norm_sq;
  int64_t n4;
  int n_seg, packed_dtype, zero_grad;
  float grad_scale, clip, decay_mul, w1, beta2, w2, eps, step_size, bc2_sqrt, ema_decay, ema_alpha;
};

__global__ __launch_bounds__(OPT_THREADS) void adamw_ema_kernel(OptArgs a) {
  const int64_t i4 = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x;
  if (i4 >= a.n4) return;
  const int64_t e = i4 * 4;
  const gpt_opt_segment_t sg = a.segs[find_segment(a.segs, a.n_seg, e)];

  float scale = a.grad_scale;
  if (a.clip > 0.f) {
    const float norm = sqrtf(*a.norm_sq) * a.grad_scale;
    const float coef = a.clip / (norm + 1e-6f);
    scale = scale * fminf(coef, 1.0f);
  }
  // all five streams are read before anything is written (the arena pointers may alias as far as the compiler knows, so a
  // load placed after a store waits for it), and every access is non-temporal: 1.5 GB per stream, nothing is re-used
  f32x4 p = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.p) + i4);
  f32x4 g = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.g) + i4);
  f32x4 m = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.m) + i4);
  f32x4 v = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.v) + i4);
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.ema) s = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.ema) + i4);
  const float decay = sg.decay ? a.decay_mul : 1.0f;
  const float w1 = a.w1, w2 = a.w2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gj = g[j] * scale;
    float pj = p[j] * decay;
    const float mj = m[j] + (gj - m[j]) * w1;
    const float vj = v[j] * a.beta2 + (gj * gj) * w2;
    const float denom = sqrtf(vj) / a.bc2_sqrt + a.eps;
    pj = pj - a.step_size * (mj / denom);
    p[j] = pj;
    m[j] = mj;
    v[j] = vj;
  }
  __builtin_nontemporal_store(p, reinterpret_cast<f32x4*>(a.p) + i4);
  __builtin_nontemporal_store(m, reinterpret_cast<f32x4*>(a.m) + i4);
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.v) + i4);
  if (a.zero_grad) __builtin_nontemporal_store(f32x4{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4*>(a.g) + i4);
  if (a.ema) {
    const float alpha = a.ema_alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = s[j] * a.ema_decay + p[j] * alpha;
    __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(a.ema) + i4);
  }
  if (a.packed && sg.packed_off >= 0) write_packed(a.packed, a.packed_dtype, sg, e, p);
}

__global__ __launch_bounds__(OPT_THREADS) void pack_kernel(const float* __restrict__ params, void* packed, int packed_dtype,
                                                           const gpt_opt_segment_t* __restrict__ segs, int n_seg, int64_t n4) {
  const int64_t i4 = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x;
  if (i4 >= n4) return;
  const gpt_opt_segment_t sg = segs[find_segment(segs, n_seg, i4 * 4)];
  if (sg.packed_off >= 0) write_packed(packed, packed_dtype, sg, i4 * 4, reinterpret_cast<const f32x4*>(params)[i4]);
}

// Multi-tensor variant for parameters that do NOT live in one arena (the checkpoint-layout nn.Module of the reference: 235
// separate tensors): item i covers 16-byte groups [first4, next item's first4) of a virtual concatenation.
__global__ __launch_bounds__(OPT_THREADS) void pack_tensors_kernel(const gpt_pack_item_t* __restrict__ items, int n_items, int64_t n4,
                                                                   const unsigned long long* __restrict__ changed) {
  if (changed && *changed == 0ull) return;          // gpt_pack_tensors_if_changed: the sources are what the last pack saw
  const int64_t i4 = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x;
  if (i4 >= n4) return;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first4 <= i4) lo = mid; else hi = mid - 1;
  }
  const gpt_pack_item_t it = items[lo];
  const int64_t r = (i4 - it.first4) * 4;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (r + 4 <= it.count) {
    v = *reinterpret_cast<const f32x4*>(it.src + r);
  } else {
    for (int j = 0; j < 4 && r + j < it.count; ++j) v[j] = it.src[r + j];
  }
  gpt_opt_segment_t sg;
  sg.start = 0; sg.count = it.count; sg.packed_off = 0; sg.k = it.k; sg.ld = it.ld; sg.decay = 0; sg.reserved = 0;
  if (it.k == it.ld && r + 4 > it.count) {          // ragged tail of a plain copy: element by element
    for (int j = 0; j < 4 && r + j < it.count; ++j) {
      if (it.dst_dtype == GPT_BF16) reinterpret_cast<bf16_t*>(it.dst)[r + j] = f32_to_bf16(v[j]);
      else reinterpret_cast<float*>(it.dst)[r + j] = v[j];
    }
    return;
  }
  write_packed(it.dst, it.dst_dtype, sg, r, v);
}

// Position-sensitive 64-bit checksum of every source element of a pack table: sum over the fp32 words of bits x (2 x global
// word index + 1) mod 2^64 (an odd multiplier is invertible, so any single changed word changes the sum).  One row of workgroups
// per item streams that item's 16-byte groups; wave sums by shuffles, one 64-bit vector atomic per wave.
constexpr int CKS_BLOCKS = 48;
__global__ __launch_bounds__(OPT_THREADS) void pack_checksum_kernel(const gpt_pack_item_t* __restrict__ items, unsigned long long* __restrict__ acc) {
  const gpt_pack_item_t it = items[blockIdx.y];
  const int64_t n4 = it.count >> 2;
  unsigned long long h = 0;
  for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)CKS_BLOCKS * OPT_THREADS) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(it.src + 4 * i);
    const unsigned long long w = 8ull * (unsigned long long)(it.first4 + i) + 1ull;
    h += v[0] * w + v[1] * (w + 2) + v[2] * (w + 4) + v[3] * (w + 6);
  }
  if (blockIdx.x == 0 && threadIdx.x < (it.count & 3)) {     // ragged tail (count % 4 elements)
    const int64_t j = (n4 << 2) + threadIdx.x;
    h += (unsigned long long)__float_as_uint(it.src[j]) * (2ull * (unsigned long long)(4 * it.first4 + j) + 1ull);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned int lo = __shfl_xor((unsigned int)h, o, 64), hi = __shfl_xor((unsigned int)(h >> 32), o, 64);
    h += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63) == 0 && h) atomicAdd(acc, h);
}

// state[0] = running sum (zero on entry), [1] = the sum the last pack saw, [2] = "changed" flag for the pack kernel, [3] = valid
__global__ void pack_decide_kernel(unsigned long long* __restrict__ state) {
  if (threadIdx.x == 0) {
    state[2] = (state[3] == 0ull || state[0] != state[1]) ? 1ull : 0ull;
    state[1] = state[0];
    state[0] = 0ull;
    state[3] = 1ull;
  }
}

}  // namespace

extern "C" int gpt_pack_tensors(const gpt_pack_item_t* items, int n_items, int64_t n4, void* stream) {
  GPT_REQUIRE(items && n_items > 0 && n4 > 0, "gpt_pack_tensors: bad operands");
  const int64_t blocks = (n4 + OPT_THREADS - 1) / OPT_THREADS;
  hipLaunchKernelGGL(pack_tensors_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, items, n_items, n4,
                     (const unsigned long long*)nullptr);
  return gpt_launch_status("gpt_pack_tensors");
}

extern "C" int gpt_pack_tensors_if_changed(const gpt_pack_item_t* items, int n_items, int64_t n4, uint64_t* state, void* stream) {
  GPT_REQUIRE(items && n_items > 0 && n_items <= 65535 && n4 > 0 && state && ((uintptr_t)state % 8) == 0, "gpt_pack_tensors_if_changed: bad operands");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* st = reinterpret_cast<unsigned long long*>(state);
  hipLaunchKernelGGL(pack_checksum_kernel, dim3(CKS_BLOCKS, n_items), dim3(OPT_THREADS), 0, s, items, st);
  hipLaunchKernelGGL(pack_decide_kernel, dim3(1), dim3(64), 0, s, st);
  const int64_t blocks = (n4 + OPT_THREADS - 1) / OPT_THREADS;
  hipLaunchKernelGGL(pack_tensors_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, s, items, n_items, n4, (const unsigned long long*)(st + 2));
  return gpt_launch_status("gpt_pack_tensors_if_changed");
}

extern "C" int gpt_grad_sumsq(const float* grads, int64_t n, float* scratch, float* out, void* stream) {
  GPT_REQUIRE(grads && scratch && out && n > 0 && n % 4 == 0, "gpt_grad_sumsq: need n %% 4 == 0 and non-null buffers");
  GPT_REQUIRE(((uintptr_t)grads % 16) == 0, "gpt_grad_sumsq: grads must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(OPT_THREADS), 0, (hipStream_t)stream, grads, n / 4, scratch);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(OPT_THREADS), 0, (hipStream_t)stream, scratch, SUMSQ_BLOCKS, out);
  return gpt_launch_status("gpt_grad_sumsq");
}

extern "C" int gpt_adamw_ema_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* ema, void* packed,
                                  int packed_dtype, const gpt_opt_segment_t* segments, int n_segments, int64_t n,
                                  const float* grad_norm_sq, float clip, float grad_scale, double lr, double beta1, double beta2,
                                  double eps, double weight_decay, int step, double ema_decay, int zero_grad, void* stream) {
  GPT_REQUIRE(params && grads && exp_avg && exp_avg_sq && segments, "gpt_adamw_ema_step: null operand");
  GPT_REQUIRE(n > 0 && n % 4 == 0 && n_segments > 0 && step >= 1, "gpt_adamw_ema_step: need n %% 4 == 0, segments, step >= 1");
  GPT_REQUIRE(clip <= 0.f || grad_norm_sq, "gpt_adamw_ema_step: clipping needs the squared gradient norm (gpt_grad_sumsq)");
  GPT_REQUIRE(!packed || packed_dtype == GPT_F32 || packed_dtype == GPT_BF16, "gpt_adamw_ema_step: bad packed dtype");
  for (const void* q : {(const void*)params, (const void*)grads, (const void*)exp_avg, (const void*)exp_avg_sq, (const void*)ema})
    GPT_REQUIRE(((uintptr_t)q % 16) == 0, "gpt_adamw_ema_step: arenas must be 16-byte aligned");
  OptArgs a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.ema = ema; a.packed = packed;
  a.segs = segments; a.norm_sq = grad_norm_sq; a.n4 = n / 4; a.n_seg = n_segments; a.packed_dtype = packed_dtype;
  a.zero_grad = zero_grad; a.grad_scale = grad_scale; a.clip = clip;
  // the scalar factors are formed in double on the host, as the reference's Python floats are, then rounded once
  a.decay_mul = (float)(1.0 - lr * weight_decay);
  a.w1 = (float)(1.0 - beta1); a.beta2 = (float)beta2; a.w2 = (float)(1.0 - beta2); a.eps = (float)eps;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.step_size = (float)(lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.ema_decay = (float)ema_decay; a.ema_alpha = (float)(1.0 - ema_decay);
  const int64_t blocks = (a.n4 + OPT_THREADS - 1) / OPT_THREADS;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, a);
  return gpt_launch_status("gpt_adamw_ema_step");
}

extern "C" int gpt_pack_segments(const float* params, void* packed, int packed_dtype, const gpt_opt_segment_t* segments,
                                 int n_segments, int64_t n, void* stream) {
  GPT_REQUIRE(params && packed && segments && n > 0 && n % 4 == 0 && n_segments > 0, "gpt_pack_segments: bad operands");
  GPT_REQUIRE(packed_dtype == GPT_F32 || packed_dtype == GPT_BF16, "gpt_pack_segments: bad packed dtype");
  GPT_REQUIRE(((uintptr_t)params % 16) == 0 && ((uintptr_t)packed % 16) == 0, "gpt_pack_segments: buffers must be 16-byte aligned");
  const int64_t n4 = n / 4, blocks = (n4 + OPT_THREADS - 1) / OPT_THREADS;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, packed, packed_dtype,
                     segments, n_segments, n4);
  return gpt_launch_status("gpt_pack_segments");
}
