// Short-sequence fused multi-head attention for gfx950 (T <= 32 tokens, no mask, dropout 0).
// Replaces Gpt/models/transformer.py:43-47:  softmax(q k^T / sqrt(hd)) v  per (batch, head).
//
// One wavefront owns one (batch, head).  T is padded to the 32-row MFMA tile.
//   S^T = K . Q^T        MFMA with K as the A operand and Q as the B operand, so the accumulator holds, per
//                        lane, ONE query column (lane & 31) and 16 of the 32 key rows in its registers:
//                        key(reg, half) = (reg & 3) + 8 * (reg >> 2) + 4 * half.
//   softmax over keys    = over the lane's 16 registers plus one exchange with lane ^ 32 (wavefront
//                        reduction, no LDS);
//   O^T = V^T . P^T      the softmaxed accumulator IS the B operand of the second MFMA (a product that sums
//                        over the accumulator's row index needs no lane movement); V^T fragments are read
//                        with the matching key order.  bf16: the V tile is staged in LDS by coalesced 16-byte loads
//                        (rows >= T zero-filled) and its transposed fragments (8 keys of one head-dim per lane) come
//                        from ds_read_b64_tr_b16, 12 LDS reads instead of 48 two-byte global loads per lane.
// fp32: v_mfma_f32_32x32x2_f32 (exact).  bf16: v_mfma_f32_32x32x16_bf16, P rounded to bf16 once.
#include "common.h"
#include <stdlib.h>

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s_ptr;
// LDS row pitch of the staged V tile: the 4 rows x 16 columns blocks of the transposing read land on 64 distinct banks
// with 192-byte rows (48 banks apart) and with 320-byte rows (16 banks apart), not with 128- or 256-byte rows
template <int HD>
constexpr int v_pitch = HD == 128 ? 320 : 192;

__device__ __forceinline__ int key_of(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// OUT_LDS (bf16): the output tile goes back through the wave's LDS tile (the V tile has been consumed by then) so that it
// leaves as 16-byte stores that cover whole rows of the head (HD * 2 contiguous bytes) instead of one 8-byte store per lane
// and (row, 4 head-dims): 12 -> 5 store instructions per wave at HD = 96, every one made of full lines.
template <typename T, int HD, bool OUT_LDS>
__device__ __forceinline__ void attention_body(const T* __restrict__ qkv, T* __restrict__ y, int B, int Ttok, int n_head, int Tout) {
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
  const bool rvalid = r < Ttok;
  const int64_t roff = (int64_t)(rvalid ? r : 0) * ld;

  // bf16: V tile -> registers now (in flight during the score MFMAs), -> LDS after them
  constexpr int VP = v_pitch<HD>, CPR = HD / 8, NVL = (32 * CPR + 63) / 64;
  constexpr int OP = HD * 2 + 16;                                      // pitch of the output tile (OUT_LDS), see the end
  constexpr int TILE = 32 * (OUT_LDS && OP > VP ? OP : VP);            // per wave: the V tile, later the output tile
  __shared__ __attribute__((aligned(16))) char vsm[sizeof(T) == 2 ? 4 * TILE : 16];
  char* vt = vsm + (sizeof(T) == 2 ? (threadIdx.x >> 6) * TILE : 0);
  u32x4 vreg[NVL];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c = lane + 64 * i, row = c / CPR, col = c - row * CPR;
      vreg[i] = u32x4{0, 0, 0, 0};
      if (row < Ttok) vreg[i] = *reinterpret_cast<const u32x4*>(vb + (int64_t)row * ld + col * 8);
    }
  }

  f32x16 S;
#pragma unroll
  for (int e = 0; e < 16; ++e) S[e] = 0.f;

  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
      u32x4 a = *reinterpret_cast<const u32x4*>(kb + roff + ks * 16 + hh * 8);
      u32x4 q = *reinterpret_cast<const u32x4*>(qb + roff + ks * 16 + hh * 8);
      if (!rvalid) { a = u32x4{0, 0, 0, 0}; q = u32x4{0, 0, 0, 0}; }
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, q), S, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < HD / 8; ++ks) {
      float4 a = *reinterpret_cast<const float4*>(kb + roff + ks * 8 + hh * 4);
      float4 q = *reinterpret_cast<const float4*>(qb + roff + ks * 8 + hh * 4);
      if (!rvalid) { a = make_float4(0, 0, 0, 0); q = make_float4(0, 0, 0, 0); }
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q.x, S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q.y, S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q.z, S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q.w, S, 0, 0, 0);
    }
  }

  // softmax over keys for query column r
  const float scale = 1.0f / sqrtf((float)HD);
  float m = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const bool kvalid = key_of(e, hh) < Ttok;
    S[e] = kvalid ? S[e] * scale : -INFINITY;
    m = fmaxf(m, S[e]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    S[e] = expf(S[e] - m);
    sum += S[e];
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int e = 0; e < 16; ++e) S[e] *= inv;

  // O^T tiles: 32 head-dims each
  constexpr int ND = HD / 32;
  f32x16 O[ND];
#pragma unroll
  for (int t = 0; t < ND; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[t][e] = 0.f;

  if constexpr (sizeof(T) == 2) {
    bf16x8 p[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const uint32_t pk = pack_bf16x2(S[8 * s + j], S[8 * s + j + 1]);
        p[s][j] = (short)(pk & 0xffffu);
        p[s][j + 1] = (short)(pk >> 16);
      }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c = lane + 64 * i, row = c / CPR, col = c - row * CPR;
      if (c < 32 * CPR) *reinterpret_cast<u32x4*>(vt + row * VP + col * 16) = vreg[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes only: the LDS serves one wave's instructions in order
    // transposing read: in every 16-lane group gq, lane 4 q + pp supplies row (key) q, columns 4 pp .. 4 pp + 3 of a
    // 4 x 16 block and lane i receives column i of its four rows: keys 16 s + 8 u + 4 hh + q, head-dims 32 t + 16 (gq & 1) + i
    const int gq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const char* tbase = vt + (4 * (gq >> 1) + tq) * VP + (16 * (gq & 1) + 4 * tp) * 2;
#pragma unroll
    for (int t = 0; t < ND; ++t) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(tbase + (16 * s) * VP + t * 64));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(tbase + (16 * s + 8) * VP + t * 64));
        const bf16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, p[s], O[t], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < ND; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = key_of(e, hh);
        const float v = key < Ttok ? vb[(int64_t)key * ld + t * 32 + r] : 0.f;
        O[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, S[e], O[t], 0, 0, 0);
      }
    }
  }

  if constexpr (OUT_LDS && sizeof(T) == 2) {
    // pitch HD * 2 + 16 bytes: the 16 rows of a ds_write_b64 lane group start (HD / 2 + 4) r banks apart = 16 distinct
    // multiples of 4 banks for HD = 32, 64, 96, 128; Ttok * pitch <= 32 * v_pitch<HD>
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the transposing reads of V have returned (their MFMAs consumed them)
    if (rvalid) {
#pragma unroll
      for (int t = 0; t < ND; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 o;
          o.x = pack_bf16x2(O[t][4 * g + 0], O[t][4 * g + 1]);
          o.y = pack_bf16x2(O[t][4 * g + 2], O[t][4 * g + 3]);
          *reinterpret_cast<uint2*>(vt + r * OP + (t * 32 + 8 * g + 4 * hh) * 2) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // Tout < Ttok (the last Block of an inference forward): only the query rows the decoder reads are stored, compactly
    T* yb = y + (int64_t)b * Tout * d + h * HD;
    const int nchunk = Tout * CPR;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c = lane + 64 * i, row = c / CPR, col = c - row * CPR;
      if (c < nchunk) *reinterpret_cast<u32x4*>(yb + (int64_t)row * d + col * 8) = *reinterpret_cast<const u32x4*>(vt + row * OP + col * 16);
    }
    return;
  }
  // O^T[dd][query r]: registers 4g..4g+3 are 4 consecutive head-dims -> one 8/16-byte store per group
  if (r < Tout) {
    T* yr = y + ((int64_t)b * Tout + r) * d + h * HD;
#pragma unroll
    for (int t = 0; t < ND; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = t * 32 + 8 * g + 4 * hh;
        if constexpr (sizeof(T) == 2) {
          uint2 o;
          o.x = pack_bf16x2(O[t][4 * g + 0], O[t][4 * g + 1]);
          o.y = pack_bf16x2(O[t][4 * g + 2], O[t][4 * g + 3]);
          *reinterpret_cast<uint2*>(yr + dd) = o;
        } else {
          *reinterpret_cast<float4*>(yr + dd) = make_float4(O[t][4 * g + 0], O[t][4 * g + 1], O[t][4 * g + 2], O[t][4 * g + 3]);
        }
      }
    }
  }
}

template <typename T, int HD, bool OUT_LDS>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ y, int B, int Ttok, int n_head, int Tout) {
  attention_body<T, HD, OUT_LDS>(qkv, y, B, Ttok, n_head, Tout);
}
// at most 96 registers: 5 waves per SIMD = 20 per CU (B = 320, 16 heads: 5120 waves = exactly one round of 256 CUs)
template <typename T, int HD, bool OUT_LDS>
__global__ __launch_bounds__(256, 5) void attention_kernel_w5(const T* __restrict__ qkv, T* __restrict__ y, int B, int Ttok, int n_head, int Tout) {
  attention_body<T, HD, OUT_LDS>(qkv, y, B, Ttok, n_head, Tout);
}

template <typename T>
int launch(const void* qkv, void* y, int B, int T_, int n_head, int hd, int t_out, hipStream_t s) {
  const dim3 grid((B * n_head + 3) / 4), block(256);
  // variant (A/B measurements, GPT_ATT_VARIANT): bit 0 = output through LDS, bit 1 = 5 waves per SIMD.  Same box, B = 320,
  // HD = 96: 28.6 us (0) -> 23.2 (1) -> 24.3 (2) -> 21.0 (3).  HD = 128 keeps 4 waves per SIMD (its 40 KiB of LDS per workgroup
  // allow no more).
  static const int variant = [] { const char* e = getenv("GPT_ATT_VARIANT"); return e ? atoi(e) : 3; }();
#define ATT_ARGS grid, block, 0, s, (const T*)qkv, (T*)y, B, T_, n_head, t_out
#define ATT(HD)                                                                                       \
  do {                                                                                                \
    if constexpr (HD <= 96) {                                                                         \
      if (variant == 2) { hipLaunchKernelGGL((attention_kernel_w5<T, HD, false>), ATT_ARGS); break; } \
      if (variant == 3) { hipLaunchKernelGGL((attention_kernel_w5<T, HD, true>), ATT_ARGS); break; }  \
    }                                                                                                 \
    if (variant == 0) hipLaunchKernelGGL((attention_kernel<T, HD, false>), ATT_ARGS);                 \
    else hipLaunchKernelGGL((attention_kernel<T, HD, true>), ATT_ARGS);                               \
  } while (0)
  switch (hd) {
    case 32: ATT(32); break;
    case 64: ATT(64); break;
    case 96: ATT(96); break;
    case 128: ATT(128); break;
    default: gpt_set_error("gpt_attention: unsupported head dim %d", hd); return -1;
  }
#undef ATT
#undef ATT_ARGS
  return gpt_launch_status("gpt_attention");
}

}  // namespace

extern "C" int gpt_attention(int dtype, const void* qkv, void* y, int B, int T, int n_head, int hd, void* stream) {
  return gpt_attention_rows(dtype, qkv, y, B, T, T, n_head, hd, stream);
}

// t_out <= T: only the first t_out query rows of every sample are stored, as a compact [B, t_out, d] tensor
int gpt_attention_rows(int dtype, const void* qkv, void* y, int B, int T, int t_out, int n_head, int hd, void* stream) {
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_attention: bad dtype %d", dtype);
  GPT_REQUIRE(t_out > 0 && t_out <= T, "gpt_attention: stored rows %d out of range (T=%d)", t_out, T);
  GPT_REQUIRE(qkv && y, "gpt_attention: null operand");
  GPT_REQUIRE(B > 0 && T > 0 && T <= 32 && n_head > 0, "gpt_attention: need 0 < T <= 32 (B=%d T=%d H=%d)", B, T, n_head);
  GPT_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)y % 16) == 0, "gpt_attention: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  return dtype == GPT_BF16 ? launch<bf16_t>(qkv, y, B, T, n_head, hd, t_out, s) : launch<float>(qkv, y, B, T, n_head, hd, t_out, s);
}
