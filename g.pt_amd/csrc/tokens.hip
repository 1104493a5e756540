// Token-side kernels of the G.pt denoiser: frequency embedding of the scalar inputs, the ragged
// per-token encoder Linears and the ragged per-token decoder Linears (+ global residual).
//   FrequencyEmbedder.forward / Gpt.embed_loss   Gpt/models/transformer.py:324-334, 427-438
//   GPT.encode_parameters                        :215-228   (26 Linears, K in {10..1000}, M = batch)
//   GPT.decode_parameters + global residual      :230-242, 462-463
// The encoder reads its four sources (x, x_prev, scalar tokens) in place, so the (B, 16948)
// concatenation of :458-459 is never materialised.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// The reference rounds every product and sum separately (separate ATen kernels); hipcc contracts a*b+c into an
// FMA by default, __fmul_rn/__fadd_rn included, which changes last bits.  Keep IEEE op-by-op rounding in this file.
#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------ scalar tokens
// One thread per output element.  The product f * s is formed in fp32 with one rounding and fed to the
// full-range sinf/cosf (arguments reach 1.6e7 rad, where one fp32 ulp is a full radian).
__global__ void scalar_tokens_kernel(const int64_t* __restrict__ t, const void* __restrict__ lt,
                                     const void* __restrict__ lp, int is_f64, const float* __restrict__ freqs,
                                     int nfreq, int n_scalar, float* __restrict__ out, int B) {
  const int W = 2 * nfreq + 1;
  const int64_t total = (int64_t)B * n_scalar * W;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int f = (int)(idx % W);
  const int j = (int)((idx / W) % n_scalar);
  const int b = (int)(idx / ((int64_t)W * n_scalar));
  float s;
  if (j == 0) {
    s = (float)t[b];
  } else if (is_f64) {
    const double a = ((const double*)lt)[b], c = ((const double*)lp)[b];
    s = j == 1 ? (float)(a - c) : (j == 2 ? (float)a : (float)c);
  } else {
    const float a = ((const float*)lt)[b], c = ((const float*)lp)[b];
    s = j == 1 ? a - c : (j == 2 ? a : c);
  }
  float v;
  if (f < nfreq) {
    float arg = freqs[f] * s;
    asm volatile("" : "+v"(arg));   // the rounded product is the argument: nothing may be fused into sinf's reduction
    v = sinf(arg);
  } else if (f < 2 * nfreq) {
    float arg = freqs[f - nfreq] * s;
    asm volatile("" : "+v"(arg));
    v = cosf(arg);
  } else {
    v = s;
  }
  out[idx] = v;
}

// ------------------------------------------------------------------------------------------ ragged GEMMs
// Workgroup tile: 64 batch rows x 128 output columns of ONE token; 4 waves as 2 x 2, each 32 x 64
// (1 x 2 MFMA 32x32 tiles).  K is walked in chunks of KC elements through a single LDS buffer, the next chunk's global loads
// in flight (register staging) during the MFMAs of the current one; ragged K (10, 100, 257, 840, 1000 ...) is zero-filled in
// LDS, never in memory on the activation side.  KC = 128 for bf16 (32 for fp32, whose chunk is twice the bytes): every chunk
// costs one exposed memory latency, so r01's 32-element chunks (4 MFMAs per wave between two latencies) made these kernels
// latency-bound: 61 / 48 us per sampling step at B = 320 for ~100 MB of traffic (r02: see DESIGN.md section 4).
constexpr int MAX_TOK = 40;
constexpr int TB = 64;             // batch rows per workgroup; output columns: 64 NJ (NJ = 32-column MFMA tiles per wave, 1 or 2)
constexpr int KC_PAD = 32;          // the host pads encoder weight rows to a multiple of this (denoiser.py)

struct EncTok {
  const float* src;   // fp32 [B, >=k], row stride src_ld
  const void* w;      // T [d, w_ld], zero padded to a multiple of KC_PAD
  int64_t src_ld, w_ld;
  int k;
  int slot;           // output token index
};
struct EncArgs {
  EncTok tok[MAX_TOK];
  float* out;         // fp32 [B, n_tok, d]
  int n_tok, B, d;
};

struct DecTok {
  const void* w;      // T [n, d]
  int n, off;
};
struct DecArgs {
  DecTok tok[MAX_TOK];
  const void* feat;   // T [B, T, d]
  const float* resid; // fp32 [B, D] or null
  float* out;         // fp32 [B, D]
  int n_tok, T, B, d, D;
};

template <typename T, int KC>
struct Pitch {
  static constexpr int value = KC * sizeof(T) + 16;  // bytes per LDS row: 272 (bf16, KC 128) / 80 (bf16, 32) / 144 (fp32, 32)
};

template <typename T, int KC, int NJ>
__device__ __forceinline__ void mma_chunk(const char* ldsA, const char* ldsW, int wr, int wc, int lane, f32x16 (&acc)[NJ]) {
  constexpr int P = Pitch<T, KC>::value;
  const int r = lane & 31, hh = lane >> 5;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int s = 0; s < KC / 16; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(ldsA + (wr * 32 + r) * P + (s * 16 + hh * 8) * 2);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(ldsW + (wc * 32 * NJ + j * 32 + r) * P + (s * 16 + hh * 8) * 2);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < KC / 8; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(ldsA + (wr * 32 + r) * P + (q * 8 + hh * 4) * 4);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float4 b = *reinterpret_cast<const float4*>(ldsW + (wc * 32 * NJ + j * 32 + r) * P + (q * 8 + hh * 4) * 4);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
      }
    }
  }
}

// a [ROWS x KC] chunk of a row-major T matrix (16-byte aligned rows) = NV 16-byte vectors per thread; rows >= nrows are clamped
// (dropped at store), vectors starting at or beyond column kmax (a multiple of the vector width) are zero.  Split into a register
// load and an LDS store so that the global loads of chunk t+1 fly during the MFMAs of chunk t (single LDS buffer).
// BOUND = false: every vector lies inside the rows (kmax is ignored: no clamp, no zeroing)
template <typename T, int KC, int ROWS, bool BOUND = true>
struct Stage {
  static constexpr int V = KC * sizeof(T) / 16;   // 16-byte vectors per row
  static constexpr int NV = ROWS * V / 256;       // per thread
  static constexpr int EPV = 16 / sizeof(T);      // elements per vector
  u32x4 v[NV];
  int k0, kmax;
  // Nothing here may CONSUME a loaded value (the zeroing of out-of-range vectors happens in store()), and no branch may
  // surround a load (clamped address instead): either makes hipcc wait for the loads inside the chunk that issued them.
  __device__ __forceinline__ void load(const T* base, int64_t ld, int row0, int nrows, int k0_, int kmax_, int tid) {
    k0 = k0_;
    kmax = kmax_;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + j * 256, row = i / V, c = i % V;
      int gr = row0 + row;
      gr = gr < nrows ? gr : nrows - 1;
      const int k = k0 + c * EPV;
      v[j] = *reinterpret_cast<const u32x4*>(base + (int64_t)gr * ld + (!BOUND || k < kmax ? k : 0));
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
    constexpr int P = Pitch<T, KC>::value;
    __builtin_amdgcn_sched_barrier(0);    // the zeroing selects below consume the loads: they must not be scheduled into the chunk before
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int i = tid + j * 256, row = i / V, c = i % V;
      const bool ok = !BOUND || k0 + c * EPV < kmax;
      u32x4 t = v[j];
      t.x = ok ? t.x : 0u; t.y = ok ? t.y : 0u; t.z = ok ? t.z : 0u; t.w = ok ? t.w : 0u;
      *reinterpret_cast<u32x4*>(lds + row * P + c * 16) = t;
    }
  }
};

template <typename T, int KC, int NJ>
__global__ __launch_bounds__(256) void encode_kernel(EncArgs g) {
  constexpr int P = Pitch<T, KC>::value, TN = 64 * NJ;
  __shared__ __attribute__((aligned(16))) char smem[(TB + TN) * P];
  char* ldsA = smem;
  char* ldsW = smem + TB * P;
  const EncTok tk = g.tok[blockIdx.z];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int b0 = blockIdx.y * TB, n0 = blockIdx.x * TN;
  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // activations: 64 rows x 4 threads; a thread holds groups of 8 consecutive k, 32 apart (the 4 threads of a row read 128
  // contiguous bytes per group).  The sources are fp32 rows inside x / x_prev / the scalar tokens: 16-byte aligned for the big
  // MNIST tokens, 8 for the CIFAR-10 ones (D = 5370), 4 for the 257-wide scalar tokens -> the widest load the token allows
  // (workgroup-uniform choice): one dword per lane and instruction costs the vector-memory path as much as four.
  constexpr int NG = KC / 32;
  const int arow = tid >> 2, acol = (tid & 3) * 8;
  const int gb = b0 + arow;
  const float* srow = tk.src + (int64_t)(gb < g.B ? gb : g.B - 1) * tk.src_ld;
  const int avec = (((uintptr_t)tk.src % 16) == 0 && tk.src_ld % 4 == 0 && tk.k % 4 == 0) ? 4
                   : (((uintptr_t)tk.src % 8) == 0 && tk.src_ld % 2 == 0 && tk.k % 2 == 0) ? 2 : 1;
  struct AStage {
    float v[NG][8];
    int k0;
  };
  const int kw = (int)tk.w_ld;                     // weight rows are valid (zero padded) up to w_ld
  const int nchunk = (tk.k + KC - 1) / KC, last = nchunk - 1;
  // the whole K loop once per load width (no branch on it inside the loop)
  auto run = [&](auto avec_c) {
    constexpr int AVEC = decltype(avec_c)::value;
    // loads only (clamped addresses); out-of-range elements are zeroed in store_a
    auto load_a = [&](AStage& st, int k0) {
      st.k0 = k0;
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        const int kq = k0 + q * 32 + acol;
#pragma unroll
        for (int h = 0; h < 8 / AVEC; ++h) {
          const int k = kq + AVEC * h;             // k % AVEC == 0 and tk.k % AVEC == 0: a vector is all inside or all outside
          const float* p = srow + (k < tk.k ? k : 0);
          if constexpr (AVEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            st.v[q][4 * h] = t.x; st.v[q][4 * h + 1] = t.y; st.v[q][4 * h + 2] = t.z; st.v[q][4 * h + 3] = t.w;
          } else if constexpr (AVEC == 2) {
            const float2 t = *reinterpret_cast<const float2*>(p);
            st.v[q][2 * h] = t.x; st.v[q][2 * h + 1] = t.y;
          } else {
            st.v[q][h] = *p;
          }
        }
      }
    };
    auto store_a = [&](const AStage& st) {
      __builtin_amdgcn_sched_barrier(0);  // as in Stage::store
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        float e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = st.k0 + q * 32 + acol + i < tk.k ? st.v[q][i] : 0.f;
        char* dst = ldsA + arow * P + (q * 32 + acol) * sizeof(T);
        if constexpr (sizeof(T) == 2) {
          u32x4 pk;
          pk.x = pack_bf16x2(e[0], e[1]); pk.y = pack_bf16x2(e[2], e[3]);
          pk.z = pack_bf16x2(e[4], e[5]); pk.w = pack_bf16x2(e[6], e[7]);
          *reinterpret_cast<u32x4*>(dst) = pk;
        } else {
          reinterpret_cast<float4*>(dst)[0] = make_float4(e[0], e[1], e[2], e[3]);
          reinterpret_cast<float4*>(dst)[1] = make_float4(e[4], e[5], e[6], e[7]);
        }
      }
    };
    // two register stages: the loads of chunk c + 2 are issued before the MFMAs of chunk c (one chunk ahead left a whole
    // memory latency exposed per chunk).  Past the last chunk the last one is simply loaded again: no branch around a load.
    AStage a0, a1;
    Stage<T, KC, TN> w0, w1;
    load_a(a0, 0);
    w0.load((const T*)tk.w, tk.w_ld, n0, g.d, 0, kw, tid);
    load_a(a1, min(1, last) * KC);
    w1.load((const T*)tk.w, tk.w_ld, n0, g.d, min(1, last) * KC, kw, tid);
    // (pairs of chunks, then the odd one: a `break` between the halves would enter the loop header with the two stages'
    // loads in the other order, and the wait insertion would again assume the worst)
    for (int c = 0; c + 1 < nchunk; c += 2) {
      store_a(a0);
      w0.store(ldsW, tid);
      __syncthreads();
      load_a(a0, min(c + 2, last) * KC);
      w0.load((const T*)tk.w, tk.w_ld, n0, g.d, min(c + 2, last) * KC, kw, tid);
      mma_chunk<T, KC, NJ>(ldsA, ldsW, wr, wc, lane, acc);
      __syncthreads();
      store_a(a1);
      w1.store(ldsW, tid);
      __syncthreads();
      load_a(a1, min(c + 3, last) * KC);
      w1.load((const T*)tk.w, tk.w_ld, n0, g.d, min(c + 3, last) * KC, kw, tid);
      mma_chunk<T, KC, NJ>(ldsA, ldsW, wr, wc, lane, acc);
      __syncthreads();
    }
    if (nchunk & 1) {                               // stage 0 holds the last chunk
      store_a(a0);
      w0.store(ldsW, tid);
      __syncthreads();
      mma_chunk<T, KC, NJ>(ldsA, ldsW, wr, wc, lane, acc);
    }
  };
  if (avec == 4) run(std::integral_constant<int, 4>{});
  else if (avec == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 1>{});
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n0 + wc * 32 * NJ + j * 32 + r;
    if (n >= g.d) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int b = b0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (b < g.B) g.out[((int64_t)b * g.n_tok + tk.slot) * g.d + n] = acc[j][e];
    }
  }
}

// d % KC == 0 (host)
template <typename T, int KC, int NJ>
__global__ __launch_bounds__(256) void decode_kernel(DecArgs g) {
  constexpr int P = Pitch<T, KC>::value, TN = 64 * NJ;
  __shared__ __attribute__((aligned(16))) char smem[(TB + TN) * P];
  char* ldsA = smem;
  char* ldsW = smem + TB * P;
  const DecTok tk = g.tok[blockIdx.z];
  const int n0 = blockIdx.x * TN;
  if (n0 >= tk.n) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int b0 = blockIdx.y * TB;
  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  // features of this token: rows b of feat[B, T, d] = a matrix of row stride T * d starting at token * d
  const T* feat = (const T*)g.feat + (int64_t)blockIdx.z * g.d;
  const int64_t fld = (int64_t)g.T * g.d;
  Stage<T, KC, TB, false> f0, f1;
  Stage<T, KC, TN, false> w0, w1;
  const int nchunk = g.d / KC;
  const int last = nchunk - 1;               // no branch around any load (see encode_kernel)
  f0.load(feat, fld, b0, g.B, 0, g.d, tid);
  w0.load((const T*)tk.w, g.d, n0, tk.n, 0, g.d, tid);
  f1.load(feat, fld, b0, g.B, min(1, last) * KC, g.d, tid);
  w1.load((const T*)tk.w, g.d, n0, tk.n, min(1, last) * KC, g.d, tid);
  for (int c = 0; c + 1 < nchunk; c += 2) {  // two register stages, pairs of chunks then the odd one, as in encode_kernel
    f0.store(ldsA, tid);
    w0.store(ldsW, tid);
    __syncthreads();
    f0.load(feat, fld, b0, g.B, min(c + 2, last) * KC, g.d, tid);
    w0.load((const T*)tk.w, g.d, n0, tk.n, min(c + 2, last) * KC, g.d, tid);
    mma_chunk<T, KC, NJ>(ldsA, ldsW, wr, wc, lane, acc);
    __syncthreads();
    f1.store(ldsA, tid);
    w1.store(ldsW, tid);
    __syncthreads();
    f1.load(feat, fld, b0, g.B, min(c + 3, last) * KC, g.d, tid);
    w1.load((const T*)tk.w, g.d, n0, tk.n, min(c + 3, last) * KC, g.d, tid);
    mma_chunk<T, KC, NJ>(ldsA, ldsW, wr, wc, lane, acc);
    __syncthreads();
  }
  if (nchunk & 1) {
    f0.store(ldsA, tid);
    w0.store(ldsW, tid);
    __syncthreads();
    mma_chunk<T, KC, NJ>(ldsA, ldsW, wr, wc, lane, acc);
  }
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = n0 + wc * 32 * NJ + j * 32 + r;
    if (n >= tk.n) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int b = b0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (b < g.B) {
        const int64_t o = (int64_t)b * g.D + tk.off + n;
        g.out[o] = acc[j][e] + (g.resid ? g.resid[o] : 0.f);
      }
    }
  }
}

}  // namespace

// 32-column MFMA tiles per wave: 2 (128 output columns per workgroup) when that already gives the chip a workgroup per CU, else 1
// (64 columns, twice the workgroups).  These kernels are latency-bound chains of K chunks: at B = 64 the decoder launched 66
// workgroups of 128 columns on 256 CUs (29 us for 24.5 MB of weights).
static int pick_nj(int wgs128) {
  static const int forced = [] { const char* e = getenv("GPT_TOKENS_NJ"); return e ? atoi(e) : 0; }();   // 1 / 2: force (A/B)
  if (forced == 1 || forced == 2) return forced;
  int dev = 0, n_cu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  return wgs128 >= n_cu ? 2 : 1;
}

extern "C" int gpt_scalar_tokens(const int64_t* t, const void* loss_target, const void* loss_prev, int loss_is_f64,
                                 const float* freqs, int nfreq, int n_scalar, float* out, int B, void* stream) {
  GPT_REQUIRE(t && loss_target && loss_prev && freqs && out, "gpt_scalar_tokens: null operand");
  GPT_REQUIRE(B > 0 && nfreq > 0 && (n_scalar == 2 || n_scalar == 4), "gpt_scalar_tokens: bad sizes");
  const int64_t total = (int64_t)B * n_scalar * (2 * nfreq + 1);
  hipLaunchKernelGGL(scalar_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t,
                     loss_target, loss_prev, loss_is_f64, freqs, nfreq, n_scalar, out, B);
  return gpt_launch_status("gpt_scalar_tokens");
}

extern "C" int gpt_encode_tokens_subset(int dtype, int n_sel, const float* const* host_src, const int64_t* host_src_ld,
                                        const void* const* host_w, const int64_t* host_w_ld, const int* host_k,
                                        const int* host_slot, int n_tok, float* out, int B, int d, void* stream) {
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_encode_tokens: bad dtype %d", dtype);
  GPT_REQUIRE(n_tok > 0 && n_tok <= MAX_TOK && n_sel > 0 && n_sel <= n_tok, "gpt_encode_tokens: n_tok=%d / n_sel=%d out of range (max %d)",
              n_tok, n_sel, MAX_TOK);
  GPT_REQUIRE(B > 0 && d > 0 && d % 8 == 0 && out, "gpt_encode_tokens: bad sizes");
  const int esz = dtype == GPT_BF16 ? 2 : 4;
  EncArgs g;
  for (int i = 0; i < n_sel; ++i) {
    GPT_REQUIRE(host_src[i] && host_w[i] && host_k[i] > 0, "gpt_encode_tokens: token %d: null/empty", i);
    const int kp = (host_k[i] + KC_PAD - 1) / KC_PAD * KC_PAD;
    GPT_REQUIRE(host_w_ld[i] >= kp && (host_w_ld[i] * esz) % 16 == 0 && ((uintptr_t)host_w[i] % 16) == 0,
                "gpt_encode_tokens: token %d: weight rows must be zero-padded to a multiple of %d and 16-byte aligned", i, KC_PAD);
    GPT_REQUIRE(host_src_ld[i] >= host_k[i], "gpt_encode_tokens: token %d: src_ld < k", i);
    const int slot = host_slot ? host_slot[i] : i;
    GPT_REQUIRE(slot >= 0 && slot < n_tok, "gpt_encode_tokens: token %d: output slot %d out of range", i, slot);
    g.tok[i] = EncTok{host_src[i], host_w[i], host_src_ld[i], host_w_ld[i], host_k[i], slot};
  }
  g.out = out; g.n_tok = n_tok; g.B = B; g.d = d;
  const int nj = pick_nj((d + 127) / 128 * ((B + TB - 1) / TB) * n_sel);
  const dim3 grid((d + 64 * nj - 1) / (64 * nj), (B + TB - 1) / TB, n_sel), block(256);
  static const bool wide = [] { const char* e = getenv("GPT_TOKENS_KC128"); return !e || atoi(e) != 0; }();   // 0: r01's chunks (A/B)
#define ENC(T, KC)                                                                                          \
  do {                                                                                                      \
    if (nj == 1) hipLaunchKernelGGL((encode_kernel<T, KC, 1>), grid, block, 0, (hipStream_t)stream, g);     \
    else hipLaunchKernelGGL((encode_kernel<T, KC, 2>), grid, block, 0, (hipStream_t)stream, g);             \
  } while (0)
  if (dtype == GPT_BF16 && wide) ENC(bf16_t, 128);
  else if (dtype == GPT_BF16) ENC(bf16_t, 32);
  else ENC(float, 32);
#undef ENC
  return gpt_launch_status("gpt_encode_tokens");
}

extern "C" int gpt_encode_tokens(int dtype, int n_tok, const float* const* host_src, const int64_t* host_src_ld,
                                 const void* const* host_w, const int64_t* host_w_ld, const int* host_k, float* out,
                                 int B, int d, void* stream) {
  return gpt_encode_tokens_subset(dtype, n_tok, host_src, host_src_ld, host_w, host_w_ld, host_k, nullptr, n_tok, out, B, d, stream);
}

extern "C" int gpt_decode_tokens(int dtype, int n_tok, const void* feat, int T, const void* const* host_w,
                                 const int* host_n, const int* host_off, const float* resid, float* out, int B, int d,
                                 int D, void* stream) {
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_decode_tokens: bad dtype %d", dtype);
  GPT_REQUIRE(n_tok > 0 && n_tok <= MAX_TOK && n_tok <= T, "gpt_decode_tokens: n_tok=%d out of range", n_tok);
  GPT_REQUIRE(B > 0 && d > 0 && d % 32 == 0 && D > 0 && feat && out, "gpt_decode_tokens: bad sizes (d must be a multiple of 32)");
  GPT_REQUIRE(((uintptr_t)feat % 16) == 0, "gpt_decode_tokens: feat must be 16-byte aligned");
  DecArgs g;
  int max_n = 0;
  for (int j = 0; j < n_tok; ++j) {
    GPT_REQUIRE(host_w[j] && host_n[j] > 0 && host_off[j] >= 0 && host_off[j] + host_n[j] <= D,
                "gpt_decode_tokens: token %d: bad descriptor", j);
    GPT_REQUIRE(((uintptr_t)host_w[j] % 16) == 0, "gpt_decode_tokens: token %d: weight must be 16-byte aligned", j);
    g.tok[j] = DecTok{host_w[j], host_n[j], host_off[j]};
    max_n = host_n[j] > max_n ? host_n[j] : max_n;
  }
  g.feat = feat; g.resid = resid; g.out = out; g.n_tok = n_tok; g.T = T; g.B = B; g.d = d; g.D = D;
  const int nj = pick_nj((max_n + 127) / 128 * ((B + TB - 1) / TB) * n_tok);
  const dim3 grid((max_n + 64 * nj - 1) / (64 * nj), (B + TB - 1) / TB, n_tok), block(256);
  static const bool wide = [] { const char* e = getenv("GPT_TOKENS_KC128"); return !e || atoi(e) != 0; }();
#define DEC(T, KC)                                                                                          \
  do {                                                                                                      \
    if (nj == 1) hipLaunchKernelGGL((decode_kernel<T, KC, 1>), grid, block, 0, (hipStream_t)stream, g);     \
    else hipLaunchKernelGGL((decode_kernel<T, KC, 2>), grid, block, 0, (hipStream_t)stream, g);             \
  } while (0)
  if (dtype == GPT_BF16 && wide && d % 128 == 0) DEC(bf16_t, 128);
  else if (dtype == GPT_BF16) DEC(bf16_t, 32);
  else DEC(float, 32);
#undef DEC
  return gpt_launch_status("gpt_decode_tokens");
}
