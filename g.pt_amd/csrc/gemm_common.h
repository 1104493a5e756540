// Shared pieces of the MFMA GEMM kernels (gemm.hip: 64-row wave tiles; gemm256.hip: 256 x 256 block tile).
#pragma once
#include "common.h"

namespace gemmk {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T>
struct GemmArgs {
  const T* A;
  const T* W;
  const float* bias;
  void* C;
  void* aux;
  int64_t lda, ldw, ldc, ldaux;
  int M, N, K;
  // K spans (gemm256 only, GPT_EPI_STORE_F32 into a workspace): workgroup (span, tile) contracts K-tiles
  // [span * ceil(nk / ksplit), ...) and stores its partial tile at C + span * M * ldc
  int ksplit = 1;
  // GPT_EPI_DGELU only: += the column sums of the output (the bias gradient of the Linear whose output gradient this GEMM
  // produces: its separate colsum pass reads the whole [M, 4d] matrix again), fp32 atomics, one per column and wave; or NULL
  float* colsum = nullptr;
};

template <typename T>
__device__ __forceinline__ void store4(T* p, const float4& v) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(p) = v;
  } else {
    uint2 o;
    o.x = pack_bf16x2(v.x, v.y);
    o.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = o;
  }
}
template <typename T>
__device__ __forceinline__ float4 load4(const T* p) {
  if constexpr (sizeof(T) == 4) {
    return *reinterpret_cast<const float4*>(p);
  } else {
    const uint2 o = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16),
                       __uint_as_float(o.y & 0xffff0000u));
  }
}
// bf16 forward GELU: x * sigmoid(a x + b x^3 + c x^5), coefficients fitted (minimax over [-8, 8]) to the EXACT erf GELU of
// nn.GELU(): |error| <= 2.6e-5 absolute, i.e. below half a bf16 ulp of the result wherever |gelu| >= 0.013 and below the
// bf16 rounding of the activations it is summed with elsewhere.  fp32 mode: libm erff (parity).
// Raw v_rcp_f32 / v_exp_f32 (1 ulp each): __frcp_rn expands to the full IEEE division sequence, ~10 instructions, and made
// the GELU epilogue 40 k cycles.  Evaluated on a pair of elements with packed fp32 VALU (v_pk_mul_f32 / v_pk_fma_f32 /
// v_pk_add_f32: two elements per issue slot).  The GELU epilogue is VALU-bound (128 elements per lane after every tile, no MFMA
// beside it): per element 3 packed + min + exp + rcp issue slots.  The range guard sits on x^2
// (min(x^2, 64)): beyond |x| = 8 the exponent is then x * P(64) = -4.98 x, which saturates the sigmoid to 0 / 1
// (2^40 and up, or inf: rcp gives 0) - covered by test_gemm_gelu_bf16_range.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  constexpr float A = -1.5950157686361435f * 1.4426950408889634f, B = -0.07401129203229191f * 1.4426950408889634f,
                  Cc = 0.000703033580348536f * 1.4426950408889634f;
  f32x2 x2 = x * x;
  x2.x = fminf(x2.x, 64.0f);
  x2.y = fminf(x2.y, 64.0f);
  const f32x2 p = __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, f32x2{Cc, Cc}, f32x2{B, B}), f32x2{A, A});
  const f32x2 q = x * p;
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)} + f32x2{1.0f, 1.0f};
  return x * f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
}
// d/dx of the SAME fitted function, gelu(x) ~ x s(x), s = sigmoid(p(x)), p = a x + b x^3 + c x^5:
//   gelu'(x) ~ s (1 + x (1 - s) p'(x)),  p' = a + 3 b x^2 + 5 c x^4
// |error| <= 1.1e-4 against Phi(x) + x phi(x) over the whole line (x^2 clamped at 64 as above: s -> 0 / 1 beyond), i.e. below the
// bf16 rounding of the gradients it multiplies; 8 issue slots per element with the packed fp32 VALU against ~20 for the
// erf-based form (the DGELU epilogue is VALU-bound like the GELU one).  bf16 training mode only; fp32 mode keeps libm.
__device__ __forceinline__ f32x2 gelu_fast_grad2(f32x2 x) {
  constexpr float L2E = 1.4426950408889634f;
  constexpr float a = 1.5950157686361435f, b = 0.07401129203229191f, c = -0.000703033580348536f;
  constexpr float A = -a * L2E, B = -b * L2E, Cc = -c * L2E;
  f32x2 x2 = x * x;
  x2.x = fminf(x2.x, 64.0f);
  x2.y = fminf(x2.y, 64.0f);
  const f32x2 q = x * __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, f32x2{Cc, Cc}, f32x2{B, B}), f32x2{A, A});
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)} + f32x2{1.0f, 1.0f};
  const f32x2 sg = f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
  const f32x2 pp = __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, f32x2{5.0f * c, 5.0f * c}, f32x2{3.0f * b, 3.0f * b}), f32x2{a, a});
  const f32x2 u = x * pp;
  const f32x2 w = __builtin_elementwise_fma(-sg, u, u);     // x p' (1 - s)
  return __builtin_elementwise_fma(sg, w, sg);
}
template <typename T>
__device__ __forceinline__ float4 gelu4(const float4& v) {
  if constexpr (sizeof(T) == 4) {
    return make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
  } else {
    const f32x2 a = gelu_fast2(f32x2{v.x, v.y}), b = gelu_fast2(f32x2{v.z, v.w});
    return make_float4(a.x, a.y, b.x, b.y);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Wave-level epilogue shared by the kernels.  A wave owns NI x 2 MFMA 32x32 accumulator tiles = (32 NI) x 64
// outputs.  32 rows at a time go through a PRIVATE LDS patch [32][68] fp32 (8.5 KiB, padded pitch: the
// 16-byte row reads are conflict-free), after which each lane holds 8 consecutive columns of one row:
// bf16 outputs leave as ONE 16-byte store per lane and every row of the wave's 64 columns is written as a
// full 128-byte line (fp32: 256 bytes = two lines).  Only the wave itself touches its patch: no workgroup
// barrier.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// Ordering of the wave's own patch writes and reads: the LDS executes one wave's DS instructions in issue order, so only
// the COMPILER must be kept from moving them across this point.  (A wavefront-scope fence is lowered by hipcc to
// s_waitcnt vmcnt(0): it drained the global stores 16 times per tile and made the epilogue 19 k cycles long.)
#define WAVE_LDS_ORDER()                                   \
  do {                                                     \
    asm volatile("" ::: "memory");                         \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)
constexpr int PATCH_PITCH = 68;
constexpr int PATCH_BYTES = 32 * PATCH_PITCH * 4;

template <typename T>
__device__ __forceinline__ void store8(T* p, const float4& a, const float4& b) {
  if constexpr (sizeof(T) == 4) {
    reinterpret_cast<float4*>(p)[0] = a;
    reinterpret_cast<float4*>(p)[1] = b;
  } else {
    uint4 o;
    o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w);
    o.z = pack_bf16x2(b.x, b.y); o.w = pack_bf16x2(b.z, b.w);
    *reinterpret_cast<uint4*>(p) = o;
  }
}
template <typename T>
__device__ __forceinline__ void load8(const T* p, float4& a, float4& b) {
  if constexpr (sizeof(T) == 4) {
    a = reinterpret_cast<const float4*>(p)[0];
    b = reinterpret_cast<const float4*>(p)[1];
  } else {
    const uint4 o = *reinterpret_cast<const uint4*>(p);
    a = make_float4(__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u));
    b = make_float4(__uint_as_float(o.z << 16), __uint_as_float(o.z & 0xffff0000u), __uint_as_float(o.w << 16), __uint_as_float(o.w & 0xffff0000u));
  }
}
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <typename T>
__device__ __forceinline__ float4 dgelu4(const float4& v, const float4& pre) {
  if constexpr (sizeof(T) == 4)
    return make_float4(v.x * gelu_erf_grad(pre.x), v.y * gelu_erf_grad(pre.y), v.z * gelu_erf_grad(pre.z), v.w * gelu_erf_grad(pre.w));
  else {
    const f32x2 g0 = gelu_fast_grad2(f32x2{pre.x, pre.y}), g1 = gelu_fast_grad2(f32x2{pre.z, pre.w});
    return make_float4(v.x * g0.x, v.y * g0.y, v.z * g1.x, v.w * g1.y);
  }
}

// What an epilogue READS for 8 consecutive columns of one row (the residual, the running sum or the pre-activation).
// Loaded one 32-row step ahead of its use so that no wait on it ever covers the stores issued after it.
template <int EPI>
constexpr bool epi_reads = EPI == GPT_EPI_RESADD || EPI == GPT_EPI_ACCUM_F32 || EPI == GPT_EPI_DGELU;
struct EpiPre {
  float4 a, b;
};
template <typename T, int EPI>
__device__ __forceinline__ void epi_load8(const GemmArgs<T>& g, int m, int n, EpiPre& p) {
  const int64_t ci = (int64_t)m * g.ldc + n;
  const int64_t ai = (int64_t)m * g.ldaux + n;
  if constexpr (EPI == GPT_EPI_RESADD) {
    const float* rp = g.aux ? (const float*)g.aux + ai : (const float*)g.C + ci;
    p.a = reinterpret_cast<const float4*>(rp)[0];
    p.b = reinterpret_cast<const float4*>(rp)[1];
  } else if constexpr (EPI == GPT_EPI_ACCUM_F32) {
    const float* c = (const float*)g.C + ci;
    p.a = reinterpret_cast<const float4*>(c)[0];
    p.b = reinterpret_cast<const float4*>(c)[1];
  } else if constexpr (EPI == GPT_EPI_DGELU) {
    load8<T>((const T*)g.aux + ai, p.a, p.b);
  }
}

// 8 consecutive columns of row m starting at column n (n % 8 == 0; m < M, n < N; N % 8 == 0); a | b = acc + bias
template <typename T, int EPI>
__device__ __forceinline__ void epi_store8(const GemmArgs<T>& g, int m, int n, const float4& a, const float4& b, const EpiPre& p) {
  const int64_t ci = (int64_t)m * g.ldc + n;
  const int64_t ai = (int64_t)m * g.ldaux + n;
  if constexpr (EPI == GPT_EPI_STORE) {
    store8<T>((T*)g.C + ci, a, b);
  } else if constexpr (EPI == GPT_EPI_GELU) {
    store8<T>((T*)g.C + ci, gelu4<T>(a), gelu4<T>(b));
  } else if constexpr (EPI == GPT_EPI_RESADD || EPI == GPT_EPI_ACCUM_F32) {
    store8<float>((float*)g.C + ci, add4(p.a, a), add4(p.b, b));
  } else if constexpr (EPI == GPT_EPI_STORE_F32) {
    store8<float>((float*)g.C + ci, a, b);
  } else if constexpr (EPI == GPT_EPI_GELU_AUX) {
    store8<T>((T*)g.aux + ai, a, b);
    store8<T>((T*)g.C + ci, gelu4<T>(a), gelu4<T>(b));
  } else if constexpr (EPI == GPT_EPI_DGELU) {
    store8<T>((T*)g.C + ci, dgelu4<T>(a, p.a), dgelu4<T>(b, p.b));
  }
}

// to_patch(step): writes rows [32 step, 32 step + 32) x 64 columns of the wave's accumulators to patch[row][col]
// ROWS <= 32 NI: rows of the wave's block that exist (a multiple of 8; the last step may be partly empty)
template <typename T, int EPI, int NI, int ROWS = 32 * NI, typename ToPatch>
__device__ __forceinline__ void epilogue_wave(const GemmArgs<T>& g, ToPatch&& to_patch, float* patch, int mrow0,
                                              int ncol0, int lane) {
  const int pr = lane >> 3, pc8 = (lane & 7) * 8;
  const int n = ncol0 + pc8;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (g.bias && n < g.N) {
    b0 = reinterpret_cast<const float4*>(g.bias + n)[0];
    b1 = reinterpret_cast<const float4*>(g.bias + n)[1];
  }
  // the bias is consumed HERE as far as the compiler can tell: otherwise its first use sits inside the steps below and
  // hipcc's wait for it (s_waitcnt vmcnt(0), repeated after every branch) drains the stores of every step
  asm volatile("" : "+v"(b0.x), "+v"(b0.y), "+v"(b0.z), "+v"(b0.w), "+v"(b1.x), "+v"(b1.y), "+v"(b1.z), "+v"(b1.w));
  // column sums of the stored values (GemmArgs::colsum): this lane's 8 columns over the rows it handles
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_cs = EPI == GPT_EPI_DGELU && g.colsum != nullptr;                 // workgroup-uniform
  auto store_rows = [&](int m, const float4& x0, const float4& x1, const EpiPre& pre) {
    if constexpr (EPI == GPT_EPI_DGELU) {
      const float4 o0 = dgelu4<T>(x0, pre.a), o1 = dgelu4<T>(x1, pre.b);
      store8<T>((T*)g.C + (int64_t)m * g.ldc + n, o0, o1);
      if (want_cs) {
        cs[0] += o0.x; cs[1] += o0.y; cs[2] += o0.z; cs[3] += o0.w;
        cs[4] += o1.x; cs[5] += o1.y; cs[6] += o1.z; cs[7] += o1.w;
      }
    } else {
      epi_store8<T, EPI>(g, m, n, x0, x1, pre);
    }
  };
  auto flush_cs = [&]() {
    if constexpr (EPI == GPT_EPI_DGELU) {
      if (!want_cs) return;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cs[j] += __shfl_xor(cs[j], 8, 64);
        cs[j] += __shfl_xor(cs[j], 16, 64);
        cs[j] += __shfl_xor(cs[j], 32, 64);
      }
      if (pr == 0 && n < g.N) {                    // lanes 0..7: one hardware float atomic per column (N % 8 == 0)
#pragma unroll
        for (int j = 0; j < 8; ++j) unsafeAtomicAdd(g.colsum + n + j, cs[j]);
      }
    }
  };
  // wave-uniform: the wave's whole (32 NI) x 64 block lies inside the matrix -> straight-line code, counted waits
  const bool full = __builtin_amdgcn_readfirstlane((mrow0 + ROWS <= g.M && ncol0 + 64 <= g.N) ? 1 : 0) != 0;
  if (full) {
    EpiPre pre[2][4];
    if constexpr (epi_reads<EPI>) {
#pragma unroll
      for (int p = 0; p < 4; ++p) epi_load8<T, EPI>(g, mrow0 + p * 8 + pr, n, pre[0][p]);   // ROWS >= 32
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      to_patch(i);
      WAVE_LDS_ORDER();
      if constexpr (epi_reads<EPI>) {
        if (i + 1 < NI) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if ((i + 1) * 32 + p * 8 < ROWS) epi_load8<T, EPI>(g, mrow0 + (i + 1) * 32 + p * 8 + pr, n, pre[(i + 1) & 1][p]);
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (i * 32 + p * 8 >= ROWS) continue;
        const int row = p * 8 + pr;
        const float4 v0 = *reinterpret_cast<const float4*>(patch + row * PATCH_PITCH + pc8);
        const float4 v1 = *reinterpret_cast<const float4*>(patch + row * PATCH_PITCH + pc8 + 4);
        store_rows(mrow0 + i * 32 + row, add4(v0, b0), add4(v1, b1), pre[i & 1][p]);
      }
      WAVE_LDS_ORDER();
    }
    flush_cs();
    return;
  }
  // edge tiles: per-lane bounds
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    to_patch(i);
    WAVE_LDS_ORDER();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (i * 32 + p * 8 >= ROWS) continue;
      const int row = p * 8 + pr;
      const int m = mrow0 + i * 32 + row;
      const float4 v0 = *reinterpret_cast<const float4*>(patch + row * PATCH_PITCH + pc8);
      const float4 v1 = *reinterpret_cast<const float4*>(patch + row * PATCH_PITCH + pc8 + 4);
      if (m < g.M && n < g.N) {
        EpiPre pre;
        if constexpr (epi_reads<EPI>) epi_load8<T, EPI>(g, m, n, pre);
        store_rows(m, add4(v0, b0), add4(v1, b1), pre);
      }
    }
    WAVE_LDS_ORDER();
  }
  flush_cs();
}

// Optional per-launch timing with HIP events recorded on the launch stream (bench.py roofline pass).
struct GemmProf {
  bool on = false;
  int n = 0;
  static constexpr int CAP = 4096;
  hipEvent_t ev[CAP][2];
  double flops[CAP];
  bool made = false;
};
extern GemmProf g_prof_tab[GPT_MAX_DEVICES];   // one event pool per device (events belong to the device they were created on)
#define g_prof (gemmk::g_prof_tab[(unsigned)gpt_device() % GPT_MAX_DEVICES])

}  // namespace gemmk
