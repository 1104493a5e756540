// Shared device/host helpers for the gfx950 G.pt kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/gpt_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // one 32x32x16 / 16x16x32 bf16 MFMA operand
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

void gpt_set_error(const char* fmt, ...);

#define GPT_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      gpt_set_error(__VA_ARGS__);       \
      return -1;                        \
    }                                   \
  } while (0)

// Per-device "done once" state.  One process per GPU is the intended use, but nothing in the C ABI forbids a caller that drives
// several devices: whatever is set up once (LDS grants of a kernel, event pools) is keyed by the CURRENT device.
#define GPT_MAX_DEVICES 16
static inline int gpt_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) d = 0;
  return d;
}
struct GptPerDevice {
  bool done[GPT_MAX_DEVICES] = {};
  bool& here() {
    static thread_local bool never;   // device index out of range: set up again on every call (cheap, always right)
    const int d = gpt_device();
    if (d < 0 || d >= GPT_MAX_DEVICES) return never = false;
    return done[d];
  }
};

static inline int gpt_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    gpt_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN stays NaN: one v_cvt_pk_bf16_f32 (the integer-arithmetic form costs ~7 VALU per element,
// which is what bound the GEMM epilogues)
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T>
struct elem;
template <>
struct elem<float> {
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct elem<bf16_t> {
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU, as nn.GELU() (Gpt/models/transformer.py:64)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// gpt_gemm_tn_grouped with one more (internal) argument: sumsq != NULL (store mode only) adds the sum of squares of every stored
// tile to that device scalar (gpt_backward_t.wgrad_sumsq)
int gpt_tn_grouped_sumsq(int n, const void* const* P, const int64_t* ldp, const void* const* Q, const int64_t* ldq, float* const* C,
                         const int64_t* ldc, const int* I, const int* J, int R, int accumulate, float* sumsq, void* stream);
// ... and with one contraction length per problem (store mode, or accumulate mode where all lengths are equal)
int gpt_tn_grouped_sumsq_r(int n, const void* const* P, const int64_t* ldp, const void* const* Q, const int64_t* ldq, float* const* C,
                           const int64_t* ldc, const int* I, const int* J, const int* R, int accumulate, float* sumsq, void* stream);

// gpt_randn (stream (seed, sid) over a [B, D] buffer) + gpt_posterior_step with every row at timestep t, in one kernel (the DDPM
// sampling loop's step; bit-identical to the two calls)
int gpt_posterior_step_philox(const float* model_out, const float* x_t, int64_t t, const float* coef, int predict_xstart, int use_clamp,
                              float lo, float hi, uint64_t seed, uint64_t sid, float* x_out, int B, int D, int64_t* t_next,
                              int64_t t_next_value, void* stream);   // t_next != NULL: also t_next[0..B) = t_next_value

// gpt_layernorm_bwd_colsum with a partials buffer: part != NULL ([<= 256][3][d] floats) receives every workgroup's dgamma / dbeta /
// gcol sums by plain stores instead of float atomics into the outputs, *n_blocks the number of rows written; the caller adds them up
// with gpt_ln_partials_reduce (one launch for many layernorm launches, fixed order: reproducible sums)
#define GPT_LN_PART_MAX_CALLS 28
struct LnPartCalls {
  struct {
    const float* part;
    float* out[3];     // dgamma, dbeta, gcol (NULL = none)
    int nblocks;
  } c[GPT_LN_PART_MAX_CALLS];
  int n = 0;
};
int gpt_layernorm_bwd_part(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* resid, float* gout, void* gout_t, int gout_t_dtype, float* dgamma, float* dbeta, float* gcol,
                           float* part, int* n_blocks, int rows, int d, void* stream);
int gpt_ln_partials_reduce(const LnPartCalls& calls, int d, void* stream);

// ln_in (+ position embedding) and the first Block's ln1 in one launch (inference forward), for the widths _has() names
bool gpt_layernorm_pair_has(int d);
int gpt_layernorm_pair(const float* x, const float* g0, const float* b0, const float* add, int add_rows, float* x0, const float* g1,
                       const float* b1, void* y, int out_dtype, int rows, int d, float eps, void* stream);

// gpt_attention storing only the first t_out query rows of every sample, compactly ([B, t_out, d]); gather of the matching fp32
// residual rows (the pruned last Block of an inference forward, model.cpp)
int gpt_attention_rows(int dtype, const void* qkv, void* y, int B, int T, int t_out, int n_head, int hd, void* stream);
int gpt_gather_token_rows(const float* src, float* dst, int B, int T, int t_out, int d, void* stream);
// the inverse: dst[b * T + r, :] = r < t_out ? src[b * t_out + r, :] : 0, rows of d elements of elem_bytes (2 or 4) each
int gpt_scatter_token_rows(const void* src, void* dst, int B, int T, int t_out, int d, int elem_bytes, void* stream);

// gpt_small_gemm_batched with one more (internal) argument: bf16_mma != 0 rounds both operands to bf16 and multiplies them on the
// bf16 matrix pipe (the denoiser backward in bf16 compute mode); 0 = the exact-fp32 pipe
int gpt_small_gemm_batched_mode(int n_problems, int a_dtype, const void* const* host_A, int64_t a_sm, int64_t a_sk, int b_dtype,
                                const void* const* host_B, int64_t b_sk, int64_t b_sn, float* const* host_C, const int* host_ldc,
                                const int* host_M, const int* host_N, const int* host_K, int accumulate, int bf16_mma,
                                void* stream);

// gpt_gemm_nt with one more (internal) argument: colsum != NULL (GPT_EPI_DGELU only) += the column sums of the output, i.e. the
// bias gradient of the Linear whose output gradient the call produces, from the epilogue instead of a second pass (fp32 atomics)
int gpt_gemm_nt_colsum(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                       void* aux, int64_t ldaux, int M, int N, int K, int epilogue, float* colsum, void* stream);
