// Backward kernels of the denoiser that are not plain GEMMs (those reuse gpt_gemm_nt on transposed operands):
// LayerNorm backward, short-sequence attention backward, column sums (bias / position-embedding gradients)
// and a small strided fp32 GEMM for the ragged encoder / decoder gradients.
// They implement what autograd derives for Gpt/models/transformer.py:34-51 (attention), :69-72 (Block),
// :215-242 (encoder / decoder) in the reference; parity is pinned by the gradient fixtures g6_training_*.
#include "gemm_common.h"
#include <stdlib.h>

namespace {
using gemmk::load8;

// ------------------------------------------------------------------------------------------ LayerNorm backward
// One wave per row, the row held in registers (lane l owns columns 256 ch + 4 l .. + 3 of chunk ch < NCH).
// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); the rows' contributions to dgamma / dbeta stay in
// registers until the wave has done all its rows, then: one LDS add per wave, one fp32 global atomic per column per
// workgroup (gamma / beta gradients are sums over all B*T rows).
//   resid != NULL: gout = resid + dx (the residual-stream gradient passes the LayerNorm branch point)
//   gout_t != NULL: also emits gout in the compute dtype (operand of the next dgrad / wgrad GEMMs)
template <typename T>
__device__ __forceinline__ f32x4 load_row4(const T* p);
template <>
__device__ __forceinline__ f32x4 load_row4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <>
__device__ __forceinline__ f32x4 load_row4<bf16_t>(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ void store_row4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store_row4(bf16_t* p, f32x4 v) {
  uint2 u;
  u.x = pack_bf16x2(v[0], v[1]);
  u.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
}

// LNB_WAVES waves per workgroup: few workgroups (the dgamma / dbeta flush is one fp32 atomic per column per workgroup, and atomics
// are slow) but enough waves per CU to hide the row loads.  8 of them have 256 registers per lane each: at NCH = 12 (n_embd 3072) the
// row and its three accumulator sets need about 400, and the 8-wave build spilled 293 registers to scratch (86.7 us per launch at
// M = 1768, 0.8 TB/s); that width runs 4 waves per workgroup, one per SIMD, 512 registers each.
template <int NCH>
constexpr int lnb_waves = NCH > 8 ? 4 : 8;
template <typename DyT, typename OutT, int NCH>
__global__ __launch_bounds__(lnb_waves<NCH> * 64) void layernorm_bwd_kernel(const DyT* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ resid,
                                                            float* __restrict__ gout, OutT* __restrict__ gout_t,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ gcol, float* __restrict__ part, int rows, int d,
                                                            int rows_per_block) {
  constexpr int LNB_WAVES = lnb_waves<NCH>;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [LNB_WAVES][d]: the waves' partial dgamma, then dbeta, then gcol
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // gcol (optional): column sums of gout = the bias gradient of the Linear whose output gradient gout is (the launch is at most
  // one workgroup per CU, so the third accumulator set costs no occupancy)
  f32x4 gam[NCH], pg[NCH], pb[NCH], pc[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c = ch * 256 + lane * 4;
    gam[ch] = c < d ? load_row4<float>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    pg[ch] = pb[ch] = pc[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int row_begin = blockIdx.x * rows_per_block;
  const int row_end = min(rows, row_begin + rows_per_block);
  const float inv_d = 1.0f / (float)d;
  for (int row = row_begin + wave; row < row_end; row += LNB_WAVES) {
    const float mu = mean[row], rs = rstd[row];
    const int64_t base = (int64_t)row * d;
    f32x4 xh[NCH], dyv[NCH], rv[NCH];
    float s1 = 0.f, s2 = 0.f;
    // all three operands of the row are requested before anything is computed (the residual used to be loaded after the two
    // wave reductions: a second memory latency per row, and a wave has only one or two rows at the small batches)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = ch * 256 + lane * 4;
      rv[ch] = (resid && c < d) ? load_row4<float>(resid + base + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = ch * 256 + lane * 4;
      if (c < d) {
        xh[ch] = load_row4<float>(x + base + c);
        dyv[ch] = load_row4<DyT>(dy + base + c);
      } else {
        xh[ch] = f32x4{mu, mu, mu, mu};
        dyv[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[ch][j] = (xh[ch][j] - mu) * rs;
        const float gdy = gam[ch][j] * dyv[ch][j];
        s1 += gdy;
        s2 += gdy * xh[ch][j];
      }
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = ch * 256 + lane * 4;
      f32x4 dx;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dx[j] = rs * (gam[ch][j] * dyv[ch][j] - s1 - xh[ch][j] * s2);
        pg[ch][j] += dyv[ch][j] * xh[ch][j];
        pb[ch][j] += dyv[ch][j];
      }
      if (c < d) {
        dx += rv[ch];
        store_row4(gout + base + c, dx);
        if (gout_t) store_row4(gout_t + base + c, dx);
        pc[ch] += dx;
      }
    }
  }
  // workgroup sums of the waves' partials: every wave stores its registers to its own LDS row [wave][d] (plain 16-byte stores),
  // then each thread adds the 8 rows of its columns and issues the one global atomic per column - first dgamma, then dbeta
  // through the same buffer.  (LDS float atomics here - 48 ds_add_f32 per lane from 8 waves on the same addresses - were a
  // 12 us tail of every launch.)
#ifdef GPT_LNB_NOFLUSH                               // diagnostic build only (results are garbage): the price of the flush below
  if (rows > 0) return;
#endif
  const int passes = gcol ? 3 : 2;                   // uniform over the grid
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    if (pass >= passes) break;
    if (pass) __syncthreads();
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = ch * 256 + lane * 4;
      if (c < d) *reinterpret_cast<f32x4*>(red + wave * d + c) = pass == 0 ? pg[ch] : pass == 1 ? pb[ch] : pc[ch];
    }
    __syncthreads();
    float* out = pass == 0 ? dgamma : pass == 1 ? dbeta : gcol;
    // part != NULL: the workgroup's sums are STORED to its own row of a partials buffer [grid][3][d] (gpt_ln_partials_reduce adds
    // them up later, in a fixed order): the 2 - 3 x d float atomics per workgroup on the same addresses were a third of this
    // kernel's time at M = 3328 (22.6 us with them, 15.2 without)
    float* prow = part ? part + ((int64_t)blockIdx.x * 3 + pass) * d : nullptr;
    for (int c = threadIdx.x; c < d; c += LNB_WAVES * 64) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) t += red[w * d + c];
      if (prow) prow[c] = t;
      else unsafeAtomicAdd(out + c, t);            // hardware global_atomic_add_f32 (plain atomicAdd is a CAS loop)
    }
  }
}

// ------------------------------------------------------------------------------------------ column sums
// out[c] += sum_r X[r, c]   (X: rows x cols, leading dimension ld).  A workgroup owns 256 columns x one row range:
// thread (cg = tid & 31, rl = tid >> 5) sums 8 adjacent columns (one 16-byte load of bf16, two of fp32) over rows
// rl, rl + 8, ...; the 8 row lanes meet in LDS and one hardware fp32 atomic per column leaves the workgroup.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t ld, float* __restrict__ out, int rows,
                                                     int cols, int rows_per_block, int vec) {
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cg * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (vec && c0 + 8 <= cols) {
    // four rows' loads in flight per thread (one per iteration kept a CU's bytes in flight too few for the HBM latency)
    int r = r0 + rl;
    for (; r + 24 < r1; r += 32) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) load8<T>(X + (int64_t)(r + 8 * u) * ld + c0, a[u], b[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[0] += a[u].x; s[1] += a[u].y; s[2] += a[u].z; s[3] += a[u].w;
        s[4] += b[u].x; s[5] += b[u].y; s[6] += b[u].z; s[7] += b[u].w;
      }
    }
    for (; r < r1; r += 8) {
      float4 a, b;
      load8<T>(X + (int64_t)r * ld + c0, a, b);
      s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
      s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
    }
  } else {
    for (int r = r0 + rl; r < r1; r += 8)
      for (int j = 0; j < 8; ++j)
        if (c0 + j < cols) s[j] += elem<T>::load(X + (int64_t)r * ld + c0 + j);
  }
  __shared__ float red[8][256 + 8];
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cg * 8 + j] = s[j];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    unsafeAtomicAdd(out + c, t);
  }
}

// ------------------------------------------------------------------------------------------ small strided GEMM
// C[m, n] (+)= sum_k A(m, k) * B(k, n), exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), arbitrary element strides: the
// ragged encoder / decoder gradients (0.4 % of the FLOPs; K = batch or a token width).  64 x 64 tile per workgroup
// (one 32 x 32 accumulator per wave), 32-deep chunks staged through LDS so either operand may be K- or M/N-major.
// An operand's [KC x 64] chunk is 8 elements per thread.  VEC: its M / N index is the contiguous one, base and row stride are
// 16-byte aligned and the extent is a multiple of the vector width (the products that contract over the batch, big tokens):
// 16-byte loads along the row.  Otherwise one element per load, laid out along whichever index is contiguous.
// r02: the staging is a two-stage register pipeline (the loads of chunk c + 2 are issued before the MFMAs of chunk c; r01 loaded,
// stored and multiplied one chunk at a time: one exposed memory latency per 32 of K) written so that hipcc's wait insertion can
// count (see tokens.hip): no branch around a load (clamped addresses), zeroing and widening at the LDS store, pairs of chunks.
constexpr int SG_KC = 32, SG_PITCH = 68;   // 68 floats: 16-byte aligned rows, conflict-free 32-lane reads

template <typename T, bool VEC>
struct SgStage {
  uint32_t r[8];
  int k0;
  // element e of this thread in scalar mode: flat index i = tid + 256 e of the chunk, along k if the operand is K-major
  __device__ __forceinline__ static void pos(int i, bool kmajor, int& kk, int& ii) {
    kk = kmajor ? (i & (SG_KC - 1)) : (i >> 6);
    ii = kmajor ? (i >> 5) : (i & 63);
  }
  __device__ __forceinline__ void load(const T* __restrict__ base, int64_t s_i, int64_t s_k, int i0, int I, int k0_, int K) {
    k0 = k0_;
    if constexpr (VEC) {
      constexpr int VE = 16 / sizeof(T), VPR = 64 / VE, NVEC = SG_KC * VPR / 256;
#pragma unroll
      for (int j = 0; j < NVEC; ++j) {
        const int i = threadIdx.x + 256 * j, kk = i / VPR, iv = (i % VPR) * VE;
        const int k = min(k0 + kk, K - 1), idx = i0 + iv < I ? i0 + iv : 0;
        const u32x4 raw = *reinterpret_cast<const u32x4*>(base + (int64_t)k * s_k + idx);
        r[4 * j] = raw.x; r[4 * j + 1] = raw.y; r[4 * j + 2] = raw.z; r[4 * j + 3] = raw.w;
      }
    } else {
      const bool kmajor = s_k == 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int kk, ii;
        pos(threadIdx.x + 256 * e, kmajor, kk, ii);
        const T* p = base + (int64_t)min(i0 + ii, I - 1) * s_i + (int64_t)min(k0 + kk, K - 1) * s_k;
        if constexpr (sizeof(T) == 4) r[e] = __float_as_uint(*p);
        else r[e] = (uint32_t)*p;                               // widened at the store: nothing here may consume the load
      }
    }
  }
  __device__ __forceinline__ void store(float (*S)[SG_PITCH], int64_t s_k, int i0, int I, int K) const {
    __builtin_amdgcn_sched_barrier(0);   // what follows consumes the loads: it must not be scheduled into the chunk before
    if constexpr (VEC) {
      constexpr int VE = 16 / sizeof(T), VPR = 64 / VE, NVEC = SG_KC * VPR / 256;
#pragma unroll
      for (int j = 0; j < NVEC; ++j) {
        const int i = threadIdx.x + 256 * j, kk = i / VPR, iv = (i % VPR) * VE;
        const bool ok = k0 + kk < K && i0 + iv < I;                 // I % VE == 0: a vector is all inside or all outside
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(&S[kk][iv]) = f32x4{ok ? __uint_as_float(r[4 * j]) : 0.f, ok ? __uint_as_float(r[4 * j + 1]) : 0.f,
                                                        ok ? __uint_as_float(r[4 * j + 2]) : 0.f, ok ? __uint_as_float(r[4 * j + 3]) : 0.f};
        } else {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] = ok ? __uint_as_float(r[4 * j + e] << 16) : 0.f;
            v[2 * e + 1] = ok ? __uint_as_float(r[4 * j + e] & 0xffff0000u) : 0.f;
          }
          *reinterpret_cast<f32x4*>(&S[kk][iv]) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(&S[kk][iv + 4]) = f32x4{v[4], v[5], v[6], v[7]};
        }
      }
    } else {
      const bool kmajor = s_k == 1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int kk, ii;
        pos(threadIdx.x + 256 * e, kmajor, kk, ii);
        S[kk][ii] = (k0 + kk < K && i0 + ii < I) ? __uint_as_float(sizeof(T) == 4 ? r[e] : r[e] << 16) : 0.f;
      }
    }
  }
};

// bf16 compute mode: the same chunk as bf16 in LDS, [KC][SG_P16] (k-major rows of 64 + 8 columns), for the bf16 matrix pipe
constexpr int SG_P16 = 72;
template <typename T, bool VEC>
__device__ __forceinline__ void sg_store_bf16(const SgStage<T, VEC>& st, unsigned short (*S)[SG_P16], int64_t s_k, int i0, int I, int K) {
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (VEC) {
    constexpr int VE = 16 / sizeof(T), VPR = 64 / VE, NVEC = SG_KC * VPR / 256;
#pragma unroll
    for (int j = 0; j < NVEC; ++j) {
      const int i = threadIdx.x + 256 * j, kk = i / VPR, iv = (i % VPR) * VE;
      const bool ok = st.k0 + kk < K && i0 + iv < I;
      if constexpr (sizeof(T) == 4) {
        uint2 o;
        o.x = ok ? pack_bf16x2(__uint_as_float(st.r[4 * j]), __uint_as_float(st.r[4 * j + 1])) : 0u;
        o.y = ok ? pack_bf16x2(__uint_as_float(st.r[4 * j + 2]), __uint_as_float(st.r[4 * j + 3])) : 0u;
        *reinterpret_cast<uint2*>(&S[kk][iv]) = o;
      } else {
        *reinterpret_cast<u32x4*>(&S[kk][iv]) = u32x4{ok ? st.r[4 * j] : 0u, ok ? st.r[4 * j + 1] : 0u, ok ? st.r[4 * j + 2] : 0u, ok ? st.r[4 * j + 3] : 0u};
      }
    }
  } else {
    const bool kmajor = s_k == 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int kk, ii;
      SgStage<T, VEC>::pos(threadIdx.x + 256 * e, kmajor, kk, ii);
      const bool ok = st.k0 + kk < K && i0 + ii < I;
      unsigned short v;
      if constexpr (sizeof(T) == 4) v = (unsigned short)(pack_bf16x2(__uint_as_float(st.r[e]), 0.f) & 0xffffu);
      else v = (unsigned short)st.r[e];
      S[kk][ii] = ok ? v : (unsigned short)0;
    }
  }
}

typedef short sg_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) sg_v4s* sg_lds_v4s_ptr;

// The same tile on the bf16 matrix pipe (bf16 compute mode): operands rounded to bf16 on their way into LDS (as every other weight
// / input gradient of that mode sees them), v_mfma_f32_16x16x32_bf16 with fragments gathered by ds_read_b64_tr_b16 from the k-major
// chunk, fp32 accumulation.  The exact-fp32 pipe (256 FLOP per clock and CU) bounded the fp32 tile at the large batches: the encoder
// weight gradients are 53 GFLOP at B = 1024 = 340 us at its peak, 483 us measured.
template <typename TA, typename TB>
__device__ __forceinline__ void small_gemm_tile_bf16(const TA* __restrict__ A, int64_t a_sm, int64_t a_sk,
                                                     const TB* __restrict__ B, int64_t b_sk, int64_t b_sn,
                                                     float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                     int accumulate) {
  constexpr int KC = SG_KC;
  __shared__ __attribute__((aligned(16))) unsigned short As[KC][SG_P16], Bs[KC][SG_P16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  constexpr int VA = 16 / sizeof(TA), VB = 16 / sizeof(TB);
  const bool a_vec = a_sm == 1 && ((uintptr_t)A % 16) == 0 && (a_sk * (int64_t)sizeof(TA)) % 16 == 0 && M % VA == 0;
  const bool b_vec = b_sn == 1 && ((uintptr_t)B % 16) == 0 && (b_sk * (int64_t)sizeof(TB)) % 16 == 0 && N % VB == 0;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // transposing read: the 16-lane group gq reads rows (k) 8 gq + tq (then + 4), columns c0 + 4 tp .. + 3; lane i of the group
  // receives column c0 + i of the four rows: its 8 consecutive k of one m / n (the 16x16x32 operand)
  const int gq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  auto frag = [&](const unsigned short (*S)[SG_P16], int c0) {
    const char* base = reinterpret_cast<const char*>(&S[8 * gq + tq][c0 + 4 * tp]);
    const sg_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sg_lds_v4s_ptr)base);
    const sg_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sg_lds_v4s_ptr)(base + 4 * SG_P16 * 2));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto mma = [&]() {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = frag(As, wm + 16 * i);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = frag(Bs, wn + 16 * j);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  const int nchunk = (K + KC - 1) / KC, last = nchunk - 1;
  auto run = [&](auto av, auto bv) {
    SgStage<TA, decltype(av)::value> a0, a1;
    SgStage<TB, decltype(bv)::value> b0, b1;
    a0.load(A, a_sm, a_sk, m0, M, 0, K);
    b0.load(B, b_sn, b_sk, n0, N, 0, K);
    a1.load(A, a_sm, a_sk, m0, M, min(1, last) * KC, K);
    b1.load(B, b_sn, b_sk, n0, N, min(1, last) * KC, K);
    for (int c = 0; c + 1 < nchunk; c += 2) {
      sg_store_bf16(a0, As, a_sk, m0, M, K);
      sg_store_bf16(b0, Bs, b_sk, n0, N, K);
      __syncthreads();
      a0.load(A, a_sm, a_sk, m0, M, min(c + 2, last) * KC, K);
      b0.load(B, b_sn, b_sk, n0, N, min(c + 2, last) * KC, K);
      mma();
      __syncthreads();
      sg_store_bf16(a1, As, a_sk, m0, M, K);
      sg_store_bf16(b1, Bs, b_sk, n0, N, K);
      __syncthreads();
      a1.load(A, a_sm, a_sk, m0, M, min(c + 3, last) * KC, K);
      b1.load(B, b_sn, b_sk, n0, N, min(c + 3, last) * KC, K);
      mma();
      __syncthreads();
    }
    if (nchunk & 1) {
      sg_store_bf16(a0, As, a_sk, m0, M, K);
      sg_store_bf16(b0, Bs, b_sk, n0, N, K);
      __syncthreads();
      mma();
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  if (a_vec && b_vec) run(Yes{}, Yes{});
  else if (a_vec) run(Yes{}, No{});
  else if (b_vec) run(No{}, Yes{});
  else run(No{}, No{});
  // C/D map of the 16x16 MFMA: col (n) = lane & 15, row (m) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + 16 * j + (lane & 15);
      if (n >= N) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm + 16 * i + 4 * (lane >> 4) + e;
        if (m >= M) continue;
        float* c = C + (int64_t)m * ldc + n;
        *c = accumulate ? *c + acc[i][j][e] : acc[i][j][e];
      }
    }
}

template <typename TA, typename TB>
__device__ __forceinline__ void small_gemm_tile(const TA* __restrict__ A, int64_t a_sm, int64_t a_sk,
                                                const TB* __restrict__ B, int64_t b_sk, int64_t b_sn,
                                                float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                int accumulate) {
  constexpr int KC = SG_KC;
  __shared__ __attribute__((aligned(16))) float As[KC][SG_PITCH], Bs[KC][SG_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  constexpr int VA = 16 / sizeof(TA), VB = 16 / sizeof(TB);
  const bool a_vec = a_sm == 1 && ((uintptr_t)A % 16) == 0 && (a_sk * (int64_t)sizeof(TA)) % 16 == 0 && M % VA == 0;
  const bool b_vec = b_sn == 1 && ((uintptr_t)B % 16) == 0 && (b_sk * (int64_t)sizeof(TB)) % 16 == 0 && N % VB == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto mma = [&]() {
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
      const float a = As[kk + (lane >> 5)][wm + (lane & 31)];
      const float b = Bs[kk + (lane >> 5)][wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  };
  const int nchunk = (K + KC - 1) / KC, last = nchunk - 1;
  // the whole K loop once per pair of staging modes (workgroup-uniform): no branch on them inside it
  auto run = [&](auto av, auto bv) {
    SgStage<TA, decltype(av)::value> a0, a1;
    SgStage<TB, decltype(bv)::value> b0, b1;
    a0.load(A, a_sm, a_sk, m0, M, 0, K);
    b0.load(B, b_sn, b_sk, n0, N, 0, K);
    a1.load(A, a_sm, a_sk, m0, M, min(1, last) * KC, K);       // past the last chunk the last one is loaded again: no branch
    b1.load(B, b_sn, b_sk, n0, N, min(1, last) * KC, K);
    for (int c = 0; c + 1 < nchunk; c += 2) {
      a0.store(As, a_sk, m0, M, K);
      b0.store(Bs, b_sk, n0, N, K);
      __syncthreads();
      a0.load(A, a_sm, a_sk, m0, M, min(c + 2, last) * KC, K);
      b0.load(B, b_sn, b_sk, n0, N, min(c + 2, last) * KC, K);
      mma();
      __syncthreads();
      a1.store(As, a_sk, m0, M, K);
      b1.store(Bs, b_sk, n0, N, K);
      __syncthreads();
      a1.load(A, a_sm, a_sk, m0, M, min(c + 3, last) * KC, K);
      b1.load(B, b_sn, b_sk, n0, N, min(c + 3, last) * KC, K);
      mma();
      __syncthreads();
    }
    if (nchunk & 1) {                                           // stage 0 holds the last chunk
      a0.store(As, a_sk, m0, M, K);
      b0.store(Bs, b_sk, n0, N, K);
      __syncthreads();
      mma();
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  if (a_vec && b_vec) run(Yes{}, Yes{});
  else if (a_vec) run(Yes{}, No{});
  else if (b_vec) run(No{}, Yes{});
  else run(No{}, No{});
  const int n = n0 + wn + (lane & 31);
  if (n < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= M) continue;
      float* c = C + (int64_t)m * ldc + n;
      *c = accumulate ? *c + acc[r] : acc[r];
    }
  }
}

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void small_gemm_kernel(const TA* __restrict__ A, int64_t a_sm, int64_t a_sk,
                                                         const TB* __restrict__ B, int64_t b_sk, int64_t b_sn,
                                                         float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                         int accumulate) {
  small_gemm_tile<TA, TB>(A, a_sm, a_sk, B, b_sk, b_sn, C, ldc, M, N, K, accumulate);
}

// The same tile over a ragged batch of problems (blockIdx.z = problem): one launch for all tokens of the encoder /
// decoder instead of one per token.
constexpr int SG_MAX_PROBLEMS = 40;
struct SmallGemmBatch {
  const void* A[SG_MAX_PROBLEMS];
  const void* B[SG_MAX_PROBLEMS];
  float* C[SG_MAX_PROBLEMS];
  int M[SG_MAX_PROBLEMS], N[SG_MAX_PROBLEMS], K[SG_MAX_PROBLEMS], ldc[SG_MAX_PROBLEMS];
  int64_t a_sm, a_sk, b_sk, b_sn;
  int accumulate;
};

template <typename TA, typename TB, bool BF16_MMA>
__global__ __launch_bounds__(256) void small_gemm_batched_kernel(SmallGemmBatch g) {
  const int z = blockIdx.z;
  const int M = g.M[z], N = g.N[z];
  if ((int)blockIdx.y * 64 >= M || (int)blockIdx.x * 64 >= N) return;      // uniform per workgroup
  if constexpr (BF16_MMA)
    small_gemm_tile_bf16<TA, TB>((const TA*)g.A[z], g.a_sm, g.a_sk, (const TB*)g.B[z], g.b_sk, g.b_sn, g.C[z], g.ldc[z], M, N, g.K[z],
                                 g.accumulate);
  else
    small_gemm_tile<TA, TB>((const TA*)g.A[z], g.a_sm, g.a_sk, (const TB*)g.B[z], g.b_sk, g.b_sn, g.C[z], g.ldc[z], M, N, g.K[z],
                            g.accumulate);
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
// written to dst[col][dd] (row-major [T][ld_out] slice of one head).
// bf16: as in attention.hip, the src tile [T x HD] is staged into the wave's LDS tile `lt` by coalesced 16-byte loads (rows >= T
// zero) and its transposed fragments (8 rows of one head-dim per lane) come from ds_read_b64_tr_b16 - the scalar form took 48
// two-byte global loads per lane and call; the result goes back through the same tile and leaves as 16-byte stores that cover
// whole rows of the head.  Only the wave itself touches its tile.
typedef short bwd_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bwd_v4s* bwd_lds_v4s_ptr;
template <int HD>
constexpr int bwd_v_pitch = HD == 128 ? 320 : 192;   // conflict-free pitches of the transposing read (attention.hip)
template <int HD>
constexpr int bwd_tile_bytes = 32 * ((HD * 2 + 16) > bwd_v_pitch<HD> ? (HD * 2 + 16) : bwd_v_pitch<HD>);

template <typename T, int HD>
__device__ __forceinline__ void cols_product_store(const T* __restrict__ src, int64_t ld, const f32x16& X, float scale,
                                                   T* __restrict__ dst, int64_t ld_out, int r, int hh, int Ttok, char* lt) {
  constexpr int ND = HD / 32;
  if constexpr (sizeof(T) == 2) {
    constexpr int VP = bwd_v_pitch<HD>, OP = HD * 2 + 16, CPR = HD / 8, NVL = (32 * CPR + 63) / 64;
    const int lane = r + 32 * hh;
    bf16x8 xp[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const uint32_t pk = pack_bf16x2(X[8 * s + j], X[8 * s + j + 1]);
        xp[s][j] = (short)(pk & 0xffffu);
        xp[s][j + 1] = (short)(pk >> 16);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous use of the tile (its output reads) is over
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c = lane + 64 * i, row = c / CPR, col = c - row * CPR;
      u32x4 v = u32x4{0, 0, 0, 0};
      if (row < Ttok) v = *reinterpret_cast<const u32x4*>(src + (int64_t)row * ld + col * 8);
      if (c < 32 * CPR) *reinterpret_cast<u32x4*>(lt + row * VP + col * 16) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own writes only: the LDS serves one wave's instructions in order
    __builtin_amdgcn_wave_barrier();
    const int gq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const char* tbase = lt + (4 * (gq >> 1) + tq) * VP + (16 * (gq & 1) + 4 * tp) * 2;
    f32x16 O[ND];
#pragma unroll
    for (int t = 0; t < ND; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) O[t][e] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bwd_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bwd_lds_v4s_ptr)(tbase + (16 * s) * VP + t * 64));
        const bwd_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bwd_lds_v4s_ptr)(tbase + (16 * s + 8) * VP + t * 64));
        const bf16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, xp[s], O[t], 0, 0, 0);
      }
    }
    // output tile through LDS: pitch HD * 2 + 16 (see attention.hip), then 16-byte row stores
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (r < Ttok) {
#pragma unroll
      for (int t = 0; t < ND; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 o;
          o.x = pack_bf16x2(O[t][4 * g] * scale, O[t][4 * g + 1] * scale);
          o.y = pack_bf16x2(O[t][4 * g + 2] * scale, O[t][4 * g + 3] * scale);
          *reinterpret_cast<uint2*>(lt + r * OP + (t * 32 + 8 * g + 4 * hh) * 2) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int nchunk = Ttok * CPR;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c = lane + 64 * i, row = c / CPR, col = c - row * CPR;
      if (c < nchunk) *reinterpret_cast<u32x4*>(dst + (int64_t)row * ld_out + col * 8) = *reinterpret_cast<const u32x4*>(lt + row * OP + col * 16);
    }
  } else {
#pragma unroll
    for (int t = 0; t < ND; ++t) {
      f32x16 O;
#pragma unroll
      for (int e = 0; e < 16; ++e) O[e] = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row_of(e, hh);
        const float v = row < Ttok ? src[(int64_t)row * ld + t * 32 + r] : 0.f;
        O = __builtin_amdgcn_mfma_f32_32x32x2f32(v, X[e], O, 0, 0, 0);
      }
      if (r < Ttok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          T* p = dst + (int64_t)r * ld_out + t * 32 + 8 * g + 4 * hh;
          *reinterpret_cast<float4*>(p) = make_float4(O[4 * g] * scale, O[4 * g + 1] * scale, O[4 * g + 2] * scale, O[4 * g + 3] * scale);
        }
      }
    }
  }
}

// Two workgroups per CU asked of the register allocator for head dims <= 96: 182 registers and no AGPR copies instead of 168 + 48
// (same two waves per SIMD): 195 -> 182 us at B = 1024, 29.7 -> 28.4 at B = 128; head dim 128 measures 4 % slower that way and
// three workgroups per CU (8 spilled registers) slower everywhere (tools/attn_bench.py, A/B x2 on one box).
template <typename T, int HD>
__global__ __launch_bounds__(256, HD <= 96 ? 2 : 1) void attention_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dy,
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
  __shared__ __attribute__((aligned(16))) char lds_tiles[sizeof(T) == 2 ? 4 * bwd_tile_bytes<HD> : 16];
  char* lt = lds_tiles + (sizeof(T) == 2 ? (threadIdx.x >> 6) * bwd_tile_bytes<HD> : 0);

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
  cols_product_store<T, HD>(kb, ld, dSt, scale, dqb, ld, r, hh, Ttok, lt);

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
  cols_product_store<T, HD>(qb, ld, dS, scale, dkb, ld, r, hh, Ttok, lt);
  cols_product_store<T, HD>(dob, d, P, 1.0f, dvb, ld, r, hh, Ttok, lt);
}

}  // namespace

extern "C" int gpt_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* resid, float* gout, void* gout_t, int gout_t_dtype,
                                 float* dgamma, float* dbeta, int rows, int d, void* stream) {
  return gpt_layernorm_bwd_colsum(dy, dy_dtype, x, mean, rstd, gamma, resid, gout, gout_t, gout_t_dtype, dgamma, dbeta, nullptr, rows,
                                  d, stream);
}

extern "C" int gpt_layernorm_bwd_colsum(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                                        const float* gamma, const float* resid, float* gout, void* gout_t, int gout_t_dtype,
                                        float* dgamma, float* dbeta, float* gcol, int rows, int d, void* stream) {
  return gpt_layernorm_bwd_part(dy, dy_dtype, x, mean, rstd, gamma, resid, gout, gout_t, gout_t_dtype, dgamma, dbeta, gcol, nullptr,
                                nullptr, rows, d, stream);
}

// Sums of the partials buffers of up to GPT_LN_PART_MAX_CALLS layernorm backward launches: out[c] += sum_b part[b][k][c], blocks in
// order (deterministic).  One launch for all of them: grid (column blocks, call x 3).
namespace {
__global__ __launch_bounds__(256) void ln_partials_reduce_kernel(LnPartCalls calls, int d) {
  const int call = blockIdx.y / 3, k = blockIdx.y - 3 * call;
  float* out = calls.c[call].out[k];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (!out || c >= d) return;
  const float* p = calls.c[call].part + (int64_t)k * d + c;
  const int nb = calls.c[call].nblocks;
  const int64_t stride = 3 * (int64_t)d;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int b = 0;
  for (; b + 4 <= nb; b += 4) {          // four loads in flight per thread
    t0 += p[(b + 0) * stride];
    t1 += p[(b + 1) * stride];
    t2 += p[(b + 2) * stride];
    t3 += p[(b + 3) * stride];
  }
  for (; b < nb; ++b) t0 += p[b * stride];
  out[c] += (t0 + t1) + (t2 + t3);
}
}  // namespace

int gpt_ln_partials_reduce(const LnPartCalls& calls, int d, void* stream) {
  if (calls.n == 0) return 0;
  hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3((d + 255) / 256, calls.n * 3), dim3(256), 0, (hipStream_t)stream, calls, d);
  return gpt_launch_status("gpt_layernorm_bwd: partials reduce");
}

int gpt_layernorm_bwd_part(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* resid, float* gout, void* gout_t, int gout_t_dtype, float* dgamma, float* dbeta, float* gcol,
                           float* part, int* n_blocks, int rows, int d, void* stream) {
  GPT_REQUIRE(dy && x && mean && rstd && gamma && gout && dgamma && dbeta, "gpt_layernorm_bwd: null operand");
  GPT_REQUIRE(rows > 0 && d > 0 && d % 4 == 0 && d <= 3072, "gpt_layernorm_bwd: bad sizes (need d %% 4 == 0, d <= 3072)");
  // every workgroup ends with one atomic per column on the SAME 2d addresses: their number, not the streaming, sets the tail
  const int waves = d > 2048 ? lnb_waves<12> : lnb_waves<8>;  // as the instance chosen below
  int blocks = (rows + waves - 1) / waves;
  blocks = blocks > 256 ? 256 : blocks;
  const int rpb = (rows + blocks - 1) / blocks;
  const dim3 grid((rows + rpb - 1) / rpb), block(waves * 64);
  if (n_blocks) *n_blocks = (int)grid.x;                      // rows of the partials buffer this launch fills (<= 256)
  const size_t lds = (size_t)waves * d * sizeof(float);       // <= 64 KiB
  hipStream_t s = (hipStream_t)stream;
  const bool dyb = dy_dtype == GPT_BF16, ob = gout_t_dtype == GPT_BF16;
#define LNB(DY, OUT, NCH)                                                                                              \
  do {                                                                                                                  \
    static GptPerDevice attr; /* more than 64 KiB of dynamic LDS (d > 2048) must be granted once per kernel and device */ \
    bool& attr_set = attr.here();                                                                                       \
    if (!attr_set) {                                                                                                    \
      if (hipFuncSetAttribute((const void*)layernorm_bwd_kernel<DY, OUT, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              lnb_waves<NCH> * NCH * 256 * (int)sizeof(float)) != hipSuccess) {                              \
        gpt_set_error("gpt_layernorm_bwd: cannot reserve LDS");                                                         \
        return -1;                                                                                                      \
      }                                                                                                                 \
      attr_set = true;                                                                                                  \
    }                                                                                                                   \
    hipLaunchKernelGGL((layernorm_bwd_kernel<DY, OUT, NCH>), grid, block, lds, s, (const DY*)dy, x, mean, rstd, gamma, resid, \
                       gout, (OUT*)gout_t, dgamma, dbeta, gcol, part, rows, d, rpb);                                              \
  } while (0)
#define LNB_T(NCH)                         \
  do {                                     \
    if (dyb && ob) LNB(bf16_t, bf16_t, NCH); \
    else if (dyb) LNB(bf16_t, float, NCH); \
    else if (ob) LNB(float, bf16_t, NCH);  \
    else LNB(float, float, NCH);           \
  } while (0)
  if (d <= 256) LNB_T(1);
  else if (d <= 512) LNB_T(2);
  else if (d <= 1024) LNB_T(4);
  else if (d <= 1536) LNB_T(6);
  else if (d <= 2048) LNB_T(8);
  else LNB_T(12);
#undef LNB_T
#undef LNB
  return gpt_launch_status("gpt_layernorm_bwd");
}

extern "C" int gpt_colsum(int dtype, const void* X, int64_t ld, float* out, int rows, int cols, void* stream) {
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_colsum: bad dtype %d", dtype);
  GPT_REQUIRE(X && out && rows > 0 && cols > 0 && ld >= cols, "gpt_colsum: bad arguments");
  // enough workgroups to fill the chip (2-3 per CU), at least 64 rows each so the LDS meet / atomics stay amortised
  const int col_blocks = (cols + 255) / 256;
  int row_blocks = (768 + col_blocks - 1) / col_blocks;
  if (row_blocks > (rows + 63) / 64) row_blocks = (rows + 63) / 64;
  const int rpb = (rows + row_blocks - 1) / row_blocks;
  const dim3 grid((cols + 255) / 256, (rows + rpb - 1) / rpb), block(256);
  const int esz = dtype == GPT_BF16 ? 2 : 4;
  const int vec = ((uintptr_t)X % 16 == 0) && ((ld * esz) % 16 == 0);
  if (dtype == GPT_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)X, ld, out, rows, cols, rpb, vec);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)X, ld, out, rows, cols, rpb, vec);
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

extern "C" int gpt_small_gemm_batched(int n_problems, int a_dtype, const void* const* host_A, int64_t a_sm, int64_t a_sk,
                                      int b_dtype, const void* const* host_B, int64_t b_sk, int64_t b_sn, float* const* host_C,
                                      const int* host_ldc, const int* host_M, const int* host_N, const int* host_K,
                                      int accumulate, void* stream) {
  return gpt_small_gemm_batched_mode(n_problems, a_dtype, host_A, a_sm, a_sk, b_dtype, host_B, b_sk, b_sn, host_C, host_ldc, host_M,
                                     host_N, host_K, accumulate, 0, stream);
}

int gpt_small_gemm_batched_mode(int n_problems, int a_dtype, const void* const* host_A, int64_t a_sm, int64_t a_sk, int b_dtype,
                                const void* const* host_B, int64_t b_sk, int64_t b_sn, float* const* host_C, const int* host_ldc,
                                const int* host_M, const int* host_N, const int* host_K, int accumulate, int bf16_mma,
                                void* stream) {
  GPT_REQUIRE(n_problems > 0 && n_problems <= SG_MAX_PROBLEMS, "gpt_small_gemm_batched: 1..%d problems", SG_MAX_PROBLEMS);
  GPT_REQUIRE(host_A && host_B && host_C && host_ldc && host_M && host_N && host_K, "gpt_small_gemm_batched: null array");
  SmallGemmBatch g;
  int max_m = 0, max_n = 0;
  for (int i = 0; i < n_problems; ++i) {
    GPT_REQUIRE(host_A[i] && host_B[i] && host_C[i] && host_M[i] > 0 && host_N[i] > 0 && host_K[i] > 0 && host_ldc[i] >= host_N[i],
                "gpt_small_gemm_batched: problem %d: bad arguments", i);
    g.A[i] = host_A[i]; g.B[i] = host_B[i]; g.C[i] = host_C[i];
    g.M[i] = host_M[i]; g.N[i] = host_N[i]; g.K[i] = host_K[i]; g.ldc[i] = host_ldc[i];
    max_m = host_M[i] > max_m ? host_M[i] : max_m;
    max_n = host_N[i] > max_n ? host_N[i] : max_n;
  }
  g.a_sm = a_sm; g.a_sk = a_sk; g.b_sk = b_sk; g.b_sn = b_sn; g.accumulate = accumulate;
  const dim3 grid((max_n + 63) / 64, (max_m + 63) / 64, n_problems), block(256);
  hipStream_t s = (hipStream_t)stream;
  static const bool allow16 = [] { const char* e = getenv("GPT_SMALL_GEMM_BF16"); return !e || atoi(e) != 0; }();   // 0: always the exact pipe (A/B)
#define SGB(TA, TB)                                                                                        \
  do {                                                                                                     \
    if (bf16_mma && allow16) hipLaunchKernelGGL((small_gemm_batched_kernel<TA, TB, true>), grid, block, 0, s, g); \
    else hipLaunchKernelGGL((small_gemm_batched_kernel<TA, TB, false>), grid, block, 0, s, g);            \
  } while (0)
  const bool ab = a_dtype == GPT_BF16, bb = b_dtype == GPT_BF16;
  if (ab && bb) SGB(bf16_t, bf16_t);
  else if (ab) SGB(bf16_t, float);
  else if (bb) SGB(float, bf16_t);
  else SGB(float, float);
#undef SGB
  return gpt_launch_status("gpt_small_gemm_batched");
}
