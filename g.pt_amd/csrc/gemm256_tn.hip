// 256 x 256 block-tile bf16 MFMA GEMM with BOTH operands contracted over their ROW index ("TN"):
//     C[I, J] (epilogue)= sum_r P[r, i] * Q[r, j]          P: [R, I] row-major, Q: [R, J] row-major
// This is the weight gradient dW[n_out, k_in] = dY^T . X of every Linear of the trunk (autograd of
// Gpt/models/transformer.py:38-40,50,63-65): P = dY [B*T, n_out], Q = X [B*T, k_in] are consumed exactly as the
// forward / backward kernels wrote them - no transposed copies of the activations.
//
// Same pipeline as gemm256.hip (8 waves as 2 x 4 of 128 x 64, two 64-KiB K-tile buffers of four 16-KiB half-tiles,
// two phases per K-tile, LDS-DMA refills with counted vmcnt, two wave groups one barrier apart).  What differs:
//  * a K-tile is 64 ROWS r of the operands; a half-tile holds, for each of them, 128 CONTIGUOUS columns of the block
//    tile (full 256-byte runs for the LDS-DMA) = 64 LDS rows of 256 bytes; a wave therefore owns output rows
//    {128 hi + 64 wr + [0, 64)} and columns {128 hj + 32 wc + [0, 32)} for the halves hi, hj = 0, 1;
//  * the MFMA operands (16x16x32: 8 consecutive r for one column) are gathered with ds_read_b64_tr_b16, gfx950's
//    transposing LDS read: a 16-lane group reads a 4-row x 16-column block and every lane receives one COLUMN of it, so
//    two reads give a lane its 8 k-values.  16-byte chunk c of LDS row r is stored at chunk
//    c ^ ((r & 3) << 2) ^ (((r >> 3) & 1) << 1): the four rows of a block sit in four different 64-byte bank groups and
//    the two groups of a 32-lane half (rows 8 apart, same columns) in the two 32-byte halves of them: conflict-free.
// Needs I % 256 == 0, J % 256 == 0 (the host falls back to transposes + the NT kernel otherwise).  Any R: the rows of the last
// K-tile beyond R come from a 16-byte block of zeros for P and from row R - 1 (finite values times zero) for Q.
#include "gemm_common.h"
#include <stdlib.h>

namespace gemmk {
namespace {

constexpr int TRB = 256;             // bytes per LDS row (128 bf16 columns)
constexpr int THALF = 64 * TRB;      // 16 KiB half-tile: 64 r-rows
constexpr int TBUF = 4 * THALF;      // P-h0 | P-h1 | Q-h0 | Q-h1
constexpr int TLDS_BYTES = 2 * TBUF;  // 128 KiB

// what the LDS-DMA reads for the P rows past the end of a ragged contraction (every lane the same 16 bytes)
__device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s_ptr;

// the 8 k-values (LDS rows 32 kk + 8 (lane >> 4) + [0, 8)) of this lane's column: two transposing reads 4 rows apart
__device__ __forceinline__ u32x4 tr_frag(const char* base, int kk) {
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(base + kk * 32 * TRB));
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(base + kk * 32 * TRB + 4 * TRB));
  typedef short v8s __attribute__((ext_vector_type(8)));
  const v8s both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);   // register concatenation, no VALU
  return __builtin_bit_cast(u32x4, both);
}

// ksplit > 1 (accumulating epilogue only): the R range is cut into ksplit contiguous spans, one workgroup per (span, tile),
// and every workgroup ADDS its partial tile with hardware fp32 atomics (one accumulator register = two full 128-byte
// lines per wave instruction).  A weight gradient has few output tiles (d x d: 36, d x 4d: 144 on 256 CUs) and a very
// long contraction (R = B*T rows), so spans fill the chip for the price of the atomics (~0.2 us per partial tile).
// GROUPED launches: up to 16 independent problems (the four weight gradients of up to four Blocks) with the same contraction length share one grid (workgroups [off[p], off[p+1])
// belong to problem p).  The four weight gradients of a Block (qkv, proj, mlp.0, mlp.2) have 108 + 36 + 144 + 144 = 432 tiles at
// d = 1536: launched one by one each leaves CUs idle (144 tiles on 256 CUs) or pays atomics for spans; together they are two
// nearly full rounds of whole contractions, no atomics (B = 128: 4 x 85 us -> see DESIGN.md).
constexpr int TN_MAX_GROUP = 16;
struct TnGroup {
  GemmArgs<bf16_t> g[TN_MAX_GROUP];
  int off[TN_MAX_GROUP + 1];
  int n, ksplit;
  float* sumsq;   // store mode: += sum of squares of every stored tile (one float atomic per wave), or NULL
};

// RAGGED: some problem of the group has a contraction length that is no multiple of 64 (the branch below, although taken by one
// K-tile per workgroup at most, costs the steady loop 2.5 - 4 % when compiled in: whole-tile launches get the instance without it)
template <int EPI, bool RAGGED>
__global__ __launch_bounds__(512, 2) void gemm256_tn_kernel(TnGroup grp) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // this workgroup's problem (workgroup-uniform selects: no dynamic indexing of the kernel argument)
  GemmArgs<bf16_t> g = grp.g[0];
  int wg0 = 0;
#pragma unroll
  for (int p = 1; p < TN_MAX_GROUP; ++p)
    if (p < grp.n && (int)blockIdx.x >= grp.off[p]) { g = grp.g[p]; wg0 = grp.off[p]; }
  const int ksplit = grp.ksplit;
  const int local = (int)blockIdx.x - wg0;
  const int nwg = (g.M / 256) * (g.N / 256), span = local / nwg, bid = local - span * nwg;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  // column groups of 8 tiles, rows fastest across the group (see gemm256.hip): both strips re-used from the XCD's L2
  constexpr int GROUP_N = 8;
  const int tiles_n = g.N / 256, tiles_m = g.M / 256;
  const int gb = tile / (tiles_m * GROUP_N), grem = tile - gb * (tiles_m * GROUP_N);
  const int gw = min(GROUP_N, tiles_n - gb * GROUP_N);
  const int m0 = (grem / gw) * 256, n0 = (gb * GROUP_N + grem % gw) * 256;   // output rows (P columns) / columns (Q columns)
  const int nk_all = (g.K + 63) / 64, nk_span = (nk_all + ksplit - 1) / ksplit, kt0 = span * nk_span;
  const int nk = min(nk_span, nk_all - kt0);
  if (nk <= 0) return;                                      // workgroup-uniform

  // per-lane DMA sources: this wave fills LDS rows [wid*8, wid*8+8) of every half-tile (2 pieces of 4 rows);
  // lane -> (row = lane >> 4, chunk = lane & 15) of the piece, the chunk's logical position un-swizzled
  const T* srcP[2][2];  // [half][piece]
  const T* srcQ[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int lr = wid * 8 + p * 4 + (lane >> 4);
    const int lc = (lane & 15) ^ ((lr & 3) << 2) ^ (((lr >> 3) & 1) << 1);   // logical 16-byte chunk (8 columns) of the 128-column row
    const int e = lc * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      srcP[h][p] = g.A + ((int64_t)kt0 * 64 + lr) * g.lda + m0 + h * 128 + e;
      srcQ[h][p] = g.W + ((int64_t)kt0 * 64 + lr) * g.ldw + n0 + h * 128 + e;
    }
  }
  const int64_t stepP = 64 * g.lda, stepQ = 64 * g.ldw;
  // A ragged contraction (R % 64 != 0) ends in a K-tile with rows past R: P reads zeros for them, Q its last valid row (finite
  // values times zero).  Only that one K-tile of the last span takes the branch (workgroup-uniform), everything else the two plain
  // instructions: the selects cost the steady loop nothing.
  const int full_tiles = g.K / 64 - kt0;                     // K-tiles of this span, counted from its first, that are whole
  auto dma = [&](const T* const (&src)[2], int64_t step, int kt, int slot, bool is_q) {
    char* dst = smem + (kt & 1) * TBUF + slot * THALF + wid * 2048;
    const T* s0 = src[0] + kt * step;
    const T* s1 = src[1] + kt * step;
    if (RAGGED && kt >= full_tiles) {
      const int over0 = (kt0 + kt) * 64 + wid * 8 + (lane >> 4) - (g.K - 1), over1 = over0 + 4;   // rows past the last valid one
      if (is_q) {
        if (over0 > 0) s0 -= (int64_t)over0 * g.ldw;
        if (over1 > 0) s1 -= (int64_t)over1 * g.ldw;
      } else {
        if (over0 > 0) s0 = reinterpret_cast<const T*>(g_zero16);
        if (over1 > 0) s1 = reinterpret_cast<const T*>(g_zero16);
      }
    }
    __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)s1, (lptr_t)(dst + 1024), 16, 0, 0);
  };

  // 16x16x32 MFMAs (higher sustained clock than 32x32x16, see gemm256.hip): 8 x 4 accumulator tiles of 16 x 16 per wave
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < nk) {
      dma(srcP[0], stepP, t, 0, false);
      dma(srcQ[0], stepQ, t, 2, true);
      dma(srcQ[1], stepQ, t, 3, true);
      dma(srcP[1], stepP, t, 1, false);
    }

  // transposed-read addresses: 16-lane group gq reads the block rows 8 gq + q (then + 4), columns e0 + 4 p .. + 3
  const int gq = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  auto tr_off = [&](int e0) {
    const int e = e0 + 4 * tp;
    return (8 * gq + tq) * TRB + (((e >> 3) ^ (tq << 2) ^ ((gq & 1) << 1)) << 4) + 8 * (tp & 1);
  };
  int offP[4], offQ[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) offP[i] = tr_off(wr * 64 + i * 16);
#pragma unroll
  for (int j = 0; j < 2; ++j) offQ[j] = tr_off(wc * 32 + j * 16);

  // a[i][kk]: row tile i of the P half, b[j][kk]: column tile j of the Q half, kk = 32-deep K step of the 64-row K-tile
  auto mma = [&](const u32x4 (&a)[4][2], const u32x4 (&b)[2][2], int i0, int j0) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][kk]),
                                                                       __builtin_bit_cast(bf16x8, b[j][kk]), acc[i0 + i][j0 + j], 0, 0, 0);
  };
  auto read_p = [&](u32x4 (&a)[4][2], const char* half) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) a[i][kk] = tr_frag(half + offP[i], kk);
  };
  auto read_q = [&](u32x4 (&b)[2][2], const char* half) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) b[j][kk] = tr_frag(half + offQ[j], kk);
  };
#define SEG_BARRIER()                        \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)
#define SEG_READS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

  // two phases of 32 MFMAs per K-tile, waits and refills exactly as in gemm256.hip:
  //   P0: reads P-h0, Q-h0, Q-h1 -> (P-h0 x Q-h0), (P-h0 x Q-h1); refills P-h1(kt+1)
  //   P1: reads P-h1             -> (P-h1 x Q-h1), (P-h1 x Q-h0); refills P-h0, Q-h0, Q-h1 (kt+2)
  if (nk >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if (wr == 1) __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nk; ++kt) {
    const char* buf = smem + (kt & 1) * TBUF;
    const bool steady = kt + 2 < nk, last = kt + 1 == nk;
    u32x4 a[4][2], b0[2][2], b1[2][2];
    // ---- P0
    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    SEG_BARRIER();
    read_q(b0, buf + 2 * THALF);
    read_p(a, buf);
    read_q(b1, buf + 3 * THALF);
    if (kt >= 1 && !last) dma(srcP[1], stepP, kt + 1, 1, false);
    SEG_READS_DONE();
    SEG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    mma(a, b0, 0, 0);
    mma(a, b1, 0, 2);
    __builtin_amdgcn_s_setprio(0);
    // ---- P1
    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    SEG_BARRIER();
    read_p(a, buf + THALF);
    if (steady) {
      dma(srcP[0], stepP, kt + 2, 0, false);
      dma(srcQ[0], stepQ, kt + 2, 2, true);
      dma(srcQ[1], stepQ, kt + 2, 3, true);
    }
    SEG_READS_DONE();
    SEG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    mma(a, b1, 4, 2);
    mma(a, b0, 4, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef SEG_BARRIER
#undef SEG_READS_DONE
  __syncthreads();

  // epilogue straight from the accumulator layout (16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg): the output is
  // fp32, every wave instruction reads / writes four 64-byte row segments
  const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* c = (float*)g.C + (int64_t)(m0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + 4 * l4) * g.ldc + n0 + (j >> 1) * 128 + wc * 32 +
                 (j & 1) * 16 + l15;
      if (EPI == GPT_EPI_ACCUM_F32 && ksplit > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(c + (int64_t)e * g.ldc, acc[i][j][e]);
      } else if constexpr (EPI == GPT_EPI_ACCUM_F32) {
        float old[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) old[e] = c[(int64_t)e * g.ldc];
#pragma unroll
        for (int e = 0; e < 4; ++e) c[(int64_t)e * g.ldc] = old[e] + acc[i][j][e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) c[(int64_t)e * g.ldc] = acc[i][j][e];
      }
    }
  if constexpr (EPI == GPT_EPI_STORE_F32) {
    if (grp.sumsq) {                                          // workgroup-uniform
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) ss = fmaf(acc[i][j][e], acc[i][j][e], ss);
      ss = wave_sum(ss);
      if (lane == 0) unsafeAtomicAdd(grp.sumsq, ss);
    }
  }
}

// spans: estimated time (us) = rounds x (K-tiles per span x 1.5 + 10) + atomics at ~0.2 us per partial tile and CU round
int pick_tn_spans(int tiles, int nk_all) {
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  static const int forced = [] {
    const char* e = getenv("GPT_TN_KSPLIT");   // experiments: force a span count (0 = model)
    return e ? atoi(e) : 0;
  }();
  if (forced > 0) return forced;
  int ksplit = 1;
  double best = 1e30;
  for (int sp = 1; sp <= 16 && sp * 8 <= nk_all; ++sp) {
    const int wgs = tiles * sp, rounds = (wgs + n_cu - 1) / n_cu, per = (nk_all + sp - 1) / sp;
    const double t = rounds * (per * 1.5 + 10.0) + (sp > 1 ? 0.2 * wgs : 0.0);
    if (t < best) { best = t; ksplit = sp; }
  }
  return ksplit;
}

template <int EPI, bool RAGGED>
int launch_tn(TnGroup& grp, hipStream_t s) {
  static GptPerDevice attr;   // the LDS grant is per kernel AND per device (a process may drive several GPUs)
  bool& attr_set = attr.here();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm256_tn_kernel<EPI, RAGGED>, hipFuncAttributeMaxDynamicSharedMemorySize, TLDS_BYTES) !=
        hipSuccess) {
      gpt_set_error("gpt_gemm_tn: cannot reserve %d bytes of LDS", TLDS_BYTES);
      return -1;
    }
    attr_set = true;
  }
  int tiles = 0;
  double flops = 0;
  for (int p = 0; p < grp.n; ++p) {
    tiles += (grp.g[p].M / 256) * (grp.g[p].N / 256);
    flops += 2.0 * grp.g[p].M * grp.g[p].N * grp.g[p].K;
  }
  grp.ksplit = EPI == GPT_EPI_ACCUM_F32 ? pick_tn_spans(tiles, (grp.g[0].K + 63) / 64) : 1;   // spans add with atomics: accumulating only
  grp.off[0] = 0;
  for (int p = 0; p < grp.n; ++p) grp.off[p + 1] = grp.off[p] + (grp.g[p].M / 256) * (grp.g[p].N / 256) * grp.ksplit;
  const bool rec = g_prof.on && g_prof.n < GemmProf::CAP;
  if (rec) (void)hipEventRecord(g_prof.ev[g_prof.n][0], s);
  hipLaunchKernelGGL((gemm256_tn_kernel<EPI, RAGGED>), dim3(grp.off[grp.n]), dim3(512), TLDS_BYTES, s, grp);
  if (rec) {
    (void)hipEventRecord(g_prof.ev[g_prof.n][1], s);
    g_prof.flops[g_prof.n++] = flops;
  }
  return gpt_launch_status("gpt_gemm_tn");
}

}  // namespace
}  // namespace gemmk

static int tn_check(const void* P, int64_t ldp, const void* Q, int64_t ldq, const float* C, int64_t ldc, int I, int J, int R) {
  GPT_REQUIRE(P && Q && C, "gpt_gemm_tn: null operand");
  GPT_REQUIRE(I > 0 && J > 0 && R > 0 && I % 256 == 0 && J % 256 == 0,
              "gpt_gemm_tn: need I %% 256 == 0, J %% 256 == 0, R > 0 (got I=%d J=%d R=%d)", I, J, R);
  GPT_REQUIRE(ldp >= I && ldq >= J && ldc >= J && ldp % 8 == 0 && ldq % 8 == 0 && ldc % 4 == 0, "gpt_gemm_tn: bad leading dimensions");
  GPT_REQUIRE(((uintptr_t)P % 16) == 0 && ((uintptr_t)Q % 16) == 0 && ((uintptr_t)C % 16) == 0, "gpt_gemm_tn: operands must be 16-byte aligned");
  return 0;
}

extern "C" int gpt_gemm_tn(const void* P, int64_t ldp, const void* Q, int64_t ldq, float* C, int64_t ldc, int I, int J, int R,
                           int accumulate, void* stream) {
  return gpt_gemm_tn_grouped(1, &P, &ldp, &Q, &ldq, &C, &ldc, &I, &J, R, accumulate, stream);
}

extern "C" int gpt_gemm_tn_grouped(int n, const void* const* P, const int64_t* ldp, const void* const* Q, const int64_t* ldq,
                                   float* const* C, const int64_t* ldc, const int* I, const int* J, int R, int accumulate,
                                   void* stream) {
  return gpt_tn_grouped_sumsq(n, P, ldp, Q, ldq, C, ldc, I, J, R, accumulate, nullptr, stream);
}

int gpt_tn_grouped_sumsq(int n, const void* const* P, const int64_t* ldp, const void* const* Q, const int64_t* ldq, float* const* C,
                         const int64_t* ldc, const int* I, const int* J, int R, int accumulate, float* sumsq, void* stream) {
  int Rs[gemmk::TN_MAX_GROUP];
  for (int p = 0; p < gemmk::TN_MAX_GROUP; ++p) Rs[p] = R;
  return gpt_tn_grouped_sumsq_r(n, P, ldp, Q, ldq, C, ldc, I, J, Rs, accumulate, sumsq, stream);
}

int gpt_tn_grouped_sumsq_r(int n, const void* const* P, const int64_t* ldp, const void* const* Q, const int64_t* ldq, float* const* C,
                           const int64_t* ldc, const int* I, const int* J, const int* R, int accumulate, float* sumsq, void* stream) {
  using namespace gemmk;
  GPT_REQUIRE(!sumsq || !accumulate, "gpt_gemm_tn_grouped: the sum of squares exists in store mode only");
  GPT_REQUIRE(n >= 1 && n <= TN_MAX_GROUP && P && ldp && Q && ldq && C && ldc && I && J && R, "gpt_gemm_tn_grouped: need 1..%d problems", TN_MAX_GROUP);
  TnGroup grp;
  grp.n = n;
  grp.sumsq = sumsq;
  for (int p = 0; p < n; ++p) {
    const int rc = tn_check(P[p], ldp[p], Q[p], ldq[p], C[p], ldc[p], I[p], J[p], R[p]);
    if (rc != 0) return rc;
    grp.g[p] = GemmArgs<bf16_t>{(const bf16_t*)P[p], (const bf16_t*)Q[p], nullptr, C[p], nullptr, ldp[p], ldq[p], ldc[p], 0, I[p], J[p], R[p]};
  }
  // (an accumulating launch may cut the contractions into spans: ONE span count for the group, chosen for the first problem; every
  // problem cuts its own length by it and the kernel drops the spans a shorter problem leaves empty)
  bool ragged = false;
  for (int p = 0; p < n; ++p) ragged = ragged || R[p] % 64 != 0;
  hipStream_t hs = (hipStream_t)stream;
  if (ragged) return accumulate ? launch_tn<GPT_EPI_ACCUM_F32, true>(grp, hs) : launch_tn<GPT_EPI_STORE_F32, true>(grp, hs);
  return accumulate ? launch_tn<GPT_EPI_ACCUM_F32, false>(grp, hs) : launch_tn<GPT_EPI_STORE_F32, false>(grp, hs);
}
