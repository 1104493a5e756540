// 128 x 256 block-tile bf16 MFMA GEMM (C = A W^T, epilogues of gpt_gemm_nt) for problems with too few 256-row tiles to give every
// CU one: sampling at B = 64 is M = 1664 rows (qkv: 7 x 18 = 126 tiles of 256 x 256, but 13 x 18 = 234 of 128 x 256).
//
// r01 ran these on the 256 x 256 kernel with half of every wave's rows switched off (RT = 4): its schedule still moved the
// unused A half-tile and kept the second, now empty, phase of every K-tile with both of its barriers: ~2100 cycles per K-tile
// for 1024 cycles of MFMA.  This kernel is the same machinery cut for the smaller tile:
//   * 8 waves (2 x 4), each 64 x 64 of the output = 4 x 4 MFMA 16x16x32 tiles (64 accumulator registers);
//   * a K-tile (64 of K) is ONE phase: 16 ds_read_b128 + 32 MFMAs per wave; the two wave groups (wr = 0 / 1: one wave of each on
//     every SIMD) run one barrier apart, so one group's MFMAs cover the other's reads and DMA issue (as in gemm256.hip);
//   * a stage is 48 KiB (128 A rows + 256 W rows of 128 bytes, same XOR swizzle); THREE stages, refilled two K-tiles ahead:
//     6 LDS-DMA instructions per wave and K-tile, issued in the load segment of K-tile kt for kt + 2 into the stage kt - 1 was
//     read from;
//   * counted waits, placed per group so that every piece has >= 1.5 K-tiles to land: a stage may be read after barrier
//     instance 2 kt (group 0's first barrier of K-tile kt, group 1's second of kt - 1); group 0 retires its pieces of kt right
//     before it (vmcnt(6): the pieces of kt + 1 stay in flight), group 1 after issuing kt + 1's refill, one load segment later.
// The vector-memory path, not the matrix pipe, bounds this tile: 48 KiB per K-tile at the 47 B/cycle the path delivers
// (tools/dma_probe.hip) is 1045 cycles against 1024 of MFMA.
#include "gemm_common.h"

namespace gemmk {
namespace {

constexpr int RB = 128;                    // bytes of K per LDS row (64 bf16)
constexpr int A_ROWS = 128, W_ROWS = 256;
constexpr int STAGE = (A_ROWS + W_ROWS) * RB;   // 48 KiB
constexpr int LDS128_BYTES = 3 * STAGE;         // 144 KiB
// diagnostic builds only (tools/gemm_ablate.sh with FILE=gemm128; results are garbage): bit 0 = no DMA in the steady loop, bit 1 = no
// fragment reads, bit 2 = no barriers in the loop, bit 3 = no MFMAs
#ifndef GPT_ABL128
#define GPT_ABL128 0
#endif

__device__ __forceinline__ u32x4 frag128(const char* base, int row, int lc) {
  return *reinterpret_cast<const u32x4*>(base + row * RB + ((lc ^ ((row >> 1) & 7)) << 4));
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm128_kernel(GemmArgs<bf16_t> g) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  const int nwg = gridDim.x / g.ksplit, span = blockIdx.x / nwg, bid = blockIdx.x - span * nwg;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);   // XCD-contiguous runs of tiles
  const int tiles_n = (g.N + 255) / 256, tiles_m = (g.M + A_ROWS - 1) / A_ROWS;
  constexpr int GROUP_N = 4;              // column groups, columns fastest inside a group (see gemm256.hip)
  const int gb = tile / (tiles_m * GROUP_N), grem = tile - gb * (tiles_m * GROUP_N);
  const int gw = min(GROUP_N, tiles_n - gb * GROUP_N);
  const int m0 = (grem / gw) * A_ROWS, n0 = (gb * GROUP_N + grem % gw) * 256;

  constexpr int BK = 64;
  const int nk_tile = g.K / BK;
  const int nk_span = (nk_tile + g.ksplit - 1) / g.ksplit, kt0 = span * nk_span;
  const int nk = min(nk_span, nk_tile - kt0);    // host: every span gets at least one K-tile

  // this wave's 6 DMA pieces of a stage: LDS rows 8 (6 wid + p) + (lane >> 3); rows [0, 128) = A, [128, 384) = W
  const T* src[6];
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const int lr = (wid * 6 + p) * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((lr >> 1) & 7);
    if (lr < A_ROWS) {
      const int gm = min(m0 + lr, g.M - 1);
      src[p] = g.A + (int64_t)gm * g.lda + lc * 8 + (int64_t)kt0 * BK;
    } else {
      const int gn = min(n0 + lr - A_ROWS, g.N - 1);
      src[p] = g.W + (int64_t)gn * g.ldw + lc * 8 + (int64_t)kt0 * BK;
    }
  }
  auto dma = [&](int kt, int stage_off) {
    char* dst = smem + stage_off + wid * 6 * 1024;
#pragma unroll
    for (int p = 0; p < 6; ++p) __builtin_amdgcn_global_load_lds((gptr_t)(src[p] + kt * BK), (lptr_t)(dst + p * 1024), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  dma(0, 0);
  if (nk > 1) dma(1, STAGE);
  if (nk > 1) wait_vm<6>();
  else wait_vm<0>();
  if (wr == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind group 0

  const int l15 = lane & 15, l4 = lane >> 4;
#define SEG_BARRIER()                        \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    if (!(GPT_ABL128 & 4)) __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)

  int s0 = 0, s1 = STAGE, s2 = 2 * STAGE;               // stages of K-tiles kt, kt + 1, kt + 2
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 2 < nk;
    if (wr == 0) {                                      // group 0 retires its pieces of kt before barrier instance 2 kt
      if (kt + 1 < nk) wait_vm<6>();
      else wait_vm<0>();
    }
    SEG_BARRIER();
    const char* sa = smem + s0;
    const char* sw = sa + A_ROWS * RB;
    u32x4 a[4][2], b[4][2];
    if (GPT_ABL128 & 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { asm volatile("" : "=v"(a[j][kk])); asm volatile("" : "=v"(b[j][kk])); }
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) b[j][kk] = frag128(sw, wc * 64 + j * 16 + l15, 4 * kk + l4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) a[i][kk] = frag128(sa, wr * 64 + i * 16 + l15, 4 * kk + l4);
    }
    if (more && !((GPT_ABL128 & 1) && kt >= 2)) dma(kt + 2, s2);   // the stage K-tile kt - 1 was read from
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wr == 1) {                                      // group 1: its pieces of kt + 1, before ITS barrier instance 2 (kt + 1)
      if (more) wait_vm<6>();
      else wait_vm<0>();
    }
    SEG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    if (!(GPT_ABL128 & 8))
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][kk]), __builtin_bit_cast(bf16x8, b[j][kk]),
                                                              acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    const int t = s0;
    s0 = s1;
    s1 = s2;
    s2 = t;
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();            // balance group 1's extra barrier
#undef SEG_BARRIER
  __syncthreads();

  // epilogue through a private LDS patch per wave, 32 rows x 64 columns at a time (gemm_common.h)
  float* patch = reinterpret_cast<float*>(smem + wid * PATCH_BYTES);
  auto to_patch = [&](int step) {
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) patch[(ii * 16 + 4 * l4 + e) * PATCH_PITCH + jt * 16 + l15] = acc[2 * step + ii][jt][e];
  };
  GemmArgs<T> ge = g;
  if (g.ksplit > 1) ge.C = (float*)g.C + (int64_t)span * g.M * g.ldc;   // this span's slab of the workspace (STORE_F32)
  epilogue_wave<T, EPI, 2, 64>(ge, to_patch, patch, m0 + wr * 64, n0 + wc * 64, lane);
}

template <int EPI>
int launch128(const GemmArgs<bf16_t>& g, hipStream_t s) {
  static GptPerDevice attr;   // the LDS grant is per kernel AND per device (a process may drive several GPUs)
  bool& attr_set = attr.here();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm128_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS128_BYTES) != hipSuccess) {
      gpt_set_error("gpt_gemm_nt: cannot reserve %d bytes of LDS", LDS128_BYTES);
      return -1;
    }
    attr_set = true;
  }
  const int tiles = ((g.M + A_ROWS - 1) / A_ROWS) * ((g.N + 255) / 256);
  const bool rec = g_prof.on && g_prof.n < GemmProf::CAP;
  if (rec) (void)hipEventRecord(g_prof.ev[g_prof.n][0], s);
  hipLaunchKernelGGL((gemm128_kernel<EPI>), dim3(tiles * g.ksplit), dim3(512), LDS128_BYTES, s, g);
  if (rec) {
    (void)hipEventRecord(g_prof.ev[g_prof.n][1], s);
    g_prof.flops[g_prof.n++] = 2.0 * g.M * g.N * g.K;
  }
  return gpt_launch_status("gpt_gemm_nt");
}

}  // namespace

// needs K % 64 == 0; any M, N (edge rows / columns are clamped on load and masked on store)
int gemm128_bf16(const GemmArgs<bf16_t>& g, int epi, hipStream_t s) {
  switch (epi) {
    case GPT_EPI_STORE: return launch128<GPT_EPI_STORE>(g, s);
    case GPT_EPI_GELU: return launch128<GPT_EPI_GELU>(g, s);
    case GPT_EPI_RESADD: return launch128<GPT_EPI_RESADD>(g, s);
    case GPT_EPI_STORE_F32: return launch128<GPT_EPI_STORE_F32>(g, s);
    case GPT_EPI_GELU_AUX: return launch128<GPT_EPI_GELU_AUX>(g, s);
    case GPT_EPI_DGELU: return launch128<GPT_EPI_DGELU>(g, s);
    case GPT_EPI_ACCUM_F32: return launch128<GPT_EPI_ACCUM_F32>(g, s);
  }
  gpt_set_error("gpt_gemm_nt: unknown epilogue %d", epi);
  return -1;
}

}  // namespace gemmk
