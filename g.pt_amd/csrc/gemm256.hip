// 256 x 256 block-tile MFMA GEMM for gfx950 (the large-problem path of gpt_gemm_nt).
//
// 8 waves (2 x 4), each owning 128 x 64 of the output: bf16 as 8 x 4 MFMA 16x16x32 tiles, fp32 as 4 x 2 MFMA 32x32x2
// tiles (128 accumulator registers either way).  K-tile = 128 bytes of K per row (64 bf16 / 32 fp32).
// LDS = 2 K-tile buffers x 64 KiB, each buffer split into FOUR 16-KiB half-tiles: A-h0 / A-h1 hold the first /
// second 64 rows of every wave-row's 128, W-h0 / W-h1 the first / second 32 columns of every wave-column's 64.
// A K-tile is consumed in TWO phases of two 64 x 32 quadrants each:
//     P0: reads A-h0, W-h0, W-h1 -> (A-h0, W-h0), (A-h0, W-h1)     P1: reads A-h1 -> (A-h1, W-h1), (A-h1, W-h0)
// and the LDS-DMA refills of a buffer (for K-tile t+2) are issued as soon as its slots have been read:
//     P1(t): A-h0, W-h0, W-h1 (t+2)      P0(t+1): A-h1 (t+2)
// so every DMA has about three phases (1.5 K-tiles of MFMA work) to land, with only two buffers.
// Waits are COUNTED (s_waitcnt vmcnt(6) / vmcnt(2): the DMA instructions issued after the half-tiles a phase
// retires may still be in flight) and followed by a raw s_barrier (a __syncthreads() would drain the DMA queue).
// WAR safety: a slot read in a phase is refilled after the barrier that opens the next phase, which every wave
// reaches only after its own reads have returned (lgkmcnt(0) before its MFMAs).
// r01 measurements that shaped it (tools/gemm_stamps.py, s_memtime stamps): four phases per K-tile cost 2970
// cycles per K-tile against 2048 of MFMA issue, most of the gap being barrier hand-offs; two phases: 2700.  The chip
// is power-limited in this loop (1.6 - 2.1 GHz sustained): part of every cycle saved comes back as a lower clock, and
// the 16x16x32 MFMA shape sustains a higher clock than 32x32x16 at equal cycles and LDS traffic (+4...8 % wall).
//
// Operand traffic per FLOP is half that of a 128 x 128 tile (L2 -> LDS is the scarce resource here: a CU's
// vector-memory path moves 64 B/clk), LDS reads per MFMA 0.75 instead of 1, DMA instructions per MFMA 0.25
// instead of 0.5.  One workgroup per CU (128 KiB LDS).
#include "gemm_common.h"

#ifdef GPT_STAMPS
// diagnostic build only: per-workgroup s_memtime stamps (no output value depends on them)
__device__ unsigned long long g_stamps[2048 * 8];
#define STAMP(i)                                                                        \
  do {                                                                                  \
    if (threadIdx.x == 0) g_stamps[(blockIdx.x & 2047) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define STAMP_RT(i)                                                                     \
  do {                                                                                  \
    if (threadIdx.x == 0) g_stamps[(blockIdx.x & 2047) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" int gpt_debug_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#else
#define STAMP(i)
#define STAMP_RT(i)
#endif

namespace gemmk {
namespace {

constexpr int RB = 128;             // bytes of K per LDS row
constexpr int HALF = 128 * RB;      // 16 KiB half-tile: 128 LDS rows
constexpr int BUF = 4 * HALF;       // A-h0 | A-h1 | W-h0 | W-h1
// diagnostic builds only (tools/gemm_ablate.sh; results are garbage): bit 0 = no DMA in the steady loop, bit 1 = no fragment
// reads, bit 2 = no barriers in the loop
#ifndef GPT_ABL
#define GPT_ABL 0
#endif
constexpr int LDS_BYTES = 2 * BUF;  // 128 KiB

// 128-byte rows: chunk c of LDS row r lives at chunk c ^ ((r >> 1) & 7): conflict-free ds_read_b128
__device__ __forceinline__ u32x4 frag(const char* half, int row, int lc) {
  return *reinterpret_cast<const u32x4*>(half + row * RB + ((lc ^ ((row >> 1) & 7)) << 4));
}

// (A stream-K variant - one workgroup per CU, equal spans of the tile x K-tile space, partial tiles added with float
// atomics - was built and measured in r01: slower on every shape of this model, proj 70 -> 154 us; removed.)
// RT = 16-row MFMA tiles per wave: 8 -> the 256-row block tile; 7 -> a 224-row block tile (bf16 only) for problems whose
// row count fills the CUs badly in 256-row tiles (M = 8320 = 32.5 x 256: proj / fc2 put 198 tiles on 256 CUs; in 224-row
// tiles it is 228 tiles of 7/8 the work).  The LDS image and the DMA schedule do not change: A-h1 simply carries 16 unused
// rows per wave-row (their sources are clamped to valid rows).
template <typename T, int EPI, int RT>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs<T> g) {
  static_assert(RT == 8 || ((RT == 7 || RT == 6 || RT == 4) && sizeof(T) == 2), "224-, 192- and 128-row tiles exist for the 16x16x32 bf16 path only");
  constexpr int BM = 32 * RT, WROWS = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  STAMP_RT(4);
  STAMP(0);
  const int nwg = gridDim.x / g.ksplit, span = blockIdx.x / nwg, bid = blockIdx.x - span * nwg;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tiles_n = (g.N + 255) / 256;

  constexpr int BK = RB / sizeof(T);
  constexpr int EPC = 16 / sizeof(T);
  const int nk_tile = g.K / BK;
  const int fr = lane & 31, fh = lane >> 5;

  const int nk_span = (nk_tile + g.ksplit - 1) / g.ksplit, kt0 = span * nk_span;
  const int tile = swz, nk = min(nk_span, nk_tile - kt0);   // host: every span gets at least one K-tile
  // tile order inside an XCD's contiguous run: column groups of GROUP_N tiles, columns fastest inside a group, so the 32
  // tiles an XCD works on at once form an 8-row x 4-column block: its 4 W panels (3 MiB at K = 1536) stay in the XCD's 4 MiB
  // L2 from one round of tiles to the next while the A panels stream.  r02 PMC (FETCH_SIZE) with groups of 8: qkv fetched
  // 239 MB per launch for 40 MB of operands - every round of 32 tiles re-read 4 A + 8 W panels (9 MiB > L2); timing is
  // insensitive (groups of 2 / 4 / 8 / 16: 489.7 / 483.5 / 486.3 / 495.1 us per layer, the re-reads are Infinity-Cache hits)
#ifndef GPT_GROUP_N
#define GPT_GROUP_N 4
#endif
  constexpr int GROUP_N = GPT_GROUP_N;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int gb = tile / (tiles_m * GROUP_N), grem = tile - gb * (tiles_m * GROUP_N);
  const int gw = min(GROUP_N, tiles_n - gb * GROUP_N);
  const int m0 = (grem / gw) * BM, n0 = (gb * GROUP_N + grem % gw) * 256;

  // per-lane DMA source rows: this wave fills LDS rows [wid*16, wid*16+16) of every half-tile (2 pieces of 8 rows)
  const T* srcA[2][2];  // [half][piece]
  const T* srcW[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int lr = wid * 16 + p * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((lr >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int gm = m0 + (lr >> 6) * WROWS + min(h * 64 + (lr & 63), WROWS - 1);
      gm = gm < g.M ? gm : g.M - 1;
      srcA[h][p] = g.A + (int64_t)gm * g.lda + lc * EPC + (int64_t)kt0 * BK;
      int gn = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
      gn = gn < g.N ? gn : g.N - 1;
      srcW[h][p] = g.W + (int64_t)gn * g.ldw + lc * EPC + (int64_t)kt0 * BK;
    }
  }
  auto dma_at = [&](const T* const (&src)[2], int kt, int off) {
    if ((GPT_ABL & 1) && kt >= 2) return;
    char* dst = smem + off + wid * 2048;
    __builtin_amdgcn_global_load_lds((gptr_t)(src[0] + kt * BK), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src[1] + kt * BK), (lptr_t)(dst + 1024), 16, 0, 0);
  };
  auto dma = [&](const T* const (&src)[2], int kt, int slot) { dma_at(src, kt, (kt & 1) * BUF + slot * HALF); };

  // bf16: 16x16x32 MFMAs, 8 x 4 accumulator tiles of 16 x 16 (the chip holds a higher clock on this shape than on
  // 32x32x16 at the same cycles per FLOP and the same LDS traffic: +4...8 % measured);  fp32: 32x32x2, 4 x 2 tiles of 32 x 32
  constexpr bool M16 = sizeof(T) == 2;
  f32x4 acc16[M16 ? RT : 1][M16 ? 4 : 1];
  f32x16 acc[M16 ? 1 : 4][M16 ? 1 : 2];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }

  // prologue: K-tiles 0 and 1 in consumption order
#pragma unroll
  for (int t = 0; t < 2; ++t)
    if (t < nk) {
      dma(srcA[0], t, 0);
      dma(srcW[0], t, 2);
      dma(srcW[1], t, 3);
      dma(srcA[1], t, 1);
    }

  // one quadrant of the wave tile x one K-tile.  Fragment arrays: fp32 / 32x32: a[i][ks] = rows 32 i.., k-step ks of 4;
  // bf16 / 16x16: the same 8 + 4 registers viewed as a[i][kk] -> a[i >> 1][2 * (i & 1) + kk], i < 4 row tiles, kk < 2
  auto mma = [&](const u32x4 (&a)[2][4], const u32x4 (&b)[4], int i0, int j) {
    if constexpr (M16) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            if (4 * (i0 >> 1) + i < RT)      // compile time once unrolled (i0 is a literal at every call site)
              acc16[4 * (i0 >> 1) + i][2 * j + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(bf16x8, a[i >> 1][2 * (i & 1) + kk]), __builtin_bit_cast(bf16x8, b[2 * jj + kk]),
                  acc16[4 * (i0 >> 1) + i][2 * j + jj], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i][ks][e]), __uint_as_float(b[ks][e]),
                                                                  acc[i0 + i][j], 0, 0, 0);
    }
  };
  // fragment reads of one half-tile: A half = 64 rows of this wave (base row wr * 64), W half = 32 columns (base wc * 32)
  const int l15 = lane & 15, l4 = lane >> 4;
  auto read_a = [&](u32x4 (&a)[2][4], const char* half, int h) {
    if (GPT_ABL & 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "=v"(a[i >> 2][i & 3]));
      return;
    }
    if constexpr (M16) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          if (4 * h + i < RT) a[i >> 1][2 * (i & 1) + kk] = frag(half, wr * 64 + i * 16 + l15, 4 * kk + l4);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[i][ks] = frag(half, wr * 64 + i * 32 + fr, 2 * ks + fh);
    }
  };
  auto read_b = [&](u32x4 (&b)[4], const char* half) {
    if (GPT_ABL & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(b[i]));
      return;
    }
    if constexpr (M16) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b[2 * jj + kk] = frag(half, wc * 32 + jj * 16 + l15, 4 * kk + l4);
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b[ks] = frag(half, wc * 32 + fr, 2 * ks + fh);
    }
  };
  // Two wave groups (wr = 0 / 1: one wave of each on every SIMD) run ONE BARRIER APART: every phase is
  // {load segment: wait, barrier, ds_reads, DMA issue} barrier {compute segment: 32 MFMAs}, so while one group
  // multiplies, its SIMD partner reads LDS and issues DMA; the matrix pipe alternates between the two.
  // Because of the offset a wave retires (vmcnt) the DMA of a half-tile ONE PHASE BEFORE the phase that reads
  // it: the other group's pieces are then covered by a barrier that group has passed as well.
#define SEG_BARRIER()                                  \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    if (!(GPT_ABL & 4)) __builtin_amdgcn_s_barrier();  \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
#define SEG_READS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

  // TWO phases per K-tile (32 MFMAs each) instead of four: half the barriers, whose hand-off latency (~100 cycles of
  // idle matrix pipe per interval) was the largest loss of the 4-phase loop (2970 cycles per K-tile against 2048 of MFMA).
  //   P0: reads A-h0, W-h0, W-h1 (16 ds_read_b128)  -> (A-h0 x W-h0), (A-h0 x W-h1);   refills A-h1(kt+1)
  //   P1: reads A-h1 (8)                            -> (A-h1 x W-h1), (A-h1 x W-h0);   refills A-h0, W-h0, W-h1 (kt+2)
  // Retire-one-phase-ahead waits: P0 retires A-h1(kt) (6 younger DMA instructions), P1 retires the three half-tiles of
  // kt+1 (2 younger).
  // (r02: W-h1 in a ring of three slots, so that the refills are 4 + 4 instructions per wave and phase instead of 2 + 6 with
  // every half-tile still >= 3 phases ahead of its reads, changed nothing: 470 -> 475 us per layer; removed.  The ablation of
  // this loop - tools/gemm_ablate.sh, DESIGN.md section 4 - puts the loss in the LDS array, not in the issue order.)
  if (nk >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  STAMP(1);
  if (wr == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind group 0

  for (int kt = 0; kt < nk; ++kt) {
    const char* buf = smem + (kt & 1) * BUF;
    const bool steady = kt + 2 < nk, last = kt + 1 == nk;
    u32x4 a[2][4], b0[4], b1[4];
    // ---- P0
    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    SEG_BARRIER();
    read_b(b0, buf + 2 * HALF);
    read_a(a, buf, 0);
    read_b(b1, buf + 3 * HALF);
    if (kt >= 1 && !last) dma(srcA[1], kt + 1, 1);          // A-h1(kt+1): its slot was last read in P1(kt-1)
    SEG_READS_DONE();
    SEG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    mma(a, b0, 0, 0);
    mma(a, b1, 0, 1);
    __builtin_amdgcn_s_setprio(0);
    // ---- P1
    if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    SEG_BARRIER();
    read_a(a, buf + HALF, 1);
    if (steady) {
      dma(srcA[0], kt + 2, 0);
      dma(srcW[0], kt + 2, 2);
      dma(srcW[1], kt + 2, 3);
    }
    SEG_READS_DONE();
    SEG_BARRIER();
    __builtin_amdgcn_s_setprio(1);
    mma(a, b1, 2, 1);
    mma(a, b0, 2, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();            // balance group 1's extra barrier
#undef SEG_BARRIER
#undef SEG_READS_DONE
  __syncthreads();
  STAMP(2);

  // epilogue through a private LDS patch per wave, 32 rows x 64 columns at a time (see gemm_common.h)
  float* patch = reinterpret_cast<float*>(smem + wid * PATCH_BYTES);
  auto to_patch = [&](int step) {
    if constexpr (M16) {
      // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (2 * step + ii < RT) patch[(ii * 16 + 4 * l4 + e) * PATCH_PITCH + jt * 16 + l15] = acc16[2 * step + ii][jt][e];
    } else {
      // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + 4 * fh) * PATCH_PITCH + j * 32 + fr] = acc[step][j][e];
    }
  };
  GemmArgs<T> ge = g;
  if (g.ksplit > 1) ge.C = (float*)g.C + (int64_t)span * g.M * g.ldc;   // this span's slab of the workspace (STORE_F32)
  epilogue_wave<T, EPI, 4, WROWS>(ge, to_patch, patch, m0 + wr * WROWS, n0 + wc * 64, lane);
#ifdef GPT_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  STAMP(3);
  STAMP_RT(5);
#endif
}

template <typename T, int EPI, int RT>
int launch256(const GemmArgs<T>& g, hipStream_t s) {
  static GptPerDevice attr;   // the LDS grant is per kernel AND per device (a process may drive several GPUs)
  bool& attr_set = attr.here();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm256_kernel<T, EPI, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
        hipSuccess) {
      gpt_set_error("gpt_gemm_nt: cannot reserve %d bytes of LDS", LDS_BYTES);
      return -1;
    }
    attr_set = true;
  }
  const int tiles = ((g.M + 32 * RT - 1) / (32 * RT)) * ((g.N + 255) / 256);
  const int grid = tiles * g.ksplit;
  const bool rec = g_prof.on && g_prof.n < GemmProf::CAP;
  if (rec) (void)hipEventRecord(g_prof.ev[g_prof.n][0], s);
  hipLaunchKernelGGL((gemm256_kernel<T, EPI, RT>), dim3(grid), dim3(512), LDS_BYTES, s, g);
  if (rec) {
    (void)hipEventRecord(g_prof.ev[g_prof.n][1], s);
    g_prof.flops[g_prof.n++] = 2.0 * g.M * g.N * g.K;
  }
  return gpt_launch_status("gpt_gemm_nt");
}

template <typename T, int RT>
int dispatch256(const GemmArgs<T>& g, int epi, hipStream_t s) {
  switch (epi) {
    case GPT_EPI_STORE: return launch256<T, GPT_EPI_STORE, RT>(g, s);
    case GPT_EPI_GELU: return launch256<T, GPT_EPI_GELU, RT>(g, s);
    case GPT_EPI_RESADD: return launch256<T, GPT_EPI_RESADD, RT>(g, s);
    case GPT_EPI_STORE_F32: return launch256<T, GPT_EPI_STORE_F32, RT>(g, s);
    case GPT_EPI_GELU_AUX: return launch256<T, GPT_EPI_GELU_AUX, RT>(g, s);
    case GPT_EPI_DGELU: return launch256<T, GPT_EPI_DGELU, RT>(g, s);
    case GPT_EPI_ACCUM_F32: return launch256<T, GPT_EPI_ACCUM_F32, RT>(g, s);
  }
  gpt_set_error("gpt_gemm_nt: unknown epilogue %d", epi);
  return -1;
}

}  // namespace

// rt = 16-row tiles per wave: 8 (256-row block tiles), 7 (224-row), 6 (192-row: M = 1664 = 8.67 x 192 puts 216 tiles of
// 3/4 the work on the chip where 224-row tiles put 192 of 7/8, the last of them nearly empty) or 4 (128-row)
int gemm256_bf16(const GemmArgs<bf16_t>& g, int epi, int rt, hipStream_t s) {
  if (rt == 4) return dispatch256<bf16_t, 4>(g, epi, s);
  if (rt == 6) return dispatch256<bf16_t, 6>(g, epi, s);
  return rt == 7 ? dispatch256<bf16_t, 7>(g, epi, s) : dispatch256<bf16_t, 8>(g, epi, s);
}
int gemm256_f32(const GemmArgs<float>& g, int epi, hipStream_t s) { return dispatch256<float, 8>(g, epi, s); }

}  // namespace gemmk
