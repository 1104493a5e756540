// MFMA GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)  for gfx950, fp32 (exact) and bf16.
//
// Replaces the nn.Linear contractions of Gpt/models/transformer.py:38-40,50,63-65 (99.3 % of the
// denoiser FLOPs).  Both operands are K-contiguous, which is the
// natural MFMA operand order, so nn.Linear's (out,in) weight layout is consumed as stored.
//
// One template, two tile shapes (same staging / fragment / epilogue code):
//   big   256 x 128 per 512-thread workgroup (8 waves as 4 x 2), used when it yields >= 1 workgroup per CU;
//   small 128 x 128 per 256-thread workgroup (4 waves as 2 x 2), for small problems.
// Every wave owns a 64 x 64 output = 2 x 2 MFMA 32x32 tiles.  A K-step stages 64 BYTES of K per row
// (32 bf16 / 16 fp32), so the LDS image and the staging code are identical for both dtypes.
//
// Pipeline: a ring of LDS stages filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction,
// lane-linear destination, per-lane source).  The DMA of K-tile t+NSTAGE-1 is issued right after the barrier
// that opens K-tile t and stays in flight across the following barriers: each wave waits with a COUNTED
// s_waitcnt vmcnt(n) (only its own pieces of the tiles still ahead may be pending) and a raw s_barrier -
// __syncthreads() would drain the DMA queue with vmcnt(0).  One barrier per K-tile.
// LDS per workgroup is 72 KiB (big, 3 stages) / 64 KiB (small, 4 stages): TWO workgroups are resident per CU,
// so one workgroup's epilogue (global stores, residual read-modify-write) overlaps the other's main loop.
//
// Bank conflicts: 64-byte LDS rows; the 16-byte chunk c of row r is stored at chunk c ^ swz(r) (applied on the
// per-lane DMA SOURCE address and again on the ds_read_b128 address): conflict-free for the 16-lane groups of
// ds_read_b128 under each dtype's fragment shape (fp32, 32x32x2: rows lane & 31, swz = (r >> 2) & 3;
// bf16, 16x16x32: rows lane & 15 and chunk lane >> 4, swz = (-(r >> 2)) & 3).
//
// Epilogue: each wave passes its accumulators through a private 4 KiB LDS patch, 32 x 32 at a time, so
// that global accesses are 16-byte (fp32) / 8-byte (bf16) row vectors; no workgroup barrier is involved.
#include "gemm_common.h"
#include <stdlib.h>

#define CHECK_RC(call)          \
  do {                          \
    int _st = (call);           \
    if (_st != 0) return _st;   \
  } while (0)

namespace gemmk {
GemmProf g_prof_tab[GPT_MAX_DEVICES];
int gemm256_bf16(const GemmArgs<bf16_t>& g, int epi, int rt, hipStream_t s);  // gemm256.hip; rt = 8 | 7 (256- / 224-row tiles)
int gemm256_f32(const GemmArgs<float>& g, int epi, hipStream_t s);
int gemm128_bf16(const GemmArgs<bf16_t>& g, int epi, hipStream_t s);          // gemm128.hip: 128 x 256 tiles, three stages
}

namespace {
using namespace gemmk;

constexpr int ROWB = 64;  // bytes of K per staged row

template <typename T>
__device__ __forceinline__ int swz(int r) {
  if constexpr (sizeof(T) == 2) return (0 - (r >> 2)) & 3;
  else return (r >> 2) & 3;
}

// LDS-DMA of NP 1-KiB pieces (16 rows x 64 B each) of an operand panel
template <typename T, int NP>
__device__ __forceinline__ void stage_rows(const T* __restrict__ base, int64_t ld, int row0, int nrows_total, int k0,
                                           char* lds_tile, int first_piece, int lane) {
  constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int piece = first_piece + i;
    const int rr = piece * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ swz<T>(rr);
    int grow = row0 + rr;
    grow = grow < nrows_total ? grow : nrows_total - 1;  // clamp: rows past the edge are computed and dropped
    const T* src = base + (int64_t)grow * ld + k0 + lc * EPC;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

template <typename T>
__device__ __forceinline__ u32x4 read_frag(const char* lds_tile, int row, int lc) {
  const int off = row * ROWB + ((lc ^ swz<T>(row)) << 4);
  return *reinterpret_cast<const u32x4*>(lds_tile + off);
}

// WM x 2 waves of 64 x 64; BM = 64 * WM, BN = 128
template <typename T, int EPI, int WM, int NSTAGE>
__global__ __launch_bounds__(WM * 128, WM) void gemm_nt_kernel(GemmArgs<T> g) {
  constexpr int BM = 64 * WM, BN = 128, NWAVES = WM * 2;
  constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  constexpr int PIECES_A = BM / 16 / NWAVES, PIECES_W = BN / 16 / NWAVES;  // per wave: 2 + 1 (big), 2 + 2 (small)
  constexpr int DMA_PER_TILE = PIECES_A + PIECES_W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;

  // XCD-aware, bijective remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
  // contiguous run of tiles so neighbours re-use the same A panel / W panels from that L2.
  // K spans (g.ksplit > 1, GPT_EPI_STORE_F32 into a workspace): workgroup (span, tile) contracts its span of the K-tiles
  const int nwg = gridDim.x / g.ksplit, span = blockIdx.x / nwg, bid = blockIdx.x - span * nwg;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tm = swz / tiles_n, tn = swz % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  constexpr int BK = ROWB / sizeof(T);
  const int nk_all = g.K / BK, nk_span = (nk_all + g.ksplit - 1) / g.ksplit, kt0 = span * nk_span;
  const int nk = min(nk_span, nk_all - kt0);      // host: every span gets at least one K-tile

  // bf16: 16x16x32 MFMAs (one per 64-byte K-step and 16 x 16 tile; higher sustained clock than 32x32x16, see gemm256.hip),
  // 4 x 4 accumulator tiles;  fp32: 32x32x2, 2 x 2 tiles
  constexpr bool M16 = sizeof(T) == 2;
  f32x4 acc16[M16 ? 4 : 1][M16 ? 4 : 1];
  f32x16 acc[M16 ? 1 : 2][M16 ? 1 : 2];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }

  auto stage = [&](int kt) {
    char* buf = smem + (kt % NSTAGE) * STAGE_BYTES;
    stage_rows<T, PIECES_A>(g.A, g.lda, m0, g.M, (kt0 + kt) * BK, buf, wid * PIECES_A, lane);
    stage_rows<T, PIECES_W>(g.W, g.ldw, n0, g.N, (kt0 + kt) * BK, buf + BM * ROWB, wid * PIECES_W, lane);
  };
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s);

  const int fr = lane & 31, fh = lane >> 5, l15 = lane & 15, l4 = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    // K-tile kt has landed once at most the pieces of the tiles issued after it are pending
    const int ahead = nk - 1 - kt < NSTAGE - 2 ? nk - 1 - kt : NSTAGE - 2;
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_TILE) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1);
    const char* bufA = smem + (kt % NSTAGE) * STAGE_BYTES;
    const char* bufW = bufA + BM * ROWB;
    if constexpr (M16) {
      u32x4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = read_frag<T>(bufA, wr * 64 + i * 16 + l15, l4);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = read_frag<T>(bufW, wc * 64 + j * 16 + l15, l4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]),
                                                                acc16[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = read_frag<T>(bufA, wr * 64 + i * 32 + fr, 2 * ks + fh);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = read_frag<T>(bufW, wc * 64 + j * 32 + fr, 2 * ks + fh);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)   // 16 bytes = 4 fp32 k-values per lane half; element e of both halves forms one K=2 step
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i][e]), __uint_as_float(b[j][e]),
                                                               acc[i][j], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // every wave is done reading the last stage before the epilogue patches overwrite LDS

  float* patch = reinterpret_cast<float*>(smem + wid * PATCH_BYTES);
  auto to_patch = [&](int step) {
    if constexpr (M16) {
      // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int e = 0; e < 4; ++e) patch[(ii * 16 + 4 * l4 + e) * PATCH_PITCH + jt * 16 + l15] = acc16[2 * step + ii][jt][e];
    } else {
      // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + 4 * fh) * PATCH_PITCH + j * 32 + fr] = acc[step][j][e];
    }
  };
  if constexpr (EPI == GPT_EPI_STORE_F32) {
    if (g.ksplit > 1) {
      GemmArgs<T> ge = g;
      ge.C = (float*)g.C + (int64_t)span * g.M * g.ldc;   // this span's slab of the workspace
      epilogue_wave<T, EPI, 2>(ge, to_patch, patch, m0 + wr * 64, n0 + wc * 64, lane);
      return;
    }
  }
  epilogue_wave<T, EPI, 2>(g, to_patch, patch, m0 + wr * 64, n0 + wc * 64, lane);
}


template <typename T, int EPI, int WM, int NSTAGE>
int launch_shape(const GemmArgs<T>& g, hipStream_t s) {
  constexpr int BM = 64 * WM, BN = 128;
  constexpr int LDS = NSTAGE * (BM + BN) * ROWB;
  static GptPerDevice attr;   // the LDS grant is per kernel AND per device (a process may drive several GPUs)
  bool& attr_set = attr.here();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_nt_kernel<T, EPI, WM, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS) != hipSuccess) {
      gpt_set_error("gpt_gemm_nt: cannot reserve %d bytes of LDS", LDS);
      return -1;
    }
    attr_set = true;
  }
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  const bool rec = g_prof.on && g_prof.n < GemmProf::CAP;
  if (rec) (void)hipEventRecord(g_prof.ev[g_prof.n][0], s);
  hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, WM, NSTAGE>), dim3(tiles * g.ksplit), dim3(WM * 128), LDS, s, g);
  if (rec) {
    (void)hipEventRecord(g_prof.ev[g_prof.n][1], s);
    g_prof.flops[g_prof.n++] = 2.0 * g.M * g.N * g.K;
  }
  return gpt_launch_status("gpt_gemm_nt");
}

template <typename T, int EPI>
int launch(const GemmArgs<T>& g, hipStream_t s) {
  static int mode = [] {
    const char* e = getenv("GPT_GEMM_TILE");  // 0 = auto, 1 = 128 x 128, 2 = 256 x 128 (experiments)
    return e ? atoi(e) : 0;
  }();
  const int tiles_big = ((g.M + 255) / 256) * ((g.N + 127) / 128);
  const bool big = mode == 2 || (mode == 0 && tiles_big >= 256);
  return big ? launch_shape<T, EPI, 4, 3>(g, s) : launch_shape<T, EPI, 2, 4>(g, s);
}

int tile_mode() {
  static int mode = [] {
    const char* e = getenv("GPT_GEMM_TILE");  // 0 = auto, 1 = 128 x 128, 2 = 256 x 128, 3 = 256 x 256 (experiments)
    return e ? atoi(e) : 0;
  }();
  return mode;
}

template <typename T>
int dispatch_ring(const GemmArgs<T>& g, int epi, hipStream_t s);

// Rows [0, m_main) of the problem for the 256 x 256 kernel, the rest for the finer-grained ring kernels.
// The 256 x 256 kernel runs one workgroup per CU, so a launch costs ceil(tiles / 256) full rounds however empty the
// last one is (B = 320: 594 tiles -> 3 rounds at 77 %).  When the last round would be poorly filled, the row-tiles
// that do not fill whole rounds are handed to the 128 x 128 kernel (2 workgroups per CU, a quarter of the work each).
// ---- split-K through a caller-owned workspace -------------------------------------------------------------------------
// Problems with few 256 x 256 tiles and a long K (sampling at B = 64: 42 tiles; training at B = 128: 78) leave most CUs
// idle or fall to the slower 128 x 128 ring kernel.  With a workspace registered (gpt_gemm_set_workspace) the K range is
// cut into spans: pass 1 = the 256 x 256 kernel, one workgroup per (span, tile), fp32 partial tiles stored plainly (float
// atomics cost a CU ~50 ns per wave instruction: 51 us per tile, more than the tile's contraction); pass 2 = one
// streaming kernel that sums the spans and applies the real epilogue.
struct SplitWs {
  float* ptr = nullptr;
  size_t bytes = 0;
};
SplitWs g_split_ws_dev[32];   // per device: the registration applies to the device current at the call
thread_local SplitWs* g_ws_override = nullptr;   // set for the duration of a gpt_gemm_nt_static call (its static scratch)
SplitWs& split_ws() {
  if (g_ws_override) return *g_ws_override;
  int dev = 0;
  (void)hipGetDevice(&dev);
  return g_split_ws_dev[dev & 31];
}

// ---- the K-span scratch of gpt_gemm_nt_static: memory the library owns without ever allocating it ------------------------
// A file-scope device array: every GPU the library runs on holds its own copy at its own address from the moment the code
// object is loaded until the process ends.  The NAME `g_static_scratch` is a host-side stand-in, not a device address: host
// code must only ever use the per-device pointer hipGetSymbolAddress returns (static_scratch_ptr below).
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)GPT_GEMM_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char g_static_scratch[STATIC_SCRATCH_BYTES];
float* g_static_scratch_ptr[32] = {};

// device pointer of the CURRENT device's copy (looked up once per device, with that device current)
int static_scratch_ptr(float** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) {
    gpt_set_error("gpt_gemm_nt_static: no current device (or device index >= 32)");
    return -1;
  }
  if (!g_static_scratch_ptr[dev]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch));
    if (e != hipSuccess || !p) {
      gpt_set_error("gpt_gemm_nt_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return -1;
    }
    g_static_scratch_ptr[dev] = (float*)p;
  }
  *out = g_static_scratch_ptr[dev];
  return 0;
}

template <typename T, int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs<T> g, const float* __restrict__ part, int nspan) {
  const int groups = g.N >> 3;                       // 8 columns per thread
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)g.M * groups) return;
  const int m = (int)(idx / groups), n = (int)(idx - (int64_t)m * groups) * 8;
  const int64_t slab = (int64_t)g.M * g.N;
  const float* p = part + (int64_t)m * g.N + n;
  float4 v0 = reinterpret_cast<const float4*>(p)[0], v1 = reinterpret_cast<const float4*>(p)[1];
  for (int sp = 1; sp < nspan; ++sp) {
    const float4 a = reinterpret_cast<const float4*>(p + sp * slab)[0], b = reinterpret_cast<const float4*>(p + sp * slab)[1];
    v0 = add4(v0, a);
    v1 = add4(v1, b);
  }
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (g.bias) {
    b0 = reinterpret_cast<const float4*>(g.bias + n)[0];
    b1 = reinterpret_cast<const float4*>(g.bias + n)[1];
  }
  EpiPre pre;
  if constexpr (epi_reads<EPI>) epi_load8<T, EPI>(g, m, n, pre);
  epi_store8<T, EPI>(g, m, n, add4(v0, b0), add4(v1, b1), pre);
}

template <typename T, int EPI>
int launch_reduce(const GemmArgs<T>& g, const float* part, int nspan, hipStream_t s) {
  const int64_t total = (int64_t)g.M * (g.N >> 3);
  hipLaunchKernelGGL((splitk_reduce_kernel<T, EPI>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, part, nspan);
  return gpt_launch_status("gpt_gemm_nt");
}

template <typename T>
int dispatch_reduce(const GemmArgs<T>& g, int epi, const float* part, int nspan, hipStream_t s) {
  switch (epi) {
    case GPT_EPI_STORE: return launch_reduce<T, GPT_EPI_STORE>(g, part, nspan, s);
    case GPT_EPI_GELU: return launch_reduce<T, GPT_EPI_GELU>(g, part, nspan, s);
    case GPT_EPI_RESADD: return launch_reduce<T, GPT_EPI_RESADD>(g, part, nspan, s);
    case GPT_EPI_STORE_F32: return launch_reduce<T, GPT_EPI_STORE_F32>(g, part, nspan, s);
    case GPT_EPI_GELU_AUX: return launch_reduce<T, GPT_EPI_GELU_AUX>(g, part, nspan, s);
    case GPT_EPI_DGELU: return launch_reduce<T, GPT_EPI_DGELU>(g, part, nspan, s);
    case GPT_EPI_ACCUM_F32: return launch_reduce<T, GPT_EPI_ACCUM_F32>(g, part, nspan, s);
  }
  gpt_set_error("gpt_gemm_nt: unknown epilogue %d", epi);
  return -1;
}

// ---- residual GEMM + the LayerNorm that follows it, for small batches -----------------------------------------------------
// x_out = resid + A W^T + bias is always followed by a LayerNorm of x_out (transformer.py:70-71: ln2 after the attention
// projection, the next block's ln1 / ln_f after mlp.2).  When the GEMM runs in K spans anyway, pass 2 can be that LayerNorm:
// one wave per row sums the spans, adds bias and residual, writes the fp32 row and normalises it from registers - the reduce
// pass and the LayerNorm launch (6 us each at M = 1664) become one kernel, and the LayerNorm's re-read of x_out disappears.
// Same arithmetic, in the same order, as splitk_reduce_kernel followed by layernorm_kernel (norm.hip): bit-identical results.
template <int NV, typename OutT>
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(const float* __restrict__ part, int nspan, const float* __restrict__ bias,
                                                               const float* resid, int64_t ldr, float* x,   // may be the same rows
                                                               int64_t ldx, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, OutT* __restrict__ y,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows,
                                                               float eps) {
  constexpr int d = NV * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t slab = (int64_t)rows * d;
  const float4* pr = reinterpret_cast<const float4*>(part + (int64_t)row * d);
  const float4* rr = reinterpret_cast<const float4*>(resid + (int64_t)row * ldr);
  float4* xr = reinterpret_cast<float4*>(x + (int64_t)row * ldx);
  float4 v[NV], res[NV];
  float s = 0.f;
  // span loop outside, the row's NV loads of one span in flight together (the other nesting serialises NV x nspan latencies)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = pr[lane + i * 64];
    res[i] = rr[lane + i * 64];
  }
  for (int sp = 1; sp < nspan; ++sp) {
    const float4* ps = reinterpret_cast<const float4*>(part + sp * slab + (int64_t)row * d);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = add4(v[i], ps[lane + i * 64]);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (bias) v[i] = add4(v[i], reinterpret_cast<const float4*>(bias)[c]);
    v[i] = add4(res[i], v[i]);
    xr[c] = v[i];
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
    q += (a * a + b * b) + (c * c + e * e);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  OutT* yr = y + (int64_t)row * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    const float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
    float4 o;
    o.x = (v[i].x - mean) * rstd * g.x + b.x;
    o.y = (v[i].y - mean) * rstd * g.y + b.y;
    o.z = (v[i].z - mean) * rstd * g.z + b.z;
    o.w = (v[i].w - mean) * rstd * g.w + b.w;
    if constexpr (sizeof(OutT) == 4) {
      reinterpret_cast<float4*>(yr)[c] = o;
    } else {
      uint2 pk;
      pk.x = pack_bf16x2(o.x, o.y);
      pk.y = pack_bf16x2(o.z, o.w);
      reinterpret_cast<uint2*>(yr)[c] = pk;
    }
  }
}

// The same pass with WPR waves per row, one row per workgroup (each lane NV / WPR chunks of 16 bytes, the waves' sums meet in
// LDS): at the small batches this pass runs on (M = 1664: 6.5 rows per CU) a wave per row leaves the CUs with too few waves to
// hide the two memory latencies of a row.  The row sums are taken in another order than splitk_reduce_ln_kernel's: results agree
// to rounding.
template <int NV, int WPR, typename OutT>
__global__ __launch_bounds__(WPR * 64) void splitk_reduce_ln2_kernel(const float* __restrict__ part, int nspan, const float* __restrict__ bias,
                                                                     const float* resid, int64_t ldr, float* x, int64_t ldx,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     OutT* __restrict__ y, float* __restrict__ mean_out,
                                                                     float* __restrict__ rstd_out, int rows, float eps) {
  static_assert(NV % WPR == 0, "the waves of a row take equal shares");
  constexpr int d = NV * 256, H = NV / WPR;
  __shared__ float red[2][WPR];                           // [sum | sumsq][wave]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x;
  const int64_t slab = (int64_t)rows * d;
  const int c0 = wave * H * 64 + lane;
  const float4* pr = reinterpret_cast<const float4*>(part + (int64_t)row * d);
  const float4* rr = reinterpret_cast<const float4*>(resid + (int64_t)row * ldr);
  float4* xr = reinterpret_cast<float4*>(x + (int64_t)row * ldx);
  float4 v[H], res[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    v[i] = pr[c0 + i * 64];
    res[i] = rr[c0 + i * 64];
  }
  for (int sp = 1; sp < nspan; ++sp) {
    const float4* ps = reinterpret_cast<const float4*>(part + sp * slab + (int64_t)row * d);
#pragma unroll
    for (int i = 0; i < H; ++i) v[i] = add4(v[i], ps[c0 + i * 64]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const int c = c0 + i * 64;
    if (bias) v[i] = add4(v[i], reinterpret_cast<const float4*>(bias)[c]);
    v[i] = add4(res[i], v[i]);
    xr[c] = v[i];
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  s = wave_sum(s);
  if (lane == 0) red[0][wave] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < WPR; ++w) tot += red[0][w];
  const float mean = tot / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
    q += (a * a + b * b) + (c * c + e * e);
  }
  q = wave_sum(q);
  if (lane == 0) red[1][wave] = q;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int w = 0; w < WPR; ++w) tot += red[1][w];
  const float rstd = 1.0f / sqrtf(tot / (float)d + eps);
  if (threadIdx.x == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  OutT* yr = y + (int64_t)row * d;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const int c = c0 + i * 64;
    const float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
    float4 o;
    o.x = (v[i].x - mean) * rstd * g.x + b.x;
    o.y = (v[i].y - mean) * rstd * g.y + b.y;
    o.z = (v[i].z - mean) * rstd * g.z + b.z;
    o.w = (v[i].w - mean) * rstd * g.w + b.w;
    if constexpr (sizeof(OutT) == 4) {
      reinterpret_cast<float4*>(yr)[c] = o;
    } else {
      uint2 pk;
      pk.x = pack_bf16x2(o.x, o.y);
      pk.y = pack_bf16x2(o.z, o.w);
      reinterpret_cast<uint2*>(yr)[c] = pk;
    }
  }
}

int n_cus();

// 128-row tiles: the three-stage 128 x 256 kernel (gemm128.hip); GPT_GEMM_128=0 falls back to the 256 x 256 kernel with half of
// every wave's rows switched off (the r01 route, kept for A/B measurements)
int half_height_bf16(const GemmArgs<bf16_t>& g, int epi, hipStream_t s) {
  static const bool k128 = [] { const char* e = getenv("GPT_GEMM_128"); return !e || atoi(e) != 0; }();
  return k128 ? gemm128_bf16(g, epi, s) : gemm256_bf16(g, epi, 4, s);
}

// spans for a problem of `tiles` 256 x 256 tiles and nk K-tiles: fill the CUs once, keep >= 16 K-tiles per span, fit the
// workspace; 1 = do not split
template <typename T>
int pick_spans(const GemmArgs<T>& g, int tiles, int nk) {
  static const bool enabled = [] {
    const char* e = getenv("GPT_GEMM_SPLITK");   // 0 disables (A/B measurements)
    return !e || atoi(e) != 0;
  }();
  const SplitWs& ws = split_ws();
  if (!enabled || !ws.ptr || sizeof(T) != 2 || g.N % 8 != 0) return 1;
  int sp = n_cus() / tiles;
  // the partial tiles cost (spans + 1) x M x N x 4 bytes of traffic and a second launch: measured against the single-pass
  // kernels the split wins from ~16 K-tiles (1024 of K) per span up (K = 6144: 70.7 -> 57.3 us at M = 1664, 129 -> 90 us
  // at M = 3328) and loses below (K = 1536, 6 spans of 4: 24.5 -> 38.8 us)
  if (sp > nk / 16) sp = nk / 16;
  const size_t slab = (size_t)g.M * g.N * sizeof(float);
  if ((size_t)sp * slab > ws.bytes) sp = (int)(ws.bytes / slab);
  if (sp < 2) return 1;
  const int per = (nk + sp - 1) / sp;              // K-tiles per span; drop the spans this rounding leaves empty
  return (nk + per - 1) / per;
}

int n_cus() {
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  return n_cu;
}

// The row tail of a 256-row launch (plan_rows256) is a sliver: at B = 320 fc1's is 128 rows x 6144 columns = 48 tiles of the
// 128 x 128 ring kernel, one workgroup on 48 of the 256 CUs walking all of K alone (latency-bound: 21 us for 2.4 GFLOP).
// With a workspace its K range is cut into spans of >= 12 ring K-tiles (384 of K) so that the tail covers the chip:
// pass 1 stores fp32 partial tiles, pass 2 (splitk_reduce_kernel) sums them and applies the epilogue.
template <typename T>
int pick_tail_spans(const GemmArgs<T>& g) {
  static const bool enabled = [] {
    const char* e = getenv("GPT_GEMM_TAILSPLIT");   // 0 disables (A/B measurements)
    return !e || atoi(e) != 0;
  }();
  const SplitWs& ws = split_ws();
  if (!enabled || !ws.ptr || sizeof(T) != 2 || g.N % 8 != 0) return 1;
  const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128), nk = g.K / (ROWB / (int)sizeof(T));
  if (tiles * 2 > n_cus()) return 1;                 // the tail already gives every CU a workgroup
  int sp = 2 * n_cus() / tiles;
  if (sp > nk / 12) sp = nk / 12;
  const size_t slab = (size_t)g.M * g.N * sizeof(float);
  if ((size_t)sp * slab > ws.bytes) sp = (int)(ws.bytes / slab);
  if (sp < 2) return 1;
  const int per = (nk + sp - 1) / sp;
  return (nk + per - 1) / per;
}

// tail of a row split: tiles of the 128 x 256 kernel, when they make one reasonably filled round of it (see dispatch)
template <typename T>
bool tail_fits128(int tail_m, int N) {
  static const bool tail128 = [] { const char* e = getenv("GPT_GEMM_TAIL128"); return !e || atoi(e) != 0; }();
  const int t128 = ((tail_m + 127) / 128) * ((N + 255) / 256);
  return sizeof(T) == 2 && tail128 && t128 >= n_cus() * 3 / 8 && t128 <= n_cus();
}

// cost in units of one full round of 256 x 256 tiles; *m_main = rows given to the 256 x 256 kernel (the rest: the 128 x 256
// kernel where the tail is one round of it - measured 0.58 of a 256-row round at K = 1536 -, else the ring kernels)
template <typename T>
double plan_rows256(const GemmArgs<T>& g, int* m_main) {
  static const bool enabled = [] {
    const char* e = getenv("GPT_GEMM_SPLIT");  // 0 disables (experiments)
    return !e || atoi(e) != 0;
  }();
  const int n_cu = n_cus();
  const int tn = (g.N + 255) / 256, tm = (g.M + 255) / 256, tiles = tm * tn;
  const int rounds = tiles / n_cu, rem = tiles % n_cu;
  *m_main = g.M;
  const double whole = (tiles + n_cu - 1) / n_cu;
  if (!enabled || rounds == 0 || rem == 0) return whole;
  const int main_tiles_m = rounds * n_cu / tn;
  const int mm = main_tiles_m * 256;
  if (mm <= 0 || mm >= g.M) return whole;
  double cost_split;
  if (tail_fits128<T>(g.M - mm, g.N)) {
    cost_split = rounds + 0.70 + 0.04;   // 0.58 measured in isolation; priced so that 224-row tiles keep qkv at M = 8320 (equal time, one launch)
  } else {
    if (rem * 4 > n_cu) return whole;
    const int tail128 = ((g.M - mm + 127) / 128) * ((g.N + 127) / 128);
    cost_split = rounds + 0.5 * ((tail128 + 2 * n_cu - 1) / (2 * n_cu)) + 0.08;
  }
  if (cost_split < whole) { *m_main = mm; return cost_split; }
  return whole;
}

// 224-row tiles (bf16): whole rounds of 7/8 the main-loop work each; prologue, epilogue and the unused A rows they stage
// do not shrink (measured: fc1 in 4 such rounds 153 us against 3 rounds + tail 148 us, i.e. 0.90 per round)
template <typename T>
double cost_rows224(const GemmArgs<T>& g) {
  static const bool enabled = [] {
    const char* e = getenv("GPT_GEMM_224");    // 0 disables (experiments)
    return !e || atoi(e) != 0;
  }();
  if (!enabled || sizeof(T) != 2) return 1e30;
  const int n_cu = n_cus();
  const int tiles = ((g.M + 223) / 224) * ((g.N + 255) / 256);
  const int rounds = (tiles + n_cu - 1) / n_cu;
  // whole-round counting holds for a few rounds only: over more of them the CUs drift apart and a partly filled last
  // round costs about its share (CIFAR shapes, M = 33280: 5 rounds of 224-row tiles measured 1.5 % slower than 4 + tail)
  if (rounds > 3) return 1e30;
  return rounds * 0.90 + 0.02;
}

// 192-row tiles (bf16): ONE round only, for small problems whose 224- / 256-row tiles leave a quarter of the CUs idle
// (B = 64 sampling, fc1: M = 1664, N = 6144 -> 216 tiles of 3/4 the main-loop work instead of 192 of 7/8)
template <typename T>
double cost_rows192(const GemmArgs<T>& g) {
  static const int max_rounds = [] {
    const char* e = getenv("GPT_GEMM_192");    // 0 disables, n = at most n rounds (A/B measurements)
    return e ? atoi(e) : 1;
  }();
  if (max_rounds <= 0 || sizeof(T) != 2) return 1e30;
  const int tiles = ((g.M + 191) / 192) * ((g.N + 255) / 256);
  const int rounds = (tiles + n_cus() - 1) / n_cus();
  if (rounds > max_rounds) return 1e30;
  return rounds * 0.80 + 0.02;
}

template <typename T>
int dispatch(const GemmArgs<T>& g, int epi, hipStream_t s) {
  const int tiles256 = ((g.M + 255) / 256) * ((g.N + 255) / 256);
  const bool k_ok = g.K % (128 / (int)sizeof(T)) == 0;   // the 256 x 256 kernel moves 128 bytes of K per row and K-tile
  // (column sums come out of the single-pass epilogues only: no K spans for a problem that wants them)
  if (k_ok && tile_mode() == 0 && tiles256 < 128 && tiles256 * 2 <= n_cus() && !g.colsum) {
    const int sp = pick_spans(g, tiles256, g.K / (128 / (int)sizeof(T)));
    if (sp > 1) {
      if constexpr (sizeof(T) == 2) {
        GemmArgs<T> part = g;                       // pass 1: raw fp32 partial tiles, [sp][M][N]
        part.bias = nullptr; part.aux = nullptr; part.C = split_ws().ptr; part.ldc = g.N; part.ksplit = sp;
        const int rc = gemm256_bf16(part, GPT_EPI_STORE_F32, 8, s);
        return rc != 0 ? rc : dispatch_reduce<T>(g, epi, split_ws().ptr, sp, s);
      }
    }
  }
  if constexpr (sizeof(T) == 2) {
    // 128-row tiles of the 256-column kernel (half the MFMAs per barrier interval, but 8 waves per CU) where they put
    // every tile on a CU of its own and 256-row tiles would leave half the CUs idle: M = 1664, N = 4608 39.9 -> 32.8 us
    static const bool rt4 = [] { const char* e = getenv("GPT_GEMM_RT4"); return !e || atoi(e) != 0; }();
    const int tiles128 = ((g.M + 127) / 128) * ((g.N + 255) / 256);
    if (rt4 && k_ok && tile_mode() == 0 && tiles256 < 128 && tiles128 > 128 && tiles128 <= n_cus()) return half_height_bf16(g, epi, s);
  }
  if (k_ok && (tile_mode() == 3 || (tile_mode() == 0 && tiles256 >= 128))) {
    int m_main = g.M;
    int rt = 8;
    if (tile_mode() == 0) {
      const double c8 = plan_rows256(g, &m_main);
      double best = c8;
      if (cost_rows224(g) < best - 0.01) { rt = 7; m_main = g.M; best = cost_rows224(g); }
      if (cost_rows192(g) < best - 0.01) { rt = 6; m_main = g.M; }
    }
    GemmArgs<T> head = g;
    head.M = m_main;
    int rc;
    if constexpr (sizeof(T) == 2) rc = gemm256_bf16(head, epi, rt, s);
    else rc = gemm256_f32(head, epi, s);
    if (rc != 0 || m_main == g.M) return rc;
    const bool c32 = epi == GPT_EPI_RESADD || epi == GPT_EPI_STORE_F32 || epi == GPT_EPI_ACCUM_F32;
    const bool aux32 = epi == GPT_EPI_RESADD;
    GemmArgs<T> tail = g;
    tail.M = g.M - m_main;
    tail.A = g.A + (int64_t)m_main * g.lda;
    tail.C = (char*)g.C + (int64_t)m_main * g.ldc * (c32 ? 4 : sizeof(T));
    if (g.aux) tail.aux = (char*)g.aux + (int64_t)m_main * g.ldaux * (aux32 ? 4 : sizeof(T));
    if constexpr (sizeof(T) == 2) {
      // a tail of several hundred rows (M = 3328, N = 6144: 768 rows) is one well-filled round of the 128 x 256 kernel
      // (144 tiles) instead of 288 ring-kernel tiles: 35 -> 27 us
      if (tail_fits128<T>(tail.M, tail.N)) return half_height_bf16(tail, epi, s);
      const int sp = g.colsum ? 1 : pick_tail_spans(tail);
      if (sp > 1) {
        GemmArgs<T> part = tail;                    // pass 1: raw fp32 partial tiles of the tail rows, [sp][tail.M][N]
        part.bias = nullptr; part.aux = nullptr; part.C = split_ws().ptr; part.ldc = g.N; part.ksplit = sp;
        rc = launch_shape<T, GPT_EPI_STORE_F32, 2, 4>(part, s);
        return rc != 0 ? rc : dispatch_reduce<T>(tail, epi, split_ws().ptr, sp, s);
      }
    }
    return dispatch_ring<T>(tail, epi, s);
  }
  return dispatch_ring<T>(g, epi, s);
}

template <typename T>
int dispatch_ring(const GemmArgs<T>& g, int epi, hipStream_t s) {
  switch (epi) {
    case GPT_EPI_STORE: return launch<T, GPT_EPI_STORE>(g, s);
    case GPT_EPI_GELU: return launch<T, GPT_EPI_GELU>(g, s);
    case GPT_EPI_RESADD: return launch<T, GPT_EPI_RESADD>(g, s);
    case GPT_EPI_STORE_F32: return launch<T, GPT_EPI_STORE_F32>(g, s);
    case GPT_EPI_GELU_AUX: return launch<T, GPT_EPI_GELU_AUX>(g, s);
    case GPT_EPI_DGELU: return launch<T, GPT_EPI_DGELU>(g, s);
    case GPT_EPI_ACCUM_F32: return launch<T, GPT_EPI_ACCUM_F32>(g, s);
  }
  gpt_set_error("gpt_gemm_nt: unknown epilogue %d", epi);
  return -1;
}

}  // namespace

extern "C" int gpt_gemm_set_workspace(void* ptr, size_t bytes) {
  GPT_REQUIRE((ptr == nullptr) == (bytes == 0) && ((uintptr_t)ptr % 16) == 0, "gpt_gemm_set_workspace: need a 16-byte aligned buffer (or NULL, 0)");
  split_ws() = SplitWs{(float*)ptr, bytes};
  return 0;
}

// spans for the fused residual GEMM + LayerNorm: pass 2 replaces a LayerNorm launch that would run anyway, so shorter spans
// pay than in pick_spans (>= 8 K-tiles of 64 instead of 16)
static int pick_spans_ln(int M, int N, int K, int* rt) {
  static const bool enabled = [] {
    const char* e = getenv("GPT_GEMM_FUSELN");   // 0 disables (A/B measurements)
    return !e || atoi(e) != 0;
  }();
  const SplitWs& ws = split_ws();
  *rt = 8;
  if (!enabled || !ws.ptr || K % 64 != 0 || tile_mode() != 0) return 1;
  const int nk = K / 64, tn = (N + 255) / 256;
  const int tiles256 = ((M + 255) / 256) * tn, tiles128 = ((M + 127) / 128) * tn;
  if (tiles256 * 2 > n_cus()) return 1;
  // long K: full-height tiles in spans of >= 16 K-tiles (the rule of pick_spans).  Short K (the attention projection: 24
  // K-tiles): half-height tiles while they leave room for two spans, spans of >= 8 K-tiles.
  int sp = n_cus() / tiles256;
  if (sp > nk / 16) sp = nk / 16;
  // long K, many spans: half-height tiles need half the spans to fill the chip (fc2 at M = 1664: 78 tiles x 3 spans instead of
  // 42 x 6), i.e. half the fp32 partial tiles for pass 2 to read: B = 64 sampling 1.958 -> 1.892 ms/step
  static const bool prefer128 = [] { const char* e = getenv("GPT_LN_PREFER128"); return !e || atoi(e) != 0; }();   // 0: A/B
  if (prefer128 && sp >= 4 && tiles128 * 2 <= n_cus()) {
    int sp128 = n_cus() / tiles128;
    if (sp128 > nk / 16) sp128 = nk / 16;
    if (sp128 >= 2 && sp128 * 2 <= sp) {
      *rt = 4;
      sp = sp128;
    }
  }
  // r03: very few rows (the pruned last Block at B = 64: M = 704, K = 6144 -> 18 tiles x 6 spans = 108 workgroups): the same span
  // count on half-height tiles fills twice the CUs at the same partial-tile traffic
  static const bool fill128 = [] { const char* e = getenv("GPT_LN_FILL128"); return !e || atoi(e) != 0; }();   // 0: A/B
  if (fill128 && *rt == 8 && sp >= 2 && tiles256 * sp * 2 <= n_cus() && tiles128 * sp <= n_cus()) *rt = 4;
  if (sp < 2) {
    // r03: where half-height tiles already give more than half the CUs a tile (M = 3328 / 3520, N = 1536: 156 / 168 of them) the
    // single pass on gemm128.hip + the LayerNorm kernel beats three spans of full-height tiles + the fused pass: 36 against 45 - 47 us
    // (tools/resadd_ln_bench.py; M = 1664: 28 either way, so the span route stays there)
    if (tiles128 * 2 > n_cus()) return 1;
    *rt = 4;
    sp = n_cus() / tiles128;
    if (sp > nk / 8) sp = nk / 8;
  }
  const size_t slab = (size_t)M * N * sizeof(float);
  if ((size_t)sp * slab > ws.bytes) sp = (int)(ws.bytes / slab);
  if (sp < 2) return 1;
  const int per = (nk + sp - 1) / sp;
  return (nk + per - 1) / per;
}

extern "C" int gpt_gemm_resadd_layernorm(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                         const float* resid, int64_t ldr, float* x, int64_t ldx, const float* gamma,
                                         const float* beta, void* y, float* mean, float* rstd, int M, int N, int K, float eps,
                                         void* stream) {
  GPT_REQUIRE(A && W && resid && x && gamma && beta && y, "gpt_gemm_resadd_layernorm: null operand");
  GPT_REQUIRE(ldx == N, "gpt_gemm_resadd_layernorm: the residual stream must be dense (ldx == N)");
  GPT_REQUIRE(ldr % 4 == 0 && ((uintptr_t)resid % 16) == 0 && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0 &&
                  ((uintptr_t)y % 8) == 0,
              "gpt_gemm_resadd_layernorm: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int rt = 8;
  const int nv = N % 256 == 0 ? N / 256 : 0;
  const bool nv_ok = nv == 3 || nv == 4 || nv == 6 || nv == 8 || nv == 12;
  const int sp = (dtype == GPT_BF16 && nv_ok && M > 0 && K > 0 && lda >= K && ldw >= K && ((uintptr_t)A % 16) == 0 &&
                  ((uintptr_t)W % 16) == 0 && (lda * 2) % 16 == 0 && (ldw * 2) % 16 == 0 && ((uintptr_t)bias % 16) == 0)
                     ? pick_spans_ln(M, N, K, &rt)
                     : 1;
  if (sp < 2) {
    // large batches (and fp32): the single-pass GEMM with its residual epilogue, then the LayerNorm kernel
    const int rc = gpt_gemm_nt(dtype, A, lda, W, ldw, bias, x, ldx, resid == x ? nullptr : (void*)resid, ldr, M, N, K,
                               GPT_EPI_RESADD, stream);
    return rc != 0 ? rc : gpt_layernorm(x, gamma, beta, nullptr, 0, y, dtype, mean, rstd, M, N, eps, stream);
  }
  GemmArgs<bf16_t> part{(const bf16_t*)A, (const bf16_t*)W, nullptr, split_ws().ptr, nullptr, lda, ldw, N, 0, M, N, K};
  part.ksplit = sp;
  const int rc = rt == 4 ? half_height_bf16(part, GPT_EPI_STORE_F32, s) : gemm256_bf16(part, GPT_EPI_STORE_F32, rt, s);
  if (rc != 0) return rc;
  const dim3 grid((M + 3) / 4), block(256);
  // two waves per row at the small row counts (three measure the same, six worse: B = 64 sampling step 1.897 -> 1.876 / 1.871 / 1.894 ms)
  static const bool splitrow = [] { const char* e = getenv("GPT_LN_SPLITROW"); return !e || atoi(e) != 0; }();   // 0: a wave per row (A/B)
  if (splitrow && (nv == 6 || nv == 8 || nv == 12) && M * 2 <= 32 * n_cus()) {
#define RLN2(NV, WPR) hipLaunchKernelGGL((splitk_reduce_ln2_kernel<NV, WPR, bf16_t>), dim3(M), dim3(64 * WPR), 0, s, split_ws().ptr, sp, bias, \
                                         resid, ldr, x, ldx, gamma, beta, (bf16_t*)y, mean, rstd, M, eps)
    // n_embd 3072 (configs[4], M = 1768): four waves per row, three 16-byte chunks per lane and source - a wave per row ran this pass
    // at 1.8 TB/s (65.8 us for 120 MB, 7 waves per CU)
    if (nv == 6) RLN2(6, 2);
    else if (nv == 8) RLN2(8, 2);
    else RLN2(12, 4);
#undef RLN2
    return gpt_launch_status("gpt_gemm_resadd_layernorm");
  }
#define RLN(NV) hipLaunchKernelGGL((splitk_reduce_ln_kernel<NV, bf16_t>), grid, block, 0, s, split_ws().ptr, sp, bias, resid, ldr, x, \
                                   ldx, gamma, beta, (bf16_t*)y, mean, rstd, M, eps)
  switch (nv) {
    case 3: RLN(3); break;
    case 4: RLN(4); break;
    case 6: RLN(6); break;
    case 8: RLN(8); break;
    default: RLN(12); break;
  }
#undef RLN
  return gpt_launch_status("gpt_gemm_resadd_layernorm");
}

extern "C" int gpt_gemm_prof_enable(int on) {
  if (on && !g_prof.made) {
    for (int i = 0; i < GemmProf::CAP; ++i)
      for (int j = 0; j < 2; ++j)
        if (hipEventCreate(&g_prof.ev[i][j]) != hipSuccess) {
          gpt_set_error("gpt_gemm_prof_enable: hipEventCreate failed");
          return -1;
        }
    g_prof.made = true;
  }
  g_prof.on = on != 0;
  if (on) g_prof.n = 0;
  return 0;
}

extern "C" int gpt_gemm_prof_collect(double* total_ms, double* total_flops, int* launches) {
  GPT_REQUIRE(total_ms && total_flops && launches, "gpt_gemm_prof_collect: null output");
  double ms = 0, fl = 0;
  for (int i = 0; i < g_prof.n; ++i) {
    if (hipEventSynchronize(g_prof.ev[i][1]) != hipSuccess) { gpt_set_error("gpt_gemm_prof_collect: sync failed"); return -1; }
    float t = 0;
    if (hipEventElapsedTime(&t, g_prof.ev[i][0], g_prof.ev[i][1]) != hipSuccess) { gpt_set_error("gpt_gemm_prof_collect: elapsed failed"); return -1; }
    ms += t;
    fl += g_prof.flops[i];
  }
  *total_ms = ms; *total_flops = fl; *launches = g_prof.n;
  return 0;
}

extern "C" int gpt_gemm_nt_static(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                  void* C, int64_t ldc, void* aux, int64_t ldaux, int M, int N, int K, int epilogue,
                                  void* stream) {
  // two K spans of partial tiles must fit, else the scratch could never be used for this problem: such problems belong to
  // gpt_gemm_nt (at that size the single-pass kernels fill the chip anyway)
  GPT_REQUIRE(M > 0 && N > 0 && (size_t)M * (size_t)N * sizeof(float) * 2 <= STATIC_SCRATCH_BYTES,
              "gpt_gemm_nt_static: M x N = %d x %d exceeds the static scratch (%zu MiB holds two fp32 partial copies of at most "
              "%zu output elements); use gpt_gemm_nt", M, N, STATIC_SCRATCH_BYTES >> 20, STATIC_SCRATCH_BYTES / 8);
  SplitWs ws;
  CHECK_RC(static_scratch_ptr(&ws.ptr));
  ws.bytes = STATIC_SCRATCH_BYTES;
  g_ws_override = &ws;          // pick_spans / dispatch see this device's static scratch instead of a registered workspace
  const int rc = gpt_gemm_nt(dtype, A, lda, W, ldw, bias, C, ldc, aux, ldaux, M, N, K, epilogue, stream);
  g_ws_override = nullptr;
  return rc;
}

extern "C" int gpt_gemm_nt(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                           void* C, int64_t ldc, void* aux, int64_t ldaux, int M, int N, int K, int epilogue,
                           void* stream) {
  return gpt_gemm_nt_colsum(dtype, A, lda, W, ldw, bias, C, ldc, aux, ldaux, M, N, K, epilogue, nullptr, stream);
}

int gpt_gemm_nt_colsum(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                       void* aux, int64_t ldaux, int M, int N, int K, int epilogue, float* colsum, void* stream) {
  GPT_REQUIRE(!colsum || epilogue == GPT_EPI_DGELU, "gpt_gemm_nt: column sums exist for the DGELU epilogue only");
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_gemm_nt: bad dtype %d", dtype);
  GPT_REQUIRE(M > 0 && N > 0 && K > 0, "gpt_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  const int esz = dtype == GPT_BF16 ? 2 : 4;
  const int bk = ROWB / esz;
  GPT_REQUIRE(K % bk == 0, "gpt_gemm_nt: K=%d must be a multiple of %d", K, bk);
  GPT_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gpt_gemm_nt: leading dimension too small");
  GPT_REQUIRE((lda * esz) % 16 == 0 && (ldw * esz) % 16 == 0, "gpt_gemm_nt: rows must be 16-byte aligned");
  GPT_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "gpt_gemm_nt: operands must be 16-byte aligned");
  GPT_REQUIRE(A && W && C, "gpt_gemm_nt: null operand");
  GPT_REQUIRE(N % 8 == 0 && ldc % 4 == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)bias % 16) == 0,
              "gpt_gemm_nt: N must be a multiple of 8, ldc of 4, C and bias 16-byte aligned (vector epilogue)");
  GPT_REQUIRE(!aux || (ldaux % 4 == 0 && ((uintptr_t)aux % 16) == 0), "gpt_gemm_nt: aux must be 16-byte aligned, ldaux %% 4 == 0");
  if (epilogue == GPT_EPI_GELU_AUX || epilogue == GPT_EPI_DGELU)
    GPT_REQUIRE(aux && ldaux >= N, "gpt_gemm_nt: epilogue %d needs aux with ldaux >= N", epilogue);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GPT_BF16) {
    GemmArgs<bf16_t> g{(const bf16_t*)A, (const bf16_t*)W, bias, C, aux, lda, ldw, ldc, ldaux, M, N, K};
    g.colsum = colsum;
    return dispatch<bf16_t>(g, epilogue, s);
  }
  GemmArgs<float> g{(const float*)A, (const float*)W, bias, C, aux, lda, ldw, ldc, ldaux, M, N, K};
  g.colsum = colsum;
  return dispatch<float>(g, epilogue, s);
}
