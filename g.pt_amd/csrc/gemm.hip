// MFMA GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)  for gfx950, fp32 (exact) and bf16.
//
// Replaces the nn.Linear contractions of Gpt/models/transformer.py:38-40,50,63-65 (99.3 % of the
// denoiser FLOPs).  Both operands are K-contiguous, which is the
// natural MFMA operand order, so nn.Linear's (out,in) weight layout is consumed as stored.
//
// Tile: 128 x 128 per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 each = 2 x 2 MFMA 32x32 tiles),
// K-step = 128 bytes of K per row (64 bf16 / 32 fp32), so the LDS image and the staging code are the
// same for both dtypes.  Operands are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave-instruction, lane-linear destination) into a double buffer; the bank-conflict swizzle is applied
// on the per-lane SOURCE address and again on the ds_read_b128 address (phys chunk = chunk ^ ((row>>1)&7),
// conflict-free for the 16-lane groups of ds_read_b128 on 128-byte rows).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;           // tile rows / cols, bytes of K per staged row
constexpr int TILE_BYTES = BM * ROWB;                   // 16 KiB per operand per buffer
constexpr int NTHREADS = 256;

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
};

// stage one 128-row x 128-byte operand tile: 16 pieces of 1 KiB, 4 per wave
template <typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ base, int64_t ld, int row0, int nrows_total, int k0,
                                           char* lds_tile, int wid, int lane) {
  constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wid * 4 + i;
    const int rr = piece * 8 + (lane >> 3);
    const int pc = lane & 7;
    const int lc = pc ^ ((rr >> 1) & 7);
    int grow = row0 + rr;
    grow = grow < nrows_total ? grow : nrows_total - 1;  // clamp: rows past the edge are computed and dropped
    const T* src = base + (int64_t)grow * ld + k0 + lc * EPC;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ u32x4 read_frag(const char* lds_tile, int row, int lc) {
  const int off = row * ROWB + ((lc ^ ((row >> 1) & 7)) << 4);
  return *reinterpret_cast<const u32x4*>(lds_tile + off);
}

template <typename T, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(GemmArgs<T> g) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][A|W]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;

  // XCD-aware, bijective remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
  // contiguous run of tiles so neighbours re-use the same A panel / W panels from that L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tm = swz / tiles_n, tn = swz % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  constexpr int BK = ROWB / sizeof(T);
  const int nk = g.K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  stage_tile<T>(g.A, g.lda, m0, g.M, 0, smem, wid, lane);
  stage_tile<T>(g.W, g.ldw, n0, g.N, 0, smem + TILE_BYTES, wid, lane);
  __syncthreads();

  const int fr = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    char* bufA = smem + cur * 2 * TILE_BYTES;
    char* bufW = bufA + TILE_BYTES;
    if (kt + 1 < nk) {
      char* nA = smem + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile<T>(g.A, g.lda, m0, g.M, (kt + 1) * BK, nA, wid, lane);
      stage_tile<T>(g.W, g.ldw, n0, g.N, (kt + 1) * BK, nA + TILE_BYTES, wid, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = read_frag(bufA, wr * 64 + i * 32 + fr, 2 * ks + fh);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = read_frag(bufW, wc * 64 + j * 32 + fr, 2 * ks + fh);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (sizeof(T) == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
          } else {
            // 16 bytes = 4 fp32 k-values per lane half; element e of both halves forms one K=2 step
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i][e]), __uint_as_float(b[j][e]),
                                                               acc[i][j], 0, 0, 0);
          }
        }
    }
    __syncthreads();
  }

  // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wc * 64 + j * 32 + fr;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (m >= g.M) continue;
        float v = acc[i][j][e] + bv;
        const int64_t ci = (int64_t)m * g.ldc + n;
        if constexpr (EPI == GPT_EPI_STORE) {
          elem<T>::store((T*)g.C + ci, v);
        } else if constexpr (EPI == GPT_EPI_GELU) {
          elem<T>::store((T*)g.C + ci, gelu_erf(v));
        } else if constexpr (EPI == GPT_EPI_RESADD) {
          const float res = g.aux ? ((const float*)g.aux)[(int64_t)m * g.ldaux + n] : ((const float*)g.C)[ci];
          ((float*)g.C)[ci] = res + v;
        } else if constexpr (EPI == GPT_EPI_STORE_F32) {
          ((float*)g.C)[ci] = v;
        } else if constexpr (EPI == GPT_EPI_GELU_AUX) {
          elem<T>::store((T*)g.aux + (int64_t)m * g.ldaux + n, v);
          elem<T>::store((T*)g.C + ci, gelu_erf(v));
        } else if constexpr (EPI == GPT_EPI_DGELU) {
          const float pre = elem<T>::load((const T*)g.aux + (int64_t)m * g.ldaux + n);
          elem<T>::store((T*)g.C + ci, v * gelu_erf_grad(pre));
        } else if constexpr (EPI == GPT_EPI_ACCUM_F32) {
          float* c = (float*)g.C + ci;
          *c = *c + v;
        }
      }
    }
  }
}

// Optional per-launch timing with HIP events recorded on the launch stream (bench.py's roofline pass).
struct GemmProf {
  bool on = false;
  int n = 0;
  static constexpr int CAP = 4096;
  hipEvent_t ev[CAP][2];
  double flops[CAP];
  bool made = false;
} g_prof;

template <typename T, int EPI>
int launch(const GemmArgs<T>& g, hipStream_t s) {
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  const bool rec = g_prof.on && g_prof.n < GemmProf::CAP;
  if (rec) (void)hipEventRecord(g_prof.ev[g_prof.n][0], s);
  hipLaunchKernelGGL((gemm_nt_kernel<T, EPI>), dim3(tiles), dim3(NTHREADS), 0, s, g);
  if (rec) {
    (void)hipEventRecord(g_prof.ev[g_prof.n][1], s);
    g_prof.flops[g_prof.n++] = 2.0 * g.M * g.N * g.K;
  }
  return gpt_launch_status("gpt_gemm_nt");
}

template <typename T>
int dispatch(const GemmArgs<T>& g, int epi, hipStream_t s) {
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

extern "C" int gpt_gemm_nt(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                           void* C, int64_t ldc, void* aux, int64_t ldaux, int M, int N, int K, int epilogue,
                           void* stream) {
  GPT_REQUIRE(dtype == GPT_F32 || dtype == GPT_BF16, "gpt_gemm_nt: bad dtype %d", dtype);
  GPT_REQUIRE(M > 0 && N > 0 && K > 0, "gpt_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  const int esz = dtype == GPT_BF16 ? 2 : 4;
  const int bk = ROWB / esz;
  GPT_REQUIRE(K % bk == 0, "gpt_gemm_nt: K=%d must be a multiple of %d", K, bk);
  GPT_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gpt_gemm_nt: leading dimension too small");
  GPT_REQUIRE((lda * esz) % 16 == 0 && (ldw * esz) % 16 == 0, "gpt_gemm_nt: rows must be 16-byte aligned");
  GPT_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "gpt_gemm_nt: operands must be 16-byte aligned");
  GPT_REQUIRE(A && W && C, "gpt_gemm_nt: null operand");
  if (epilogue == GPT_EPI_GELU_AUX || epilogue == GPT_EPI_DGELU)
    GPT_REQUIRE(aux && ldaux >= N, "gpt_gemm_nt: epilogue %d needs aux with ldaux >= N", epilogue);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == GPT_BF16) {
    GemmArgs<bf16_t> g{(const bf16_t*)A, (const bf16_t*)W, bias, C, aux, lda, ldw, ldc, ldaux, M, N, K};
    return dispatch<bf16_t>(g, epilogue, s);
  }
  GemmArgs<float> g{(const float*)A, (const float*)W, bias, C, aux, lda, ldw, ldc, ldaux, M, N, K};
  return dispatch<float>(g, epilogue, s);
}
