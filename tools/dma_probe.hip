// Probe: how fast can one CU pull GEMM operand K-tiles out of L2 / the fabric, with nothing else in the way?
// Same access pattern as gemm256_kernel (g.pt_amd/csrc/gemm256.hip): per K-tile 64 pieces of 1 KiB (8 rows x 128 bytes, one
// 16-byte chunk per lane) for a 256 x 256 tile, issued by the workgroup's waves; no MFMA, no fragment reads, no barriers.
// r02 result (MI355X, 8192^3 pattern): 27 TB/s chip-wide = 47 B/cycle per CU at any depth (1 - 7 K-tiles in flight, 8 or 4 issuing
// waves): twice what the GEMM's main loop draws (24 B/cycle per CU), so the feed is not its limit.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_probe.hip -o tools/bin/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));    \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WAVES = waves of the 8 that issue (each issues 64 / WAVES pieces per K-tile); DEPTH = K-tiles kept in flight
// SEG = 0: pieces of 8 rows x 128 bytes (whole lines);  1: pieces of 16 rows x 64 bytes, the two halves of a line 16 pieces apart
// (the fill shape of a ring of 32-wide k-slots: gemm256w.hip);  2: the same half-line pieces, both halves of a line by consecutive pieces
template <int WAVES, int DEPTH, int SEG = 0>
__global__ __launch_bounds__(512, 2) void probe(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, int64_t lda,
                                                int64_t ldw, int M, int N, int nk, unsigned long long* cyc, unsigned int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PPW = 64 / WAVES;   // pieces per wave and K-tile
  static_assert(PPW * DEPTH <= 56, "vmcnt range");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tiles_n = N / 256, tiles_m = M / 256;
  constexpr int GROUP_N = 4;
  const int gb = tile / (tiles_m * GROUP_N), grem = tile - gb * (tiles_m * GROUP_N);
  const int gw = min(GROUP_N, tiles_n - gb * GROUP_N);
  const int m0 = (grem / gw) * 256, n0 = (gb * GROUP_N + grem % gw) * 256;
  unsigned int acc = 0;
  if (wid < WAVES) {
    // piece j of this wave: global piece index gp = wid * PPW + j of 64: gp < 32 -> A rows 8 gp .., else W rows 8 (gp - 32) ..
    const unsigned short* src[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int gp = wid * PPW + j;
      int row = (gp & 31) * 8 + (lane >> 3);
      int lc = (lane & 7) ^ ((row >> 1) & 7);
      if (SEG) {
        const int idx = gp & 31, half = SEG == 1 ? idx >> 4 : idx & 1, rg = SEG == 1 ? idx & 15 : idx >> 1;
        row = rg * 16 + (lane >> 2);
        lc = ((lane & 3) ^ ((row >> 3) & 3)) + 4 * half;
      }
      src[j] = gp < 32 ? A + (int64_t)(m0 + row) * lda + lc * 8 : W + (int64_t)(n0 + row) * ldw + lc * 8;
    }
    for (int kt = 0; kt < nk; kt += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
          const unsigned short* p = src[j] + (int64_t)(kt + d) * 64;
          char* dst = smem + ((kt + d) & 1) * 65536 + (wid * PPW + j) * 1024;
          __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)dst, 16, 0, 0);
        }
        wait_vm<PPW*(DEPTH - 1)>();   // rolling window: at most DEPTH K-tiles in flight
      }
    }
    wait_vm<0>();
  }
  __syncthreads();
  acc = *reinterpret_cast<unsigned int*>(smem + tid * 4);
  if (acc == 0x12345678u) sink[0] = acc;
  if (tid == 0) cyc[bid] = __builtin_amdgcn_s_memtime() - t0;
}

template <int WAVES, int DEPTH, int SEG = 0>
void run(const char* name, const unsigned short* A, const unsigned short* W, int M, int N, int K, unsigned long long* cyc, unsigned int* sink) {
  const int tiles = (M / 256) * (N / 256), nk = K / 64;
  CHECK(hipFuncSetAttribute((const void*)probe<WAVES, DEPTH, SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 5;
  for (int it = 0; it < 2; ++it) {
    if (it == 1) CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < (it ? reps : 2); ++i)
      hipLaunchKernelGGL((probe<WAVES, DEPTH, SEG>), dim3(tiles), dim3(512), 131072, 0, A, W, (int64_t)K, (int64_t)K, M, N, nk, cyc, sink);
  }
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(tiles);
  CHECK(hipMemcpy(h.data(), cyc, tiles * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double c = 0;
  for (auto v : h) c += (double)v;
  c /= tiles;
  const int nk_eff = (nk + DEPTH - 1) / DEPTH * DEPTH;
  const double us = ms * 1e3 / reps, bytes = (double)tiles * nk_eff * 65536.0;
  const double per_tile_bytes = (double)nk_eff * 65536.0;
  printf("%-34s M=%5d N=%5d K=%5d  %8.1f us  %6.2f TB/s chip  %6.1f B/cycle per workgroup (its own %9.0f cycles per tile)  mfma-equivalent %5.0f TF/s\n",
         name, M, N, K, us, bytes / us * 1e-6, per_tile_bytes / c, c, 2.0 * M * N * K / us * 1e-6);
  fflush(stdout);
}

int main() {
  const int MMAX = 8320, NMAX = 8192, KMAX = 8192;
  unsigned short *A, *W;
  unsigned long long* cyc;
  unsigned int* sink;
  CHECK(hipMalloc(&A, (size_t)MMAX * KMAX * 2));
  CHECK(hipMalloc(&W, (size_t)(NMAX + 8) * KMAX * 2));
  CHECK(hipMalloc(&cyc, 8192 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(A, 1, (size_t)MMAX * KMAX * 2));
  CHECK(hipMemset(W, 1, (size_t)(NMAX + 8) * KMAX * 2));
  const int shapes[4][3] = {{8192, 8192, 8192}, {8192, 4608, 1536}, {8192, 1536, 6144}, {1536, 4608, 1536}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    run<8, 1>("lds-dma 8 waves depth 1", A, W, M, N, K, cyc, sink);
    run<8, 2>("lds-dma 8 waves depth 2", A, W, M, N, K, cyc, sink);
    run<8, 4>("lds-dma 8 waves depth 4", A, W, M, N, K, cyc, sink);
    run<8, 7>("lds-dma 8 waves depth 7", A, W, M, N, K, cyc, sink);
    run<4, 2>("lds-dma 4 waves depth 2", A, W, M, N, K, cyc, sink);
    run<4, 3>("lds-dma 4 waves depth 3", A, W, M, N, K, cyc, sink);
    run<4, 3, 1>("half lines 16 apart, 4 waves d3", A, W, M, N, K, cyc, sink);
    run<4, 3, 2>("half lines adjacent, 4 waves d3", A, W, M, N, K, cyc, sink);
    run<8, 4, 1>("half lines 16 apart, 8 waves d4", A, W, M, N, K, cyc, sink);
  }
  return 0;
}
