// On-device permutation augmentation: integer gather maps (bit-exact) and the gather itself.
// Follows Gpt/data/augment.py:25-75 (random_permute_flat: one flat index built block by block in
// state_dict order, the same index applied to both checkpoints of a pair) with the task functions
// data_gen/train_mnist.py:28-50 (MLP), data_gen/train_cifar10.py:27-47 (CNN) and data_gen/train_rl.py:51-74 (policy MLP chain).
// HBM-bound integer/byte work: 8*D bytes written per index map; 4*D read + 4*D written (+8*D index) per
// gathered vector.  Hidden-unit permutations are drawn on the host (torch.randperm on the reference's
// generator) so the maps equal the reference's for the same seed.
#include "common.h"

namespace {

// MLP: fc1 (H, n_in) 'out' | fc1.b (H) 'out' | fc2 (n_out, H) 'in' | fc2.b (n_out) unchanged
__global__ __launch_bounds__(256) void index_mlp_kernel(const int32_t* __restrict__ perms, int64_t* __restrict__ index,
                                                        int n_in, int H, int n_out) {
  const int s = blockIdx.y;
  const int32_t* p = perms + (int64_t)s * H;
  const int o1 = H * n_in, o2 = o1 + H, o3 = o2 + n_out * H, D = o3 + n_out;
  int64_t* out = index + (int64_t)s * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) {
    int64_t v;
    if (i < o1) v = (int64_t)p[i / n_in] * n_in + i % n_in;
    else if (i < o2) v = o1 + p[i - o1];
    else if (i < o3) { const int j = i - o2; v = o2 + (j / H) * H + p[j % H]; }
    else v = i;
    out[i] = v;
  }
}

// CNN: conv1 (C1, c_in, k, k) 'out' | conv2 (C2, C1, k, k) 'both' = [:, p1][p2] | fc (n_out, C2) 'in' | fc.b unchanged
__global__ __launch_bounds__(256) void index_cnn_kernel(const int32_t* __restrict__ perms, int64_t* __restrict__ index,
                                                        int c_in, int C1, int C2, int kk, int n_out) {
  const int s = blockIdx.y;
  const int32_t* p1 = perms + (int64_t)s * (C1 + C2);
  const int32_t* p2 = p1 + C1;
  const int r1 = c_in * kk, r2 = C1 * kk;
  const int o1 = C1 * r1, o2 = o1 + C2 * r2, o3 = o2 + n_out * C2, D = o3 + n_out;
  int64_t* out = index + (int64_t)s * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) {
    int64_t v;
    if (i < o1) v = (int64_t)p1[i / r1] * r1 + i % r1;
    else if (i < o2) {
      const int j = i - o1, o = j / r2, rem = j % r2;
      v = o1 + (int64_t)p2[o] * r2 + p1[rem / kk] * kk + rem % kk;
    } else if (i < o3) { const int j = i - o2; v = o2 + (j / C2) * C2 + p2[j % C2]; }
    else v = i;
    out[i] = v;
  }
}

// Policy MLP chain (data_gen/train_rl.py:51-74, random_permute_actor on VisActorCritic, train_rl.py:95-121): `prefix` unpermuted
// leading scalars (log_std), then n_lin Linears (weight [dims[l+1], dims[l]] then bias [dims[l+1]]) in state_dict order.  Hidden
// layer l < n_lin - 1 has its own permutation p_l (drawn input layer first): weight rows and bias follow p_l ('out'), weight
// columns follow p_{l-1} ('in'; 'both' = [:, p_{l-1}][p_l] for the hidden layers); the output layer keeps its rows and bias.
constexpr int MAX_CHAIN = 8;
struct ChainArgs {
  int dims[MAX_CHAIN + 1];
  int n_lin, prefix, perm_len, D;
};
__global__ __launch_bounds__(256) void index_chain_kernel(const int32_t* __restrict__ perms, int64_t* __restrict__ index, ChainArgs c) {
  const int s = blockIdx.y;
  const int32_t* p = perms + (int64_t)s * c.perm_len;
  int64_t* out = index + (int64_t)s * c.D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.D; i += gridDim.x * blockDim.x) {
    int64_t v = i;
    int off = c.prefix, poff_in = 0, poff_out = 0;      // poff_out: where p_l starts; poff_in: where p_{l-1} starts
    if (i >= off) {
#pragma unroll 1
      for (int l = 0; l < c.n_lin; ++l) {
        const int n_in = c.dims[l], n_out = c.dims[l + 1];
        const bool perm_out = l < c.n_lin - 1, perm_in = l > 0;
        const int wsz = n_out * n_in;
        if (i < off + wsz) {
          const int j = i - off, r = j / n_in, col = j % n_in;
          v = off + (int64_t)(perm_out ? p[poff_out + r] : r) * n_in + (perm_in ? p[poff_in + col] : col);
          break;
        }
        off += wsz;
        if (i < off + n_out) {
          const int r = i - off;
          v = off + (perm_out ? p[poff_out + r] : r);
          break;
        }
        off += n_out;
        poff_in = poff_out;
        poff_out += n_out;
      }
    }
    out[i] = v;
  }
}

constexpr int MAX_NETS = 4;
struct GatherArgs {
  const float* src[MAX_NETS];
  float* dst[MAX_NETS];
  int n_nets;
};

__global__ __launch_bounds__(256) void gather_kernel(GatherArgs g, const int64_t* __restrict__ index, int D) {
  const int s = blockIdx.y;
  const int64_t* idx = index + (int64_t)s * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) {
    const int64_t j = idx[i];
#pragma unroll
    for (int n = 0; n < MAX_NETS; ++n)
      if (n < g.n_nets) g.dst[n][(int64_t)s * D + i] = g.src[n][(int64_t)s * D + j];
  }
}

// Training-pair fetch from a checkpoint store resident in HBM: row gather + parameter normalisation + permutation
// augmentation in one pass (ParameterDataset.__getitem__, Gpt/data/dataset_lmdb.py:218-225: database[...] -> normalize ->
// aug_fn on both checkpoints of the pair).  HBM-bound: 16 bytes per element and pair (+ 8 of index when augmenting).
__global__ __launch_bounds__(256) void pair_gather_kernel(const float* __restrict__ store, int64_t ld, const int64_t* __restrict__ row0,
                                                          const int64_t* __restrict__ row1, const int64_t* __restrict__ index,
                                                          float scale, float* __restrict__ out0, float* __restrict__ out1, int D) {
  const int s = blockIdx.y;
  const float* a = store + row0[s] * ld;
  const float* b = store + row1[s] * ld;
  const int64_t* idx = index ? index + (int64_t)s * D : nullptr;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D; i += gridDim.x * blockDim.x) {
    const int64_t j = idx ? idx[i] : i;
    out0[(int64_t)s * D + i] = a[j] * scale;
    out1[(int64_t)s * D + i] = b[j] * scale;
  }
}

}  // namespace

extern "C" int gpt_pair_gather(const float* store, int64_t ld, int64_t n_rows, const int64_t* row0, const int64_t* row1,
                               const int64_t* index, float scale, float* out0, float* out1, int n_samples, int D, void* stream) {
  GPT_REQUIRE(store && row0 && row1 && out0 && out1 && n_samples > 0 && D > 0 && ld >= D && n_rows > 0,
              "gpt_pair_gather: bad arguments (need ld >= D, n_rows > 0)");
  GPT_REQUIRE(n_samples <= 65535, "gpt_pair_gather: at most 65535 pairs per call");
  const dim3 grid((D + 255) / 256 > 16 ? 16 : (D + 255) / 256, n_samples);
  hipLaunchKernelGGL(pair_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, store, ld, row0, row1, index, scale, out0, out1, D);
  return gpt_launch_status("gpt_pair_gather");
}

extern "C" int gpt_perm_index_mlp(const int32_t* perms, int64_t* index, int n_samples, int n_in, int hidden, int n_out,
                                  void* stream) {
  GPT_REQUIRE(perms && index && n_samples > 0 && n_in > 0 && hidden > 0 && n_out > 0, "gpt_perm_index_mlp: bad arguments");
  const int D = hidden * n_in + hidden + n_out * hidden + n_out;
  const dim3 grid((D + 255) / 256 > 16 ? 16 : (D + 255) / 256, n_samples);
  hipLaunchKernelGGL(index_mlp_kernel, grid, dim3(256), 0, (hipStream_t)stream, perms, index, n_in, hidden, n_out);
  return gpt_launch_status("gpt_perm_index_mlp");
}

extern "C" int gpt_perm_index_cnn(const int32_t* perms, int64_t* index, int n_samples, int c_in, int c1, int c2, int ksz,
                                  int n_out, void* stream) {
  GPT_REQUIRE(perms && index && n_samples > 0 && c_in > 0 && c1 > 0 && c2 > 0 && ksz > 0 && n_out > 0,
              "gpt_perm_index_cnn: bad arguments");
  const int kk = ksz * ksz;
  const int D = c1 * c_in * kk + c2 * c1 * kk + n_out * c2 + n_out;
  const dim3 grid((D + 255) / 256 > 16 ? 16 : (D + 255) / 256, n_samples);
  hipLaunchKernelGGL(index_cnn_kernel, grid, dim3(256), 0, (hipStream_t)stream, perms, index, c_in, c1, c2, kk, n_out);
  return gpt_launch_status("gpt_perm_index_cnn");
}

extern "C" int gpt_perm_index_chain(const int32_t* perms, int64_t* index, int n_samples, int prefix, int n_dims, const int* host_dims,
                                    void* stream) {
  GPT_REQUIRE(perms && index && host_dims && n_samples > 0 && prefix >= 0, "gpt_perm_index_chain: bad arguments");
  GPT_REQUIRE(n_dims >= 3 && n_dims <= MAX_CHAIN + 1, "gpt_perm_index_chain: 2 to %d Linears (3 to %d widths)", MAX_CHAIN, MAX_CHAIN + 1);
  GPT_REQUIRE(n_samples <= 65535, "gpt_perm_index_chain: at most 65535 samples per call");
  ChainArgs c;
  c.n_lin = n_dims - 1;
  c.prefix = prefix;
  c.perm_len = 0;
  int64_t D = prefix;
  for (int l = 0; l < n_dims; ++l) {
    GPT_REQUIRE(host_dims[l] > 0, "gpt_perm_index_chain: width %d is not positive", l);
    c.dims[l] = host_dims[l];
    if (l > 0) D += (int64_t)host_dims[l] * host_dims[l - 1] + host_dims[l];
    if (l > 0 && l < n_dims - 1) c.perm_len += host_dims[l];
  }
  GPT_REQUIRE(D < (1ll << 31), "gpt_perm_index_chain: the flat vector has to stay below 2^31 elements");
  c.D = (int)D;
  const dim3 grid((c.D + 255) / 256 > 16 ? 16 : (c.D + 255) / 256, n_samples);
  hipLaunchKernelGGL(index_chain_kernel, grid, dim3(256), 0, (hipStream_t)stream, perms, index, c);
  return gpt_launch_status("gpt_perm_index_chain");
}

extern "C" int gpt_perm_gather(const float* const* host_src, float* const* host_dst, int n_nets, const int64_t* index,
                               int n_samples, int D, void* stream) {
  GPT_REQUIRE(host_src && host_dst && index && n_nets > 0 && n_nets <= MAX_NETS && n_samples > 0 && D > 0,
              "gpt_perm_gather: bad arguments (at most %d nets share an index)", MAX_NETS);
  GatherArgs g;
  g.n_nets = n_nets;
  for (int n = 0; n < MAX_NETS; ++n) {
    g.src[n] = n < n_nets ? host_src[n] : nullptr;
    g.dst[n] = n < n_nets ? host_dst[n] : nullptr;
    if (n < n_nets) GPT_REQUIRE(g.src[n] && g.dst[n] && g.src[n] != g.dst[n], "gpt_perm_gather: net %d: null or in-place", n);
  }
  const dim3 grid((D + 255) / 256 > 16 ? 16 : (D + 255) / 256, n_samples);
  hipLaunchKernelGGL(gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, index, D);
  return gpt_launch_status("gpt_perm_gather");
}
