// Batched evaluation of sampled MNIST-MLP parameter vectors (the quality metric of the sampling harness):
// replaces moduleify + the per-net Python loop of batch_test_fn (Gpt/vis.py:398-413, 539-570) around the task's
// test_epoch (data_gen/train_mnist.py:58-71, 324-332):  logits = fc2(relu(fc1(x)));  loss = mean cross-entropy,
// err = 100 * mean(argmax != label).  fc1 of ALL nets runs as one exact-fp32 MFMA GEMM (gpt_gemm_nt, host side);
// this kernel is the rest: bias + ReLU + the tiny second layer + log-softmax + the mean over the images, per net.
#include "common.h"

namespace {

constexpr int MAX_HIDDEN = 64, MAX_CLASSES = 16;

// H: [B * hidden, ldh] fc1 pre-activations WITHOUT bias, image index contiguous.  One workgroup per net.
__global__ __launch_bounds__(256) void mlp_head_kernel(const float* __restrict__ H, int64_t ldh, const float* __restrict__ nets,
                                                       int64_t net_stride, int off_b1, int off_w2, int off_b2,
                                                       const int64_t* __restrict__ labels, float* __restrict__ loss,
                                                       float* __restrict__ err, int hidden, int classes, int n_img) {
  __shared__ float b1[MAX_HIDDEN], w2[MAX_CLASSES * MAX_HIDDEN], b2[MAX_CLASSES];
  __shared__ float red[2][4];
  const int b = blockIdx.x;
  const float* net = nets + (int64_t)b * net_stride;
  for (int i = threadIdx.x; i < hidden; i += 256) b1[i] = net[off_b1 + i];
  for (int i = threadIdx.x; i < classes * hidden; i += 256) w2[i] = net[off_w2 + i];
  for (int i = threadIdx.x; i < classes; i += 256) b2[i] = net[off_b2 + i];
  __syncthreads();
  const float* Hb = H + (int64_t)b * hidden * ldh;
  float sum_loss = 0.f, sum_err = 0.f;
  for (int img = threadIdx.x; img < n_img; img += 256) {
    float logit[MAX_CLASSES];
    for (int c = 0; c < classes; ++c) logit[c] = b2[c];
    for (int j = 0; j < hidden; ++j) {
      const float h = fmaxf(Hb[(int64_t)j * ldh + img] + b1[j], 0.f);
      for (int c = 0; c < classes; ++c) logit[c] += w2[c * hidden + j] * h;
    }
    float mx = logit[0];
    int arg = 0;
    for (int c = 1; c < classes; ++c)
      if (logit[c] > mx) { mx = logit[c]; arg = c; }      // first maximum, as torch.argmax
    float se = 0.f;
    for (int c = 0; c < classes; ++c) se += expf(logit[c] - mx);
    const int lab = (int)labels[img];
    sum_loss += (mx + logf(se)) - logit[lab];
    sum_err += arg != lab ? 1.f : 0.f;
  }
  sum_loss = wave_sum(sum_loss);
  sum_err = wave_sum(sum_err);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sum_loss; red[1][threadIdx.x >> 6] = sum_err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    loss[b] = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)n_img;
    err[b] = 100.f * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)n_img;
  }
}

// ---- CIFAR-10 ConvNet (data_gen/train_cifar10.py:53-68) -----------------------------------------------------------------
//   conv1 3 -> 16, 3x3, stride 2, pad 1, no bias, ReLU  (32 x 32 -> 16 x 16)
//   conv2 16 -> 32, 3x3, stride 1, pad 1, no bias, ReLU (16 x 16)
//   global average pool -> fc 32 -> 10 (+ bias) -> cross-entropy / top-1 error (test_epoch, :426-434)
// Flat vector in state_dict order: conv1.weight [16,3,3,3] | conv2.weight [32,16,3,3] | fc.weight [10,32] | fc.bias [10].
// Every net has its own weights in BOTH convolutions, so nothing but the images is shared between nets: 2.6 MFLOP per
// (net, image), 33 TFLOP for 1280 nets x the 10 000 test images, 91 % of it in conv2.  One wave owns one net: conv2's
// weights live in 72 VGPRs as the B operands of v_mfma_f32_32x32x2_f32 (exact fp32; rows = 32 output positions, columns =
// the 32 output channels, K = 144 walked as (input-channel pair, ky, kx) so that the A operand of every step is ONE
// ds_read_b32 at a compile-time offset from a per-lane base into the zero-haloed conv1 activations in LDS).  conv1 (9 % of
// the FLOPs, K = 27) runs on the VALU from the image in LDS, which the four waves (nets) of a workgroup share and which is
// double-buffered against the next image's global loads.  The ReLU + pooling happens on the accumulators (channel =
// lane & 31: 16 in-lane adds per tile), the 32 -> 10 layer and the log-softmax on cross-lane sums.
constexpr int CNN_D = 5370, CNN_OFF_W2 = 432, CNN_OFF_FC = 5040, CNN_OFF_FCB = 5360;
constexpr int IMG_PITCH = 33, IMG_PLANE = 33 * 33, IMG_FLOATS = 3 * IMG_PLANE;   // one zero row / column before index 0
constexpr int H1_PITCH = 18, H1_PLANE = 18 * 18, H1_FLOATS = 16 * H1_PLANE;      // zero halo all round
constexpr int CNN_WAVE_FLOATS = H1_FLOATS + 432;
constexpr int CNN_LDS_FLOATS = 2 * IMG_FLOATS + 4 * CNN_WAVE_FLOATS;
constexpr int CNN_LDS_BYTES = CNN_LDS_FLOATS * 4;

typedef float cnn_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 1) void cnn_eval_kernel(const float* __restrict__ images, const int64_t* __restrict__ labels,
                                                          const float* __restrict__ nets, int64_t net_stride,
                                                          float* __restrict__ partial, int B, int n_img, int chunks) {
  extern __shared__ float cnn_lds[];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, hh = lane >> 5, l31 = lane & 31;
  float* h1 = cnn_lds + 2 * IMG_FLOATS + wid * CNN_WAVE_FLOATS;
  float* w1 = h1 + H1_FLOATS;
  const int net_raw = blockIdx.x * 4 + wid, chunk = blockIdx.y;
  const float* net = nets + (int64_t)min(net_raw, B - 1) * net_stride;      // a ragged last group recomputes net B - 1
  const int per = (n_img + chunks - 1) / chunks, i0 = chunk * per, i1 = min(n_img, i0 + per);

  for (int i = threadIdx.x; i < CNN_LDS_FLOATS; i += 256) cnn_lds[i] = 0.f;   // the halos stay zero from here on
  __syncthreads();
  for (int i = lane; i < 432; i += 64) w1[i] = net[i];
  float wreg[72];                      // conv2.weight[out = l31][cin = 2 * cin2 + hh][ky][kx],  j = cin2 * 9 + ky * 3 + kx
#pragma unroll
  for (int j = 0; j < 72; ++j) wreg[j] = net[CNN_OFF_W2 + l31 * 144 + (2 * (j / 9) + hh) * 9 + (j % 9)];
  float fcw[10], fcb[10];
#pragma unroll
  for (int c = 0; c < 10; ++c) {
    fcw[c] = net[CNN_OFF_FC + c * 32 + l31];
    fcb[c] = net[CNN_OFF_FCB + c];
  }

  // image staging: thread t moves floats [4 t, 4 t + 4) of each colour plane
  float4 stage[3];
  auto fetch = [&](int i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) stage[c] = reinterpret_cast<const float4*>(images + (int64_t)i * 3072 + c * 1024)[threadIdx.x];
  };
  auto put = [&](float* img) {
    const int y = threadIdx.x >> 3, x = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float* d = img + c * IMG_PLANE + (y + 1) * IMG_PITCH + x + 1;
      d[0] = stage[c].x; d[1] = stage[c].y; d[2] = stage[c].z; d[3] = stage[c].w;
    }
  };
  if (i0 < i1) {
    fetch(i0);
    put(cnn_lds);
  }
  __syncthreads();

  float sum_loss = 0.f, sum_err = 0.f;
  const int py = lane >> 4, px = lane & 15;
  for (int i = i0; i < i1; ++i) {
    const float* img = cnn_lds + ((i - i0) & 1) * IMG_FLOATS;
    const bool more = i + 1 < i1;
    if (more) fetch(i + 1);
    // ---- conv1 + ReLU: lane -> positions (y = py + 4 q, x = px), q < 4, all 16 channels
    {
      float in[4][27];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
              in[q][c * 9 + ky * 3 + kx] = img[c * IMG_PLANE + (2 * (py + 4 * q) + ky) * IMG_PITCH + 2 * px + kx];
#pragma unroll 4
      for (int o = 0; o < 16; ++o) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 27; ++k) {
          const float w = w1[o * 27 + k];
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q] = fmaf(w, in[q][k], a[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) h1[o * H1_PLANE + (py + 4 * q + 1) * H1_PITCH + px + 1] = fmaxf(a[q], 0.f);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();     // h1 is written and read by this wave only: LDS serves one wave in order
    // ---- conv2 on the matrix pipe, ReLU, pooling
    float g = 0.f;
#pragma unroll 1
    for (int tp = 0; tp < 4; ++tp) {
      // tiles 2 tp, 2 tp + 1: 32 positions each = rows 2 t, 2 t + 1 of the 16 x 16 map
      const float* b0 = h1 + hh * H1_PLANE + (4 * tp + (l31 >> 4)) * H1_PITCH + (l31 & 15);
      const float* b1 = b0 + 2 * H1_PITCH;
      cnn_f32x16 acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
      for (int j = 0; j < 72; ++j) {
        const int off = (j / 9) * 2 * H1_PLANE + ((j % 9) / 3) * H1_PITCH + (j % 3);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[off], wreg[j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[off], wreg[j], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) g += fmaxf(acc0[e], 0.f) + fmaxf(acc1[e], 0.f);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();     // the next image's conv1 overwrites h1 only after these reads
    g += __shfl_xor(g, 32);
    g *= (1.f / 256.f);                  // AdaptiveAvgPool2d((1, 1)) over 16 x 16
    float logit[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      float v = fcw[c] * g;
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      logit[c] = v + fcb[c];
    }
    float mx = logit[0];
    int arg = 0;
#pragma unroll
    for (int c = 1; c < 10; ++c)
      if (logit[c] > mx) { mx = logit[c]; arg = c; }       // first maximum, as torch.argmax
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 10; ++c) se += expf(logit[c] - mx);
    const int lab = (int)labels[i];
    float at = logit[0];
#pragma unroll
    for (int c = 1; c < 10; ++c) at = lab == c ? logit[c] : at;
    sum_loss += (mx + logf(se)) - at;
    sum_err += arg != lab ? 1.f : 0.f;
    if (more) put(cnn_lds + (((i - i0) & 1) ^ 1) * IMG_FLOATS);
    __syncthreads();
  }
  if (lane == 0 && net_raw < B) {
    partial[((int64_t)net_raw * chunks + chunk) * 2 + 0] = sum_loss;
    partial[((int64_t)net_raw * chunks + chunk) * 2 + 1] = sum_err;
  }
}

__global__ void cnn_finish_kernel(const float* __restrict__ partial, float* __restrict__ loss, float* __restrict__ err, int B,
                                  int chunks, int n_img) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float sl = 0.f, se = 0.f;
  for (int c = 0; c < chunks; ++c) {
    sl += partial[((int64_t)b * chunks + c) * 2];
    se += partial[((int64_t)b * chunks + c) * 2 + 1];
  }
  loss[b] = sl / (float)n_img;
  err[b] = 100.f * se / (float)n_img;
}

}  // namespace

extern "C" int gpt_cnn_eval(const float* images, const int64_t* labels, const float* nets, int64_t net_stride, float* partial,
                            int chunks, float* loss, float* err, int B, int n_img, void* stream) {
  GPT_REQUIRE(images && labels && nets && partial && loss && err, "gpt_cnn_eval: null operand");
  GPT_REQUIRE(B > 0 && n_img > 0 && chunks > 0 && chunks <= n_img && chunks <= 65535, "gpt_cnn_eval: need B, n_img > 0 and 0 < chunks <= min(n_img, 65535)");
  GPT_REQUIRE(net_stride >= CNN_D && ((uintptr_t)images % 16) == 0, "gpt_cnn_eval: net_stride >= %d and 16-byte aligned images", CNN_D);
  static GptPerDevice attr;   // the LDS grant is per kernel AND per device (a process may drive several GPUs)
  bool& attr_set = attr.here();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)cnn_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CNN_LDS_BYTES) != hipSuccess) {
      gpt_set_error("gpt_cnn_eval: cannot reserve %d bytes of LDS", CNN_LDS_BYTES);
      return -1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(cnn_eval_kernel, dim3((B + 3) / 4, chunks), dim3(256), CNN_LDS_BYTES, (hipStream_t)stream, images, labels, nets,
                     net_stride, partial, B, n_img, chunks);
  hipLaunchKernelGGL(cnn_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, loss, err, B, chunks, n_img);
  return gpt_launch_status("gpt_cnn_eval");
}

extern "C" int gpt_mlp_head(const float* H, int64_t ldh, const float* nets, int64_t net_stride, int off_b1, int off_w2,
                            int off_b2, const int64_t* labels, float* loss, float* err, int B, int hidden, int classes,
                            int n_img, void* stream) {
  GPT_REQUIRE(H && nets && labels && loss && err, "gpt_mlp_head: null operand");
  GPT_REQUIRE(B > 0 && n_img > 0 && hidden > 0 && hidden <= MAX_HIDDEN && classes > 1 && classes <= MAX_CLASSES,
              "gpt_mlp_head: need 0 < hidden <= %d, 1 < classes <= %d", MAX_HIDDEN, MAX_CLASSES);
  GPT_REQUIRE(ldh >= n_img && off_b1 >= 0 && off_w2 >= 0 && off_b2 >= 0, "gpt_mlp_head: bad layout");
  hipLaunchKernelGGL(mlp_head_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, H, ldh, nets, net_stride, off_b1, off_w2, off_b2,
                     labels, loss, err, hidden, classes, n_img);
  return gpt_launch_status("gpt_mlp_head");
}
