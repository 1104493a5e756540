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

}  // namespace

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
