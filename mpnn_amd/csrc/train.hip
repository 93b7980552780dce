// Loss and optimizer step of the training loop, as single stream-ordered launches that never read
// anything back to the host (so a whole training step can be captured into a HIP graph).
//
//   masked mean cross-entropy   loss = mean over rows with mask != 0 of -log_softmax(logits[row])[label[row]]
//                               what the reference computes as F.cross_entropy(out[data.train_mask],
//                               data.y[data.train_mask]) (training/train.py:47, experiments/flickr_exp.py:31);
//                               the boolean-mask indexing there compacts rows on the host's say-so, here the
//                               mask only weights the reduction
//   Adam                        torch.optim.Adam(lr) as the reference constructs it (experiments/*_exp.py:26,33,49:
//                               betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad), over ONE flat buffer
#include "common.h"

namespace mpnn {

constexpr int kCeMaxBlocks = 1024;

// One thread per row.  grad[row, c] = mask * (softmax(row)[c] - [c == label]) (not yet divided by the count);
// block sums of the masked row losses and of the mask go to partial[2 * block + {0, 1}].
__global__ void __launch_bounds__(kBlock) k_masked_ce(const float* __restrict__ logits, int64_t ld,
                                                      const int64_t* __restrict__ labels,
                                                      const uint8_t* __restrict__ mask, int64_t n, int c,
                                                      float* __restrict__ grad, int64_t ldg,
                                                      float* __restrict__ partial) {
  __shared__ float s_loss[kBlock];
  __shared__ float s_cnt[kBlock];
  float loss = 0.f, cnt = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < n; row += (int64_t)gridDim.x * kBlock) {
    const float* x = logits + row * ld;
    float* g = grad + row * ldg;
    const int64_t y = labels[row];
    // F.cross_entropy's default ignore_index: such a row adds nothing to the sum, the count or the gradient
    const bool on = (mask ? mask[row] != 0 : true) && y != -100;
    if (!on) {
      for (int j = 0; j < c; ++j) g[j] = 0.f;
      continue;
    }
    float mx = x[0];
    for (int j = 1; j < c; ++j) mx = fmaxf(mx, x[j]);
    float se = 0.f;
    for (int j = 0; j < c; ++j) se += expf(x[j] - mx);
    const float lse = logf(se) + mx;
    const float inv = 1.f / se;
    for (int j = 0; j < c; ++j) g[j] = expf(x[j] - mx) * inv - (j == y ? 1.f : 0.f);
    // any other label outside [0, c) is an error in torch; here it turns the loss into NaN instead of passing unseen
    loss += (y >= 0 && y < c) ? lse - x[y] : __builtin_nanf("");
    cnt += 1.f;
  }
  s_loss[threadIdx.x] = loss;
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s_loss[threadIdx.x] += s_loss[threadIdx.x + off];
      s_cnt[threadIdx.x] += s_cnt[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = s_loss[0];
    partial[2 * blockIdx.x + 1] = s_cnt[0];
  }
}

// result[0] = sum(loss) / sum(count), result[1] = 1 / sum(count).  One wavefront: lane l adds the partials of blocks
// l, l + 64, .. in ascending order, then the 64 sums meet in a fixed tree - the same order on every run.
__global__ void k_ce_finish(const float* __restrict__ partial, int blocks, float* __restrict__ result) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  float loss = 0.f, cnt = 0.f;
  for (int b = threadIdx.x; b < blocks; b += 64) {
    loss += partial[2 * b];
    cnt += partial[2 * b + 1];
  }
  for (int off = 32; off > 0; off >>= 1) {
    loss += __shfl_down(loss, off, 64);
    cnt += __shfl_down(cnt, off, 64);
  }
  if (threadIdx.x == 0) {
    result[0] = loss / cnt;
    result[1] = 1.f / cnt;
  }
}

// state[0] = step count (as float, exact below 2^24), state[1] = lr / (1 - b1^t), state[2] = sqrt(1 - b2^t)
__global__ void k_adam_prepare(float* __restrict__ state, float lr, float b1, float b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double t = (double)state[0] + 1.0;
  state[0] = (float)t;
  state[1] = (float)((double)lr / (1.0 - pow((double)b1, t)));
  state[2] = (float)sqrt(1.0 - pow((double)b2, t));
}

__global__ void __launch_bounds__(kBlock) k_adam(float* __restrict__ p, const float* __restrict__ g,
                                                 float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                 const float* __restrict__ state, float b1, float b2, float eps) {
  const float step_size = state[1], sqrt_bc2 = state[2];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);        // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] -= step_size * (mi / denom);
  }
}

}  // namespace mpnn

using namespace mpnn;

extern "C" {

int64_t mpnn_masked_ce_scratch_floats(void) { return 2 * kCeMaxBlocks; }

int32_t mpnn_masked_ce(const float* logits, int64_t ld, const int64_t* labels, const uint8_t* mask, int64_t n,
                       int32_t c, float* grad, int64_t ldg, float* result, float* scratch, mpnn_stream_t stream) {
  if (n < 0 || c < 1) return fail(MPNN_E_INVALID, "mpnn_masked_ce: n=%lld c=%d", (long long)n, c);
  if (!result || !scratch || (n > 0 && (!logits || !labels || !grad)))
    return fail(MPNN_E_INVALID, "mpnn_masked_ce: null pointer argument");
  if (ld < c || ldg < c) return fail(MPNN_E_INVALID, "mpnn_masked_ce: leading dimension below the class count");
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > kCeMaxBlocks) blocks = kCeMaxBlocks;
  k_masked_ce<<<(int)blocks, kBlock, 0, stream>>>(logits, ld, labels, mask, n, c, grad, ldg, scratch);
  MPNN_LAUNCH_CHECK("k_masked_ce");
  k_ce_finish<<<1, 64, 0, stream>>>(scratch, (int)blocks, result);
  MPNN_LAUNCH_CHECK("k_ce_finish");
  return MPNN_OK;
}

int32_t mpnn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float* state, float lr, float beta1, float beta2, float eps, mpnn_stream_t stream) {
  if (n < 0) return fail(MPNN_E_INVALID, "mpnn_adam_step: n=%lld", (long long)n);
  if (!state || (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq)))
    return fail(MPNN_E_INVALID, "mpnn_adam_step: null pointer argument");
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f))
    return fail(MPNN_E_INVALID, "mpnn_adam_step: betas must lie in [0, 1) and eps must not be negative");
  k_adam_prepare<<<1, 64, 0, stream>>>(state, lr, beta1, beta2);
  MPNN_LAUNCH_CHECK("k_adam_prepare");
  if (n > 0) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    k_adam<<<(int)blocks, kBlock, 0, stream>>>(params, grads, exp_avg, exp_avg_sq, n, state, beta1, beta2, eps);
    MPNN_LAUNCH_CHECK("k_adam");
  }
  return MPNN_OK;
}

}  // extern "C"
