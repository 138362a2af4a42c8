// adam_kernels.hip -- torch.optim.Adam's step for the tensors the reference's training loop optimises (N1, nusc_train.py:1233,
// 1522-1525), on the device path: one launch over all tensors, per-step scalars from a table in device memory indexed by a
// step counter in device memory, so that a whole training step can sit in a HIP graph.  Compile with -ffp-contract=off
// (adam_core.hpp: every operation of torch's float32 update is rounded on its own).
#include "pstl_common.hpp"
#include "adam_core.hpp"

namespace pstl {
namespace {

struct AdamArgs {
  int n;
  int cap;
  float* p[PSTL_ADAM_MAX_TENSORS];
  const float* g[PSTL_ADAM_MAX_TENSORS];
  long numel[PSTL_ADAM_MAX_TENSORS];
  long off[PSTL_ADAM_MAX_TENSORS];   // of this tensor's moments in exp_avg / exp_avg_sq
  float* m;
  float* v;
  const float* sched;   // (cap, 2): -lr / (1 - beta1^t), sqrt(1 - beta2^t) for t = 1 ... cap
  const int* step;      // steps done so far
  float beta2, w1, w2, eps;
};

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
  const int ti = blockIdx.y;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.numel[ti]) return;
  int t = *a.step;                 // this is step t + 1
  if (t > a.cap - 1) t = a.cap - 1;   // (the host grows the table before that can happen; never read past it)
  AdamScalars s;
  s.neg_step_size = a.sched[2 * t];
  s.bc2_sqrt = a.sched[2 * t + 1];
  s.beta2 = a.beta2, s.w1 = a.w1, s.w2 = a.w2, s.eps = a.eps;
  float p = a.p[ti][i], m = a.m[a.off[ti] + i], v = a.v[a.off[ti] + i];
  adam_update(p, m, v, a.g[ti][i], s);
  a.p[ti][i] = p;
  a.m[a.off[ti] + i] = m;
  a.v[a.off[ti] + i] = v;
}

__global__ void k_adam_tick(int* step) { *step = *step + 1; }

}  // namespace
}  // namespace pstl

using namespace pstl;

extern "C" int pstl_adam_step(int n_tensors, float* const* params, const float* const* grads, const int64_t* numel, float* exp_avg,
                              float* exp_avg_sq, const float* sched, int sched_steps, int32_t* step, float one_minus_beta1,
                              float beta2, float one_minus_beta2, float eps, void* stream) {
  if (n_tensors < 1 || n_tensors > PSTL_ADAM_MAX_TENSORS) return PSTL_ERR_SHAPE;
  if (!params || !grads || !numel || !exp_avg || !exp_avg_sq || !sched || !step || sched_steps < 1) return PSTL_ERR_ARG;
  AdamArgs a = {};
  a.n = n_tensors, a.cap = sched_steps;
  long off = 0, longest = 0;
  for (int i = 0; i < n_tensors; ++i) {
    if (!params[i] || !grads[i] || numel[i] < 0) return PSTL_ERR_ARG;
    a.p[i] = params[i], a.g[i] = grads[i], a.numel[i] = (long)numel[i], a.off[i] = off;
    off += (long)numel[i];
    longest = numel[i] > longest ? (long)numel[i] : longest;
  }
  a.m = exp_avg, a.v = exp_avg_sq, a.sched = sched, a.step = step;
  // (the Python scalars of torch.optim.Adam as float32: the caller forms 1 - beta in DOUBLE precision, then rounds)
  a.beta2 = beta2, a.w1 = one_minus_beta1, a.w2 = one_minus_beta2, a.eps = eps;
  if (longest > 0) {
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((longest + 255) / 256), (unsigned)n_tensors), dim3(256), 0, as_stream(stream), a);
    if (int e = launch_status()) return e;
  }
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, as_stream(stream), step);
  return launch_status();
}
