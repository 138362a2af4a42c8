// adam_core.hpp -- one element of torch.optim.Adam's step (the optimiser of the reference's training loop,
// nusc_train.py:1233: Adam over rect_net.parameters() -- net.parameters() with --joint --, default betas, eps, no weight decay,
// no amsgrad), operation for operation what torch's single-tensor CPU path computes in float32
// (torch/optim/adam.py: _single_tensor_adam; ATen lerp / addcmul / addcdiv kernels):
//   exp_avg.lerp_(grad, 1 - beta1)                          m = fma(w, g - m, m)  if |w| < 0.5 (beta1 > 0.5: the defaults),
//                                                               fma(w - 1, g - m, g)  otherwise (ATen's lerp keeps the weight small)
//   exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)  v = v beta2 + ((1 - beta2) g) g
//   denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//   param.addcdiv_(exp_avg, denom, value=-step_size)        p = p + (value m) / denom
// with step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t) computed by the HOST in double precision and
// handed over as float32 (exactly what torch does with its Python scalars).  Every operation is rounded on its own: the
// translation unit is compiled with -ffp-contract=off, the divisions and the square root are IEEE.  __host__ __device__ so
// that the CPU tests can hold the very same function against torch.optim.Adam bit for bit (tests/test_adam_core_hostsim.py).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define PSTL_ADAM_HD __host__ __device__ __forceinline__
#else
#define PSTL_ADAM_HD inline
#endif

namespace pstl {

struct AdamScalars {
  float neg_step_size;   // -lr / (1 - beta1^t)
  float bc2_sqrt;        // sqrt(1 - beta2^t)
  float beta2;           // float32(beta2)
  float w1;              // float32(1 - beta1)   (the double difference, then rounded: what a Python scalar becomes)
  float w2;              // float32(1 - beta2)
  float eps;
};

PSTL_ADAM_HD void adam_update(float& p, float& m, float& v, float g, const AdamScalars& s) {
  m = fabsf(s.w1) < 0.5f ? fmaf(s.w1, g - m, m) : fmaf(s.w1 - 1.0f, g - m, g);
  v = fmaf(s.w2 * g, g, v * s.beta2);
  const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
  p = p + (s.neg_step_size * m) / denom;
}

}  // namespace pstl
