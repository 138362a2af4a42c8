// Closer to layer 2 of the chain kernel: 48 v_mfma_f32_16x16x32_bf16 per iteration on TWO accumulators, 32 distinct
// A operands held in registers (128 VGPRs), B operands (a) two fixed registers, (b) read from LDS one k-block ahead.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, int iters) {
  __shared__ u32x4 lds[16 * 64];
  const int lane = threadIdx.x & 63;
  bf16x8 ah[2][8], al[2][8];
  for (int o = 0; o < 2; ++o)
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) {
        ah[o][i][j] = (__bf16)in[(threadIdx.x + o * 64 + i * 8 + j) & 1023];
        al[o][i][j] = (__bf16)in[(threadIdx.x + o * 64 + i * 8 + j + 7) & 1023];
      }
  for (int i = threadIdx.x; i < 16 * 64; i += 512) lds[i] = u32x4{0, 0, 0, 0};
  __syncthreads();
  f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  const u32x4* hb = lds + lane;
  for (int it = 0; it < iters; ++it) {
    u32x4 ch = hb[0], cl = hb[64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      u32x4 nh = ch, nl = cl;
      if (MODE == 1 && kb < 7) {
        nh = hb[(2 * kb + 2) * 64];
        nl = hb[(2 * kb + 3) * 64];
      }
      const bf16x8 bh = __builtin_bit_cast(bf16x8, ch), bl = __builtin_bit_cast(bf16x8, cl);
#pragma unroll
      for (int o = 0; o < 2; ++o) acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[o][kb], bh, acc[o], 0, 0, 0);
#pragma unroll
      for (int o = 0; o < 2; ++o) acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[o][kb], bh, acc[o], 0, 0, 0);
#pragma unroll
      for (int o = 0; o < 2; ++o) acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[o][kb], bl, acc[o], 0, 0, 0);
      ch = nh;
      cl = nl;
      if (MODE == 1 && kb < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const f32x4 s = acc[0] + acc[1];
  out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int MODE>
void run(const float* in, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, in, out, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = 48.0 * iters * 2;
  printf("mode %d (B %s): %.3f ms, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", MODE,
         MODE ? "from LDS, one k-block ahead" : "in two registers", ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main() {
  float *in, *out;
  (void)hipMalloc(&in, 4096);
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMemset(in, 0, 4096);
  run<0>(in, out);
  run<1>(in, out);
  return 0;
}
