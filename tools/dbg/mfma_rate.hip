// Issue rate of v_mfma_f32_16x16x32_bf16 on gfx950 as a function of the number of independent accumulators and of
// the waves per SIMD:  hipcc --offload-arch=gfx950 -O3 tools/dbg/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void k(const float* in, float* out, int iters) {
  bf16x8 a[8], b;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)in[(threadIdx.x + i * 8 + j) & 1023];
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)in[(threadIdx.x * 3 + j) & 1023];
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 48; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 7], b, acc[m % NACC], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 1024 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int NACC>
void run(int threads, const float* in, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, in, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = 48.0 * iters * (threads / 256);   // MFMAs per SIMD
  printf("acc=%d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", NACC, threads / 256, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096);
  hipMalloc(&out, 256 * 1024 * 4);
  hipMemset(in, 0, 4096);
  for (int t : {256, 512, 1024}) {
    run<1>(t, in, out);
    run<2>(t, in, out);
    run<3>(t, in, out);
    run<4>(t, in, out);
    run<6>(t, in, out);
  }
  return 0;
}
