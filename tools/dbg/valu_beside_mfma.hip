// What n instructions of one kind cost beside each v_mfma_f32_16x16x32_f16 of ONE wave per SIMD (k_chain2's regime): a loop of
// 32 slots = [MFMA, n fillers], four accumulators round robin, fillers on registers the MFMAs do not touch; shader cycles per slot.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dbg/_variants/valu_beside_mfma tools/dbg/valu_beside_mfma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NF>
__global__ __launch_bounds__(256) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)src[(threadIdx.x + j) & 1023], b[j] = (_Float16)src[(threadIdx.x + 9 * j) & 1023];
  f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  float f[8];
  f32x2 p[4];
  unsigned u[8];
  f32x4 q = f32x4{0, 0, 0, 0};
  for (int j = 0; j < 8; ++j) f[j] = src[(lane + j) & 1023], u[j] = (unsigned)(lane * 7 + j);
  for (int j = 0; j < 4; ++j) p[j] = f32x2{f[j], f[j + 4]};
  const float kf = src[5];
  const char* lp = smem + w * 8192 + lane * 16;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        const int r = (m * NF + n) & 7;
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[r]) : "v"(kf));
        if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[r & 3]) : "s"(f32x2{kf, kf}));
        if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(f[r]), "v"(f[(r + 1) & 7]));
        if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(u[r]) : "v"(f[r]), "s"(kf));
        if (KIND == 4) asm volatile("v_max_i32 %0, 0, %0" : "+v"(u[r]));
        if (KIND == 5) asm volatile("s_nop 0");
        if (KIND == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(*reinterpret_cast<unsigned long long*>(&p[r & 3])) : "v"(u[r]), "v"(u[(r + 1) & 7]) : "vcc");
        if (KIND == 7) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 3) & 7]));
        if (KIND == 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"((unsigned)(size_t)lp), "n"(0));
        if (KIND == 9) asm volatile("v_accvgpr_write_b32 a0, %0" ::"v"(u[r]) : "a0");
        if (KIND == 10) asm volatile("v_log_f32 %0, %0" : "+v"(f[r]));
        if (KIND == 11) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3] + q;
  float fs = 0;
  for (int j = 0; j < 8; ++j) fs += f[j] + (float)u[j];
  for (int j = 0; j < 4; ++j) fs += p[j][0] + p[j][1];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w + fs;
  if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

template <int KIND, int NF>
double run1(const float* src, float* out, unsigned long long* cyc) {
  const int iters = 300, lds = 72 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(256), lds, 0, src, out, cyc, 20);
  hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(256), lds, 0, src, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 1024; ++i) s += (double)h[i];
  return s / 1024 / iters / 32;
}
template <int KIND>
void kind(const float* src, float* out, unsigned long long* cyc, const char* what) {
  printf("%-22s cycles per [MFMA + n] slot, n = 0..6: %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", what, run1<KIND, 0>(src, out, cyc),
         run1<KIND, 1>(src, out, cyc), run1<KIND, 2>(src, out, cyc), run1<KIND, 3>(src, out, cyc), run1<KIND, 4>(src, out, cyc),
         run1<KIND, 5>(src, out, cyc), run1<KIND, 6>(src, out, cyc));
}
int main() {
  float *src, *out;
  unsigned long long* cyc;
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&cyc, 1024 * 8);
  (void)hipMemset(src, 0, 1 << 20);
  kind<0>(src, out, cyc, "v_add_f32");
  kind<1>(src, out, cyc, "v_pk_mul_f32");
  kind<2>(src, out, cyc, "v_cvt_pk_f16_f32");
  kind<3>(src, out, cyc, "v_fma_mixlo_f16");
  kind<4>(src, out, cyc, "v_max_i32");
  kind<7>(src, out, cyc, "v_pk_max_u16");
  kind<6>(src, out, cyc, "v_mad_u64_u32");
  kind<10>(src, out, cyc, "v_log_f32");
  kind<9>(src, out, cyc, "v_accvgpr_write_b32");
  kind<8>(src, out, cyc, "ds_read_b128");
  kind<5>(src, out, cyc, "s_nop 0");
  kind<11>(src, out, cyc, "s_add_u32");
  return 0;
}
