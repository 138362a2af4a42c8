// Issue rate of vector instructions WITHOUT MFMAs beside them (the STL kernels' regime): W waves per SIMD, each running
// chains of independent instructions of one kind; shader cycles per instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dbg/_variants/valu_rate tools/dbg/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(const float* src, float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  float f[8];
  f32x2 p[8];
  unsigned u[8];
  for (int j = 0; j < 8; ++j) f[j] = src[(lane + j) & 1023], u[j] = (unsigned)(lane * 7 + j), p[j] = f32x2{src[(lane + 2 * j) & 1023], src[(lane + 3 * j) & 1023]};
  const float kf = src[5];
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      const int r = m & 7;
      if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[r]) : "v"(kf));
      if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(p[(r + 1) & 7]));
      if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[r]) : "v"(p[(r + 1) & 7]));
      if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[r]) : "v"(p[(r + 1) & 7]));
      if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[r]) : "v"(kf));
      if (KIND == 5) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[r]));
      if (KIND == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(f[r]));
      if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
      if (KIND == 8) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[r]) : "v"(f[(r + 1) & 7]), "v"(kf));
      if (KIND == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(*reinterpret_cast<unsigned long long*>(&p[r])) : "v"(u[r]), "v"(u[(r + 1) & 7]) : "vcc");
      if (KIND == 10) { if (m & 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(p[(r + 1) & 7])); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[r]) : "v"(kf)); }
      if (KIND == 11) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[r]) : "v"(kf)); asm volatile("s_nop 0"); }
      if (KIND == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(kf));                     // one dependent chain
      if (KIND == 13) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[0]) : "v"(p[1]));              // dependent packed chain
      if (KIND == 14) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(f[r]), "v"(kf) : "vcc");
      if (KIND == 20) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u[r]) : "v"(u[(r + 1) & 7]) : "s10", "s11");
      if (KIND == 21) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(f[r]), "v"(kf) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[r]) : "v"(u[(r + 1) & 7])); }
      if (KIND == 22) { asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" :: "v"(f[r]), "v"(kf) : "s10", "s11"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u[r]) : "v"(u[(r + 1) & 7])); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u[(r + 3) & 7]) : "v"(u[(r + 2) & 7])); }
      if (KIND == 23) asm volatile("v_cndmask_b32_e64 %0, %0, 2, s[10:11]" : "+v"(u[r]) :: "s10", "s11");
      if (KIND == 24) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[r]) : "v"(kf));
      if (KIND == 26) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(u[(r + 2) & 7]));
      if (KIND == 27) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[r]) : "v"(u[(r + 1) & 7]));
      if (KIND == 28) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(u[(r + 2) & 7]));
      if (KIND == 29) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[r]) : "v"(kf));
      if (KIND == 30) asm volatile("s_nop 0");
      if (KIND == 31) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[r]) : "s"(kf));
      if (KIND == 32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[r]));
      if (KIND == 33) asm volatile("v_log_f32 %0, %0" : "+v"(f[r]));
      if (KIND == 34) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(f[r]), "v"(kf) : "vcc"); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(r + 3) & 7]) : "v"(kf)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[r]) : "v"(u[(r + 1) & 7])); }
      if (KIND == 35) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n\tv_add_f32 %2, %2, %3" : "+v"(u[r]), "+v"(f[r]) : "v"(u[(r + 1) & 7]), "v"(kf) : "s10", "s11");
      if (KIND == 15) { if (m & 1) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[r])); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[(r + 1) & 7]) : "v"(kf)); }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float fs = 0;
  for (int j = 0; j < 8; ++j) fs += f[j] + (float)u[j] + p[j][0] + p[j][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = fs;
  if (lane == 0) cyc[blockIdx.x * 32 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void kind(const float* src, float* out, unsigned long long* cyc, const char* what) {
  printf("%-34s cycles per instruction and SIMD at 1, 2, 3, 4, 8 waves per SIMD:", what);
  const int ws[5] = {1, 2, 3, 4, 8};
  for (int wi = 0; wi < 5; ++wi) {
    const int W = ws[wi], iters = 400;
    // W waves per SIMD = W workgroups of 256 threads per CU (grid 256 W; small kernel: all co-resident)
    hipLaunchKernelGGL((k<KIND>), dim3(256 * W), dim3(256), 0, 0, src, out, cyc, 10);
    hipLaunchKernelGGL((k<KIND>), dim3(256 * W), dim3(256), 0, 0, src, out, cyc, iters);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 8 * 32];
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * 256 * W * 32, hipMemcpyDeviceToHost);
    double s = 0;
    int n = 0;
    for (int b = 0; b < 256 * W; ++b)
      for (int w = 0; w < 4; ++w) s += (double)h[b * 32 + w], ++n;
    const double per_wave_inst = s / n / iters / 64 / ((KIND == 11) ? 1 : 1);
    printf(" %6.2f", per_wave_inst / W);
  }
  printf("\n");
}
int main() {
  float *src, *out;
  unsigned long long* cyc;
  (void)hipMalloc(&src, 4096);
  (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  (void)hipMalloc(&cyc, 256 * 8 * 32 * 8);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 0.5f + (i % 17) * 0.01f;
  (void)hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
  kind<0>(src, out, cyc, "v_add_f32");
  kind<4>(src, out, cyc, "v_fma_f32");
  kind<1>(src, out, cyc, "v_pk_mul_f32");
  kind<2>(src, out, cyc, "v_pk_add_f32");
  kind<3>(src, out, cyc, "v_pk_fma_f32");
  kind<5>(src, out, cyc, "v_sqrt_f32");
  kind<6>(src, out, cyc, "v_exp_f32");
  kind<7>(src, out, cyc, "v_cndmask_b32");
  kind<8>(src, out, cyc, "v_min3_f32");
  kind<9>(src, out, cyc, "v_mad_u64_u32");
  kind<14>(src, out, cyc, "v_cmp_lt_f32");
  kind<20>(src, out, cyc, "v_cndmask_b32_e64 (sgpr mask)");
  kind<23>(src, out, cyc, "v_cndmask_b32_e64 v, v, 2, sgpr");
  kind<21>(src, out, cyc, "v_cmp vcc + v_cndmask vcc (per instr /2)");
  kind<22>(src, out, cyc, "v_cmp_e64 + 2 v_cndmask_e64 (/3)");
  kind<34>(src, out, cyc, "v_cmp, v_add, v_cndmask (/3)");
  kind<35>(src, out, cyc, "v_cndmask_e64, v_add (/2)");
  kind<24>(src, out, cyc, "v_max_f32");
  kind<26>(src, out, cyc, "v_and_or_b32");
  kind<27>(src, out, cyc, "v_min_u32");
  kind<28>(src, out, cyc, "v_min3_u32");
  kind<29>(src, out, cyc, "v_mul_f32");
  kind<31>(src, out, cyc, "v_sub_f32 v, v, sgpr");
  kind<32>(src, out, cyc, "v_rcp_f32");
  kind<33>(src, out, cyc, "v_log_f32");
  kind<30>(src, out, cyc, "s_nop 0");
  kind<10>(src, out, cyc, "v_add_f32 | v_pk_mul_f32 alternating");
  kind<15>(src, out, cyc, "v_add_f32 | v_sqrt_f32 alternating");
  kind<11>(src, out, cyc, "v_add_f32 + s_nop 0 (per pair)");
  kind<12>(src, out, cyc, "v_add_f32, ONE dependent chain");
  kind<13>(src, out, cyc, "v_pk_mul_f32, ONE dependent chain");
  return 0;
}
