// What a wave pays for ISSUING a vector memory instruction between MFMAs (one wave per SIMD, as in k_chain2): per iteration
// K v_mfma_f32_16x16x32_f16 (16 pipe cycles each) and one memory operation of kind MODE; s_waitcnt vmcnt(0) every 4
// iterations.  Output: shader cycles per iteration minus the K MFMAs' 16 K.
//   MODE 0 nothing | 1 one global_load_lds_dwordx4 (1 KB, L2-resident source) | 2 two of them back to back (shared M0)
//   3 one global_load_dwordx4 into registers | 4 one global_load_lds_dword | 5 one ds_read_b128 | 6 two DMAs, one per half iteration
//   7 one DMA without the M0 write | 8 four DMAs behind one M0 write | 9 the M0 write alone
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_issue_cost tools/dbg/dma_issue_cost.hip && /tmp/dma_issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int K>
__global__ __launch_bounds__(256) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)src[(threadIdx.x + j) & 1023], b[j] = (_Float16)src[(threadIdx.x + 9 * j) & 1023];
  f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  f32x4 sink = f32x4{0, 0, 0, 0};
  const unsigned lds_base = (unsigned)w * 16384u;
  const unsigned voff = (unsigned)lane * 16u;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_mov_b32 m0, %0" ::"s"(lds_base) : "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    const float* sp = src + ((it * 4 + w) & 255) * 256;   // 1 KB blocks of a 256 KB window
    const unsigned dst = lds_base + (unsigned)(it & 7) * 2048u;
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sp), "s"(dst) : "memory");
    if (MODE == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(sp), "s"(dst) : "memory");
    if (MODE == 3) {
      f32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sp) : "memory");
      asm volatile("" ::"v"(v));
    }
    if (MODE == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff >> 2), "s"(sp), "s"(dst) : "memory");
    if (MODE == 5) sink += *reinterpret_cast<const f32x4*>(smem + dst + voff);
    if (MODE == 6) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sp), "s"(dst) : "memory");
    if (MODE == 7) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sp) : "memory");   // M0 as the prologue left it
    if (MODE == 8)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff), "s"(sp), "s"(dst) : "memory");
    if (MODE == 9) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory");   // the M0 write alone
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < K; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 6 && m == K / 2) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sp + 256), "s"(dst + 1024u) : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  const f32x4 s = acc[0] + acc[1] + acc[2] + acc[3] + sink;
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
  if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

template <int MODE, int K>
void run(const float* src, float* out, unsigned long long* cyc, const char* what) {
  const int iters = 4000, lds = 72 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, K>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(256), lds, 0, src, out, cyc, 50);
  hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(256), lds, 0, src, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 1024; ++i) s += (double)h[i];
  const double per = s / 1024 / iters;
  printf("K=%2d %-52s %7.1f cycles per iteration = 16 K %+7.1f\n", K, what, per, per - 16.0 * K);
}

template <int K>
void all(const float* src, float* out, unsigned long long* cyc) {
  run<0, K>(src, out, cyc, "MFMAs only");
  run<1, K>(src, out, cyc, "one global_load_lds_dwordx4");
  run<2, K>(src, out, cyc, "two global_load_lds_dwordx4 back to back");
  run<6, K>(src, out, cyc, "two global_load_lds_dwordx4 half an iteration apart");
  run<3, K>(src, out, cyc, "one global_load_dwordx4 into registers");
  run<4, K>(src, out, cyc, "one global_load_lds_dword");
  run<5, K>(src, out, cyc, "one ds_read_b128");
  run<7, K>(src, out, cyc, "one global_load_lds_dwordx4, M0 not rewritten");
  run<8, K>(src, out, cyc, "four global_load_lds_dwordx4 behind one M0 write");
  run<9, K>(src, out, cyc, "the M0 write alone");
}

int main() {
  float *src, *out;
  unsigned long long* cyc;
  (void)hipMalloc(&src, 1 << 20);
  (void)hipMalloc(&out, 256 * 256 * 4);
  (void)hipMalloc(&cyc, 1024 * 8);
  (void)hipMemset(src, 0, 1 << 20);
  all<24>(src, out, cyc);
  all<8>(src, out, cyc);
  all<4>(src, out, cyc);
  return 0;
}
