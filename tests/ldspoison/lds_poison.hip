// lds_poison.hip -- TEST INFRASTRUCTURE.  Fills the LDS of every CU with hostile garbage (infinities and NaNs, as fp32 and as
// packed halves) so that a kernel which reads LDS it never wrote meets NaNs instead of whatever the previous kernel left
// there -- the -m gpu tests call it before every test.  (Round 3: the latency layout of k_chain read a piece buffer nobody
// had written for empty pipeline slots; in a fresh process that garbage now and then looked like an overflowed half and set
// the split-f16 domain flag -- one bench process in twenty.)  Never loaded by the product path.
#include <hip/hip_runtime.h>

namespace {
__global__ void k_poison(int words) {
  extern __shared__ unsigned smem[];
  // Every half the largest finite one (65504), every fp32 word 2.66e36: LARGE FINITE values.  Infinities and NaNs would be the
  // obvious poison and are the wrong one here: a sum of +-inf products is a NaN, and max(NaN, 0) = 0 -- the matrix pipe's NaNs
  // vanish in the next ReLU -- whereas large finite garbage sails through the ReLUs and overflows the next conversion to half,
  // which is exactly how the stale LDS of a fresh box tripped the domain guard.
  for (int i = threadIdx.x; i < words; i += blockDim.x) smem[i] = 0x7bff7bffu;
  __syncthreads();
  if (smem[(threadIdx.x * 7) % words] == 0u) smem[0] = 1u;   // (keeps the stores)
}
}  // namespace

extern "C" int lds_poison(void* stream) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
  const int bytes = 160 * 1024;   // all of a CU's LDS: one workgroup per CU at a time, several rounds to reach every CU
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return -2;
  hipLaunchKernelGGL(k_poison, dim3(4 * p.multiProcessorCount), dim3(256), bytes, static_cast<hipStream_t>(stream), bytes / 4);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// diagnostics: what a kernel finds in LDS it has not written (word `index` of every workgroup of one wave)
namespace {
__global__ void k_peek(int index, unsigned* out) {
  extern __shared__ unsigned smem[];
  if (threadIdx.x == 0) out[blockIdx.x] = smem[index];
}
}  // namespace
extern "C" int lds_peek(int index, int blocks, unsigned* out, void* stream) {
  const int bytes = 160 * 1024;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_peek), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return -2;
  hipLaunchKernelGGL(k_peek, dim3(blocks), dim3(64), bytes, static_cast<hipStream_t>(stream), index, out);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
