// chain2_micro32.hip -- the microkernel of chain2_micro.hip (layers 2 + 3 of the row-stationary dataflow, split-f16
// arithmetic, weights streamed L2 -> LDS) on v_mfma_f32_32x32x16_f16 instead of 16x16x32: half the MFMA instructions for the
// same matrix-pipe time (an MFMA holds its SIMD's issue port for ~8 cycles whatever its shape: 8 of 16 -> 8 of 32), i.e. the
// wave that has to issue everything -- MFMAs, conversions, LDS reads, LDS-DMA -- gets 24 instead of 8 free issue cycles per
// MFMA.  (The full kernel with 16x16x32 measured issue-bound: ~38k issue cycles per tile-step against 35k of matrix pipe.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/dbg/chain2_micro32.hip -o tools/dbg/_variants/chain2_micro32
//
// A wave owns 64 rows = two 32-row tiles and all 256 hidden features.  Lane (c = lane & 31, h = lane >> 5) of a 32 x 32
// accumulator tile holds column (batch row) c and rows (features) (reg & 3) + 8 (reg >> 2) + 4 h; registers 8 s .. 8 s + 7,
// ReLU'd and split, are the B operand of k-step s (16 deep) of the next layer when its weights are packed with the k order
// 16 s + 8 (j >> 2) + 4 h + (j & 3) (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand").
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float kSW = 1024.0f, kSX = 16.0f;
constexpr int kSlotBytes = 24 * 1024;   // 16 W2 pieces + 8 W3 pieces
constexpr int kRing = 3;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
template <int I>
using Ic = std::integral_constant<int, I>;

struct Args {
  const unsigned* w2;    // [8 T][16 ks][hi | lo][64 lanes][4 words]
  const unsigned* w3;    // [2 T3][16 ks][hi | lo][64][4]
  const float* b2;       // [256]
  const float* h1;       // (rows, 256) fp32, >= 0
  float* out;            // (rows, 48)
  int steps;
  int rows;
};

__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

#ifndef MIXLO
#define MIXLO 1
#endif

// Phase (c, half): slot pieces [ksq 8][hi | lo] of W2 tile c, k-steps 8 half + ksq; for half == 0 also the eight pieces
// [T3 2][s 2][hi | lo] of W3's k-steps 2 (c - 1), 2 (c - 1) + 1 (mod 16) at piece 16.
__global__ __launch_bounds__(256) void k_micro32(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const long row0 = ((long)blockIdx.x * 4 + w) * 64;
  constexpr int NCONV = 128;   // conversion steps per chunk: 16 pairs x 8

  f16x8 bh[16][2], bl[16][2];   // h1 pieces per k-step and row tile
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const float* hr = a.h1 + (row0 + 32 * rt + c) * 256;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(hr + 16 * ks + 4 * h) * kSX;
      const f32x4 v = *reinterpret_cast<const f32x4*>(hr + 16 * ks + 8 + 4 * h) * kSX;
      f16x8 hi, lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) hi[i] = (_Float16)u[i], hi[4 + i] = (_Float16)v[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) lo[i] = (_Float16)(u[i] - (float)hi[i]), lo[4 + i] = (_Float16)(v[i] - (float)hi[4 + i]);
      bh[ks][rt] = hi, bl[ks][rt] = lo;
      asm volatile("" : "+a"(bh[ks][rt]), "+a"(bl[ks][rt]));
    }
  }
  float* b2s = reinterpret_cast<float*>(smem + kRing * kSlotBytes);
  b2s[threadIdx.x] = a.b2[threadIdx.x] * (kSW * kSX);

  const unsigned lane16 = (unsigned)lane * 16u;
  // piece k (0..5) of this wave's share of phase (cc, half): W2 pieces w, w + 4, w + 8, w + 12; then W3 pieces w, w + 4
  auto issue_piece = [&](int cc, int half, int k, unsigned slot_byte) {
    if (k < 4) {
      const int qq = 4 * k + w;                 // = ksq * 2 + hl
      const long blk = ((long)cc * 16 + 8 * half + (qq >> 1)) * 2 + (qq & 1);
      dma16(a.w2 + blk * 256, lane16, slot_byte + (unsigned)qq * 1024u);
    } else {
      const int qq = 4 * (k - 4) + w;           // = (T3 * 2 + s) * 2 + hl
      const int cm = (cc + 7) & 7;
      const long blk = ((long)(qq >> 2) * 16 + 2 * cm + ((qq >> 1) & 1)) * 2 + (qq & 1);
      dma16(a.w3 + blk * 256, lane16, slot_byte + 16384u + (unsigned)qq * 1024u);
    }
  };
  static_for<6>([&](auto k) { issue_piece(0, 0, decltype(k)::value, 0u); });
  static_for<4>([&](auto k) { issue_piece(0, 1, decltype(k)::value, (unsigned)kSlotBytes); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned s_cur = 0u, s_nxt = (unsigned)kSlotBytes, s_nn = 2u * (unsigned)kSlotBytes;
  const u32x4* lbase = reinterpret_cast<const u32x4*>(smem) + lane;
  auto rdA = [&](unsigned slot_byte, int piece) { return __builtin_bit_cast(f16x8, lbase[(slot_byte >> 4) + piece * 64]); };
  // the bias of chunk cc as an accumulator tile: features 32 cc + 8 u + 4 h .. + 3 in registers 4 u .. 4 u + 3
  auto rd_bias = [&](int cc, int u) { return *reinterpret_cast<const f32x4*>(b2s + 32 * cc + 8 * u + 4 * h); };

  f32x16 acc3[2][2];
  f32x16 accA[2], accB[2];
  unsigned pwh[2][2][4], pwl[2][2][4];   // pieces of the finished chunk: [k-step s][row tile][word]
  f16x8 ah, al, nh, nl;
  f16x8 w3h[2], w3l[2];
  f32x16 bv;
  f32x2 cm[2], cf[2];
  unsigned chw[2];

  ah = rdA(s_cur, 0), al = rdA(s_cur, 1);
  static_for<4>([&](auto u_) {
    constexpr int u = decltype(u_)::value;
    const f32x4 q = rd_bias(0, u);
    bv[4 * u] = q[0], bv[4 * u + 1] = q[1], bv[4 * u + 2] = q[2], bv[4 * u + 3] = q[3];
  });

  // ReLU + split of accumulators S[rt] (one chunk = two k-steps), 8 steps per pair of registers, two pairs in flight:
  // step i = 16 grp + 2 stage + which; pair p = 2 grp + which = 8 rt + 4 s + wd (registers 8 s + 2 wd, + 1 of tile rt)
  auto conv_step = [&](auto& S, auto i_tag) {
    constexpr int i = decltype(i_tag)::value;
    constexpr int which = i & 1, stage = (i >> 1) & 7, p = 2 * (i >> 4) + which;
    constexpr int rt = p >> 3, s = (p >> 2) & 1, wd = p & 3, r0 = 8 * s + 2 * wd;
    if constexpr (stage == 0) {
      const float v0 = S[rt][r0];
      cm[which][0] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v0), 0));
    } else if constexpr (stage == 1) {
      const float v1 = S[rt][r0 + 1];
      cm[which][1] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v1), 0));
    } else if constexpr (stage == 2) {
      if (!MIXLO) cm[which] *= (1.0f / kSW);
      else cf[which] = cm[which] * (1.0f / kSW);
    } else if constexpr (stage == 3) {
      chw[which] = __builtin_bit_cast(unsigned, __builtin_convertvector(MIXLO ? cf[which] : cm[which], f16x2));
      pwh[s][rt][wd] = chw[which];
    } else if constexpr (stage == 4) {
      if (!MIXLO) cf[which][0] = (float)__builtin_bit_cast(f16x2, chw[which])[0];
    } else if constexpr (stage == 5) {
      if (!MIXLO) cf[which][1] = (float)__builtin_bit_cast(f16x2, chw[which])[1];
    } else if constexpr (stage == 6) {
      if (!MIXLO) cf[which] = cm[which] - cf[which];
    } else {
      if (!MIXLO) {
        pwl[s][rt][wd] = __builtin_bit_cast(unsigned, __builtin_convertvector(cf[which], f16x2));
      } else {
        // lo = f16(relu(acc) / kSW - hi) as one fused multiply-add per half (v_fma_mixlo / mixhi_f16): the product is
        // exact (a power of two), so this is the same value as the separate subtraction
        const f16x2 hh = __builtin_bit_cast(f16x2, chw[which]);
        f16x2 ll;
        ll[0] = (_Float16)__builtin_fmaf(cm[which][0], 1.0f / kSW, -(float)hh[0]);
        ll[1] = (_Float16)__builtin_fmaf(cm[which][1], 1.0f / kSW, -(float)hh[1]);
        pwl[s][rt][wd] = __builtin_bit_cast(unsigned, ll);
      }
    }
  };
  auto pieces_h = [&](int s, int rt) { return __builtin_bit_cast(f16x8, u32x4{pwh[s][rt][0], pwh[s][rt][1], pwh[s][rt][2], pwh[s][rt][3]}); };
  auto pieces_l = [&](int s, int rt) { return __builtin_bit_cast(f16x8, u32x4{pwl[s][rt][0], pwl[s][rt][1], pwl[s][rt][2], pwl[s][rt][3]}); };

  // layer 3 of a finished chunk: 2 T3 x 2 k-steps x 3 products x 2 row tiles = 24 MFMAs; A operands from pieces 16 .. 23 of
  // `slot` ([T3][s][hi | lo]); the first pair (w3h[0], w3l[0]) was read by the caller
  auto layer3 = [&](unsigned slot) {
    static_for<24>([&](auto m_tag) {
      constexpr int m = decltype(m_tag)::value;
      constexpr int g4 = m / 6, T3 = g4 >> 1, s = g4 & 1, pr = (m % 6) >> 1, rt = m & 1;
      const f16x8 wa = pr == 1 ? w3l[g4 & 1] : w3h[g4 & 1];
      const f16x8 pb = pr == 2 ? pieces_l(s, rt) : pieces_h(s, rt);
      acc3[T3][rt] = mfma(wa, pb, acc3[T3][rt]);
      if constexpr (g4 < 3 && (m % 6) == 0) w3h[(g4 + 1) & 1] = rdA(slot, 16 + (g4 + 1) * 2);
      if constexpr (g4 < 3 && (m % 6) == 1) w3l[(g4 + 1) & 1] = rdA(slot, 16 + (g4 + 1) * 2 + 1);
      FENCE();
    });
  };

  // one phase: 8 k-steps x 6 MFMAs on D (chunk, half HALF); CONV: S is converted in its shadow, its layer 3 follows
  auto phase = [&](auto& D, auto& S, auto half_tag, auto conv_tag, int c_issue, int c_bias) {
    constexpr int HALF = decltype(half_tag)::value;
    constexpr bool CONV = decltype(conv_tag)::value;
    FENCE();
    static_for<8>([&](auto ksq_tag) {
      constexpr int ksq = decltype(ksq_tag)::value;
      constexpr int ks = 8 * HALF + ksq;
      static_for<6>([&](auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
        constexpr int pr = m >> 1, rt = m & 1;
        const f16x8 wa = pr == 1 ? al : ah;
        const f16x8 xb = pr == 2 ? bl[ks][rt] : bh[ks][rt];
        if constexpr (HALF == 0 && ksq == 0 && pr == 0) D[rt] = mfma(wa, xb, bv);
        else D[rt] = mfma(wa, xb, D[rt]);
        // ---- the shadow of this MFMA ----
        if constexpr (m < 2) {   // the next k-step's A operands (ksq == 7: the next phase's first)
          const unsigned sl = ksq < 7 ? s_cur : s_nxt;
          constexpr int pc = (ksq < 7 ? (ksq + 1) * 2 : 0) + m;
          if constexpr (m == 0) nh = rdA(sl, pc);
          else nl = rdA(sl, pc);
        }
        if constexpr (CONV) {
          constexpr int slot = ksq * 6 + m;
          static_for<3>([&](auto u) {
            constexpr int i = slot * 3 + decltype(u)::value;
            if constexpr (i < NCONV) conv_step(S, Ic<i>{});
          });
        }
        if constexpr (ksq >= 4 && m == 2 && ksq - 4 < 4) issue_piece(c_issue, HALF, ksq - 4, s_nn);
        if constexpr (HALF == 0 && ksq >= 4 && ksq < 6 && m == 4) issue_piece(c_issue, HALF, ksq, s_nn);
        if constexpr (HALF == 1 && ksq == 7 && m >= 2) {   // bias of the next chunk
          constexpr int u = m - 2;
          const f32x4 q = rd_bias(c_bias, u);
          bv[4 * u] = q[0], bv[4 * u + 1] = q[1], bv[4 * u + 2] = q[2], bv[4 * u + 3] = q[3];
        }
        if constexpr (CONV && ksq == 7 && m == 2) w3h[0] = rdA(s_cur, 16);
        if constexpr (CONV && ksq == 7 && m == 3) w3l[0] = rdA(s_cur, 17);
        FENCE();
      });
      if constexpr (ksq == 3) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        FENCE();
      }
      ah = nh, al = nl;
    });
    if constexpr (CONV) layer3(s_cur);
    const unsigned t_ = s_cur;
    s_cur = s_nxt, s_nxt = s_nn, s_nn = t_;
  };
  using H0 = Ic<0>;
  using H1 = Ic<1>;
  using Yes = std::true_type;
  using No = std::false_type;

  for (int step = 0; step < a.steps; ++step) {
#pragma unroll
    for (int T3 = 0; T3 < 2; ++T3)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[T3][rt][r] = 0.0f;
    phase(accA, accB, H0{}, No{}, 1, 0);
    phase(accA, accB, H1{}, No{}, 1, 1);
#pragma unroll 1
    for (int cc = 1; cc < 7; cc += 2) {
      phase(accB, accA, H0{}, Yes{}, cc + 1, 0);
      phase(accB, accA, H1{}, No{}, cc + 1, cc + 1);
      phase(accA, accB, H0{}, Yes{}, cc + 2, 0);
      phase(accA, accB, H1{}, No{}, cc + 2, cc + 2);
    }
    phase(accB, accA, H0{}, Yes{}, 0, 0);
    phase(accB, accA, H1{}, No{}, 0, 0);
    w3h[0] = rdA(s_cur, 16);
    w3l[0] = rdA(s_cur, 17);
    static_for<NCONV>([&](auto i_tag) { conv_step(accB, i_tag); });
    FENCE();
    layer3(s_cur);
    // stands for the epilogue's stores: features 32 T3 + 8 u + 4 h .. + 3 of row 32 rt + c
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int T3 = 0; T3 < 2; ++T3)
#pragma unroll
        for (int u = 0; u < (T3 == 0 ? 4 : 2); ++u) {
          const f32x4 v = f32x4{acc3[T3][rt][4 * u], acc3[T3][rt][4 * u + 1], acc3[T3][rt][4 * u + 2], acc3[T3][rt][4 * u + 3]} *
                          (1.0f / (kSW * kSX));
          *reinterpret_cast<f32x4*>(a.out + (row0 + 32 * rt + c) * 48 + 32 * T3 + 8 * u + 4 * h) = v;
        }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- host ------------------------------------------------------------------------------------------------------------
static unsigned short f16_bits(float x) {
  _Float16 hh = (_Float16)x;
  unsigned short b;
  memcpy(&b, &hh, 2);
  return b;
}
// A operands of v_mfma_f32_32x32x16_f16 in the permuted k order: piece (T, ks, hl): word m of lane l = elements j = 2m, 2m+1 =
// W[32 T + (l & 31)][16 ks + 8 (j >> 2) + 4 (l >> 5) + (j & 3)]
static void pack32(const std::vector<float>& W, int ld, int rows_valid, int n_tiles, int nks, std::vector<unsigned>& dst) {
  dst.assign((size_t)n_tiles * nks * 2 * 256, 0u);
  for (int T = 0; T < n_tiles; ++T)
    for (int ks = 0; ks < nks; ++ks)
      for (int hl = 0; hl < 2; ++hl)
        for (int l = 0; l < 64; ++l)
          for (int m = 0; m < 4; ++m) {
            unsigned word = 0;
            for (int e = 0; e < 2; ++e) {
              const int j = 2 * m + e, row = 32 * T + (l & 31), k = 16 * ks + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
              float wv = row < rows_valid ? W[(size_t)row * ld + k] : 0.0f;
              wv *= kSW;
              const _Float16 hi = (_Float16)wv;
              const float piece = hl == 0 ? (float)hi : wv - (float)hi;
              word |= (unsigned)f16_bits(piece) << (16 * e);
            }
            dst[((((size_t)T * nks + ks) * 2 + hl) * 64 + l) * 4 + m] = word;
          }
}

static void run(int steps, int n_wg) {
  const int rows = n_wg * 256;
  std::vector<float> W2(256 * 256), W3(64 * 256, 0.0f), b2(256), h1((size_t)rows * 256);
  srand(1234);
  auto rnd = [] { return (float)rand() / (float)RAND_MAX * 2.0f - 1.0f; };
  for (auto& v : W2) v = rnd() * 0.0625f;
  for (int o = 0; o < 40; ++o)
    for (int k = 0; k < 256; ++k) W3[o * 256 + k] = rnd() * 0.0625f;
  for (auto& v : b2) v = rnd() * 0.1f;
  for (auto& v : h1) { v = rnd(); v = v > 0.0f ? v : 0.0f; }
  std::vector<unsigned> p2, p3;
  pack32(W2, 256, 256, 8, 16, p2);
  pack32(W3, 256, 64, 2, 16, p3);
  unsigned *d2, *d3;
  float *db2, *dh1, *dout;
  CHECK(hipMalloc(&d2, p2.size() * 4));
  CHECK(hipMalloc(&d3, p3.size() * 4));
  CHECK(hipMalloc(&db2, 1024));
  CHECK(hipMalloc(&dh1, h1.size() * 4));
  CHECK(hipMalloc(&dout, (size_t)rows * 48 * 4));
  CHECK(hipMemcpy(d2, p2.data(), p2.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d3, p3.data(), p3.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db2, b2.data(), 1024, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dh1, h1.data(), h1.size() * 4, hipMemcpyHostToDevice));
  const size_t lds = kRing * kSlotBytes + 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_micro32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  Args a{d2, d3, db2, dh1, dout, 2, rows};
  hipLaunchKernelGGL(k_micro32, dim3(n_wg), dim3(256), lds, 0, a);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  a.steps = steps;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_micro32, dim3(n_wg), dim3(256), lds, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::vector<float> out((size_t)rows * 48);
  CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0.0, maxref = 0.0;
  for (int r = 0; r < rows; r += rows / 97 + 1) {
    double h2[256];
    for (int o = 0; o < 256; ++o) {
      double s = b2[o];
      for (int k = 0; k < 256; ++k) s += (double)W2[o * 256 + k] * h1[(size_t)r * 256 + k];
      h2[o] = s > 0.0 ? s : 0.0;
    }
    for (int o = 0; o < 48; ++o) {
      double s = 0.0;
      for (int k = 0; k < 256; ++k) s += (double)W3[o * 256 + k] * h2[k];
      maxerr = fmax(maxerr, fabs((double)out[(size_t)r * 48 + o] - s));
      maxref = fmax(maxref, fabs(s));
    }
  }
  printf("32x32x16: %d workgroups x 256 rows, %d steps: %.3f ms -> %.2f us per 256-row tile-step of layers 2+3 per round   max|err| %.3g (max|ref| %.3g)\n",
         n_wg, steps, best, best * 1e3 / steps / (n_wg < 256 ? 1.0 : n_wg / 256.0), maxerr, maxref);
  CHECK(hipFree(d2)); CHECK(hipFree(d3)); CHECK(hipFree(db2)); CHECK(hipFree(dh1)); CHECK(hipFree(dout));
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 200;
  if (steps < 10) {
    run(steps, 4);
    return 0;
  }
  run(steps, 256);
  run(steps, 512);
  return 0;
}
