// chain2_micro.hip -- Stage B of VERDICT r4 item 1: the row-stationary / weight-streaming dataflow ("k_chain2") as a
// microkernel of layers 2 + 3 (split-f16 arithmetic, the default of k_chain) on random data, every CU busy.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg/chain2_micro.hip -o tools/dbg/_variants/chain2_micro
//   tools/dbg/_variants/chain2_micro [steps]
//
// One workgroup = 4 waves, one per SIMD.  A wave owns RT 16-row tiles (RT = 4: 64 rows, 256 per workgroup) and ALL 256
// hidden features: the ReLU'd, split layer-1 output of its rows (h1: 8 k-blocks x RT x (hi | lo) half pieces) stays in its
// registers as the B operands of layer 2; layer 2 is walked in 8 chunks of 32 output features whose accumulators, ReLU'd and
// split, ARE one k-block of layer 3's B operand (the permuted-k hand-over of k_chain, now without LDS or a barrier).  The
// weights stream L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction = one (tile, k-block, piece) block
// of the packed buffer) into a ring of three 22 KB slots; one phase = one slot = four k-blocks of one chunk (+ the three
// layer-3 blocks of the chunk before); ONE barrier per phase, placed in the middle of the phase so that the wave arrives with
// operands already in registers and leaves with MFMAs to issue.  Each A operand is read once per wave (ds_read_b128) and
// feeds 3 x RT MFMAs.
//
// Prints microseconds per (64 RT)-row tile-step of layers 2 + 3 and the maximum deviation from an fp64 host reference.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <string.h>

#include <type_traits>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float kSW = 1024.0f, kSX = 16.0f;   // the split-f16 scale factors of k_chain (weights, activations)
constexpr int kSlotBytes = 22 * 1024;
constexpr int kRing = 3;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

struct Args {
  const unsigned* w2;    // split-f16 A operands [16 T][8 kb][hi | lo][64 lanes][4 words]  (k_pack_a_split layout)
  const unsigned* w3;    // [3 j][8 kb][hi | lo][64][4]
  const float* b2;       // [256]
  const float* h1;       // (rows, 256) fp32, >= 0: the layer-1 output
  float* out;            // (rows, 48)  layer 3's output (rewritten every step)
  int steps;
  int rows;
};

// one 1 KB LDS-DMA piece: lane l moves 16 bytes from sbase + voff to LDS byte address lds_dst + 16 l.  Invisible to the
// compiler's wait-count pass by design (a tracked LDS-DMA puts s_waitcnt vmcnt(0) in front of the wave's next ds_read).
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 b = __builtin_bit_cast(i32x4, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = b[i] > 0 ? b[i] : 0;
  return __builtin_bit_cast(f32x4, b);
}

__device__ __forceinline__ void split8(const f32x4& u, const f32x4& v, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (_Float16)u[i];
    hi[4 + i] = (_Float16)v[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lo[i] = (_Float16)(u[i] - (float)hi[i]);
    lo[4 + i] = (_Float16)(v[i] - (float)hi[4 + i]);
  }
}

// Phase p of a tile-step (p = 2 c + half): slot layout [kq 4][t 2][hi | lo] 1 KB pieces of W2 (tiles 2c, 2c+1, k-blocks
// 4 half + kq), then for half == 0 the six pieces [j 3][hi | lo] of W3's k-block (c - 1) mod 8.
//
// The instruction stream is laid out BY HAND: one "slot" = one MFMA + the few other instructions that go into its shadow
// (LDS reads one k-block ahead, the ReLU + piece conversion of the chunk before at two vector instructions per slot, the
// LDS-DMA of the phase after next), every slot closed by sched_barrier(0).  hipcc still allocates registers, counts the
// LDS waits and pads hazards; it no longer chooses the order (left to it, the ~170 conversion instructions of a chunk sit
// in one block in front of the MFMAs they should hide under: sched_group_barrier pipelines were not honoured in a region of
// this size).
#define FENCE() __builtin_amdgcn_sched_barrier(0)
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  Every index below is a
// constant expression by construction (an unrolled run-time loop that the unroller gives up on puts the arrays it indexes in scratch)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// THIS COPY: NWV waves per workgroup (4 = one per SIMD as chain2_micro.hip; 8 = two per SIMD, which RT = 2 allows: 248 registers
// per wave).  Does a second wave on the SIMD hide the conversions and operand reads beside the first one's MFMAs?
template <int RT, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_micro(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, col = lane & 15;
  const long row0 = ((long)blockIdx.x * NWV + w) * (16 * RT);
  constexpr int NM = 6 * RT;          // MFMAs per k-block
  constexpr int NCONV = 32 * RT;      // conversion micro-steps per chunk (4 RT pairs x 8)

  // ---- h1 pieces of this wave's rows: B operands of layer 2, resident for the whole launch ----
  f16x8 bh[8][RT], bl[8][RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const float* hr = a.h1 + (row0 + 16 * rt + col) * 256;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(hr + 32 * kb + 4 * g);
      const f32x4 v = *reinterpret_cast<const f32x4*>(hr + 32 * kb + 16 + 4 * g);
      split8(u * kSX, v * kSX, bh[kb][rt], bl[kb][rt]);
      asm volatile("" : "+a"(bh[kb][rt]), "+a"(bl[kb][rt]));   // resident in the accumulation half of the register file
    }
  }
  float* b2s = reinterpret_cast<float*>(smem + kRing * kSlotBytes);   // [256] layer-2 bias x kSW kSX
  if (threadIdx.x < 256) b2s[threadIdx.x] = a.b2[threadIdx.x] * (kSW * kSX);

  const unsigned lane16 = (unsigned)lane * 16u;
  // piece k of this wave's DMA share of phase (c, half): W2 pieces qq = NWV k + w (k < 16 / NWV), then the W3 pieces
  constexpr int KW2 = 16 / NWV, KW3 = NWV == 4 ? 2 : 1, KALL = KW2 + KW3;
  auto issue_piece = [&](int c, int half, int k, unsigned slot_byte) {
    if (k < KW2) {
      // W2: piece index q = (kq*2 + t)*2 + hl  ->  source block ((2c + t)*8 + 4 half + kq)*2 + hl
      const int qq = NWV * k + w;
      const int kq = qq >> 2, t = (qq >> 1) & 1, hl = qq & 1;
      const long blk = (((long)(2 * c + t) * 8 + 4 * half + kq) * 2 + hl);
      dma16(a.w2 + blk * 256, lane16, slot_byte + (unsigned)qq * 1024u);
    } else {   // (branch-free: the waves beyond the sixth piece issue an earlier one a second time -- same bytes to the same place)
      const int cm = (c + 7) & 7;
      const int q0 = NWV * (k - KW2) + w;
      const int qq = q0 < 6 ? q0 : q0 - 4;
      const int j = qq >> 1, hl = qq & 1;
      const long blk = ((long)(j * 8 + cm) * 2 + hl);
      dma16(a.w3 + blk * 256, lane16, slot_byte + 16384u + (unsigned)qq * 1024u);
    }
  };

  // prologue: phases 0 and 1 of the first tile-step
#pragma unroll
  for (int k = 0; k < KALL; ++k) issue_piece(0, 0, k, 0u);
#pragma unroll
  for (int k = 0; k < KW2; ++k) issue_piece(0, 1, k, (unsigned)kSlotBytes);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned s_cur = 0u, s_nxt = (unsigned)kSlotBytes, s_nn = 2u * (unsigned)kSlotBytes;   // slots of phases p, p + 1, p + 2
  const u32x4* lbase = reinterpret_cast<const u32x4*>(smem) + lane;
  auto rdA = [&](unsigned slot_byte, int piece) { return __builtin_bit_cast(f16x8, lbase[(slot_byte >> 4) + piece * 64]); };
  auto rd_bias = [&](int c, int t) { return *reinterpret_cast<const f32x4*>(b2s + 16 * (2 * c + t) + 4 * g); };

  f32x4 acc3[3][RT];
  f32x4 accA[2][RT], accB[2][RT];     // layer-2 accumulators of the even / odd chunk
  unsigned phw[RT][4], plw[RT][4];    // pieces of the finished chunk (one k-block of layer 3), as packed words
  f16x8 ah[2], al[2], nh[2], nl[2];   // A operands of the current / the next k-block
  f16x8 w3h[2], w3l[2];               // layer 3's A operands, double buffered over j
  f32x4 bv[2];                        // initial accumulator value (bias) of the chunk that starts next
  f32x2 cm[2], cf[2];                 // conversion state of the two pairs in flight
  unsigned chw[2];

#pragma unroll
  for (int t = 0; t < 2; ++t) ah[t] = rdA(s_cur, t * 2), al[t] = rdA(s_cur, t * 2 + 1), bv[t] = rd_bias(0, t);

  // ReLU + split of the finished chunk's accumulators S into the packed pieces, as 8 single-instruction steps per pair of
  // values, two pairs in flight: step i = 16 grp + 2 stage + which
  auto conv_step = [&](auto& S, auto i_tag) {
    constexpr int i = decltype(i_tag)::value;
    constexpr int which = i & 1, stage = (i >> 1) & 7, p = 2 * (i >> 4) + which;   // pair p = 4 rt + q
    constexpr int rt = p >> 2, q = p & 3, t = q >> 1, e = (q & 1) * 2, word = 2 * t + (q & 1);
    if constexpr (stage == 0) {
      const float v0 = S[t][rt][e];   // (a copy: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0)
      cm[which][0] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v0), 0));
    } else if constexpr (stage == 1) {
      const float v1 = S[t][rt][e + 1];
      cm[which][1] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v1), 0));
    } else if constexpr (stage == 2) {
      cm[which] *= (1.0f / kSW);
    } else if constexpr (stage == 3) {
      chw[which] = __builtin_bit_cast(unsigned, __builtin_convertvector(cm[which], f16x2));
      phw[rt][word] = chw[which];
    } else if constexpr (stage == 4) {
      cf[which][0] = (float)__builtin_bit_cast(f16x2, chw[which])[0];
    } else if constexpr (stage == 5) {
      cf[which][1] = (float)__builtin_bit_cast(f16x2, chw[which])[1];
    } else if constexpr (stage == 6) {
      cf[which] = cm[which] - cf[which];
    } else {
      plw[rt][word] = __builtin_bit_cast(unsigned, __builtin_convertvector(cf[which], f16x2));
    }
  };
  auto pieces_h = [&](int rt) { return __builtin_bit_cast(f16x8, u32x4{phw[rt][0], phw[rt][1], phw[rt][2], phw[rt][3]}); };
  auto pieces_l = [&](int rt) { return __builtin_bit_cast(f16x8, u32x4{plw[rt][0], plw[rt][1], plw[rt][2], plw[rt][3]}); };

  // one k-block of layer 3: the pieces of a finished chunk against its W3 blocks (pieces 16..21 of `slot`; the first pair
  // w3h[0] / w3l[0] was read by the caller)
  auto layer3 = [&](unsigned slot) {
    static_for<9 * RT>([&](auto m_tag) {
      constexpr int m = decltype(m_tag)::value;
      constexpr int j = m / (3 * RT), pr = (m / RT) % 3, rt = m % RT;
      const f16x8 wa = pr == 1 ? w3l[j & 1] : w3h[j & 1];
      const f16x8 pb = pr == 2 ? pieces_l(rt) : pieces_h(rt);
      acc3[j][rt] = mfma(wa, pb, acc3[j][rt]);
      if constexpr (j < 2 && pr == 0 && rt == 0) w3h[(j + 1) & 1] = rdA(slot, 16 + (j + 1) * 2);
      if constexpr (j < 2 && pr == 0 && rt == 1) w3l[(j + 1) & 1] = rdA(slot, 16 + (j + 1) * 2 + 1);
      FENCE();
    });
  };

  // The four k-blocks of one phase on `D` (chunk c, half HALF).  CONV: the chunk before (accumulators S) is converted in
  // the shadow of this phase's MFMAs and its layer 3 follows the fourth k-block.  In the middle of the phase: its barrier,
  // then the DMA of the phase after next (chunk c_issue, same half).
  auto phase = [&](auto& D, auto& S, auto half_tag, auto conv_tag, int c_issue, int c_bias) {
    constexpr int HALF = decltype(half_tag)::value;
    constexpr bool CONV = decltype(conv_tag)::value;
    FENCE();
    static_for<4>([&](auto kq_tag) {
      constexpr int kq = decltype(kq_tag)::value;
      constexpr int kb = 4 * HALF + kq;
      static_for<NM>([&](auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
        constexpr int pr = m / (2 * RT), t = (m / RT) % 2, rt = m % RT;
        const f16x8 wa = pr == 1 ? al[t] : ah[t];
        const f16x8 xb = pr == 2 ? bl[kb][rt] : bh[kb][rt];
        if constexpr (HALF == 0 && kq == 0 && pr == 0) D[t][rt] = mfma(wa, xb, bv[t]);   // the chunk starts from its bias
        else D[t][rt] = mfma(wa, xb, D[t][rt]);
        // ---- the shadow of this MFMA ----
        if constexpr (m < 4) {   // A operands of the next k-block (kq == 3: of the next phase's first one; its slot was
                                 // published by the barrier in the middle of this phase)
          const unsigned sl = kq < 3 ? s_cur : s_nxt;
          constexpr int pc = (kq < 3 ? (kq + 1) * 4 : 0) + m;
          if constexpr (m & 1) nl[m >> 1] = rdA(sl, pc);
          else nh[m >> 1] = rdA(sl, pc);
        }
        if constexpr (CONV) {
          constexpr int i0 = 2 * (kq * NM + m);
          if constexpr (i0 < NCONV) conv_step(S, std::integral_constant<int, i0>{});
          if constexpr (i0 + 1 < NCONV) conv_step(S, std::integral_constant<int, i0 + 1>{});
        }
        {   // this wave's DMA share of the phase after next: one piece every DS slots of the third k-block
          constexpr int DS = (NM - 2) / 6;
          if constexpr (kq == 2 && m >= 2 && (m - 2) % DS == 0 && (m - 2) / DS < (HALF == 0 ? KALL : KW2))
            issue_piece(c_issue, HALF, (m - 2) / DS, s_nn);
        }
        if constexpr (HALF == 1 && kq == 3 && (m == 4 || m == 5)) bv[m - 4] = rd_bias(c_bias, m - 4);   // bias of the next chunk
        if constexpr (CONV && kq == 3 && m == 6) w3h[0] = rdA(s_cur, 16);
        if constexpr (CONV && kq == 3 && m == 7) w3l[0] = rdA(s_cur, 17);
        FENCE();
      });
      if constexpr (kq == 1) {
        // every wave's share of phase p + 1 has landed (issued a whole phase ago), every wave is done with phase p - 1
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        FENCE();
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) ah[t] = nh[t], al[t] = nl[t];
    });
    if constexpr (CONV) layer3(s_cur);
    const unsigned t_ = s_cur;
    s_cur = s_nxt, s_nxt = s_nn, s_nn = t_;
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;

  for (int step = 0; step < a.steps; ++step) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc3[j][rt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // a chunk = phase A (k-blocks 0..3; carries W3 of the chunk before) + phase B (k-blocks 4..7); in the middle of a phase
    // the DMA of the phase after next is issued: chunk c + 1, same half
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
    // tail: layer 3 of chunk 7.  Its W3 blocks sit in the slot of the NEXT tile-step's first phase (s_cur now): landed and
    // published by the barrier in the middle of the phase just finished.  (Nothing to hide the conversion under here; the
    // full kernel has the epilogue's own work beside it.)
    w3h[0] = rdA(s_cur, 16);
    w3l[0] = rdA(s_cur, 17);
    static_for<NCONV>([&](auto i_tag) { conv_step(accB, i_tag); });
    FENCE();
    layer3(s_cur);
    // stands for the epilogue's stores (x, candidates): layer 3's output of this step, 12 x 16 bytes per lane
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        *reinterpret_cast<f32x4*>(a.out + (row0 + 16 * rt + col) * 48 + 16 * j + 4 * g) = acc3[j][rt] * (1.0f / (kSW * kSX));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- host ------------------------------------------------------------------------------------------------------------
static unsigned short f16_bits(float x) {
  _Float16 h = (_Float16)x;
  unsigned short b;
  memcpy(&b, &h, 2);
  return b;
}
// k_pack_a_split<true> of csrc/mlp_kernels.hip, identity column map
static void pack_split(const std::vector<float>& W, int ld, int rows_valid, int n_tiles, int nkb, std::vector<unsigned>& dst) {
  dst.assign((size_t)n_tiles * nkb * 512, 0u);
  for (size_t i = 0; i < dst.size(); ++i) {
    const int lane = (int)((i >> 2) & 63), m = (int)((i & 3) | (((i >> 8) & 1) << 2));
    const long tk = (long)(i >> 9);
    const int kb = (int)(tk % nkb), Tt = (int)(tk / nkb);
    const int row = 16 * Tt + (lane & 15), g = lane >> 4;
    unsigned word = 0;
    for (int e = 0; e < 2; ++e) {
      const int s = 2 * (m & 3) + e;
      const int k = 32 * kb + 16 * (s >> 2) + 4 * g + (s & 3);
      float wv = row < rows_valid ? W[(size_t)row * ld + k] : 0.0f;
      wv *= kSW;
      const _Float16 hi = (_Float16)wv;
      const float piece = m < 4 ? (float)hi : wv - (float)hi;
      word |= (unsigned)f16_bits(piece) << (16 * e);
    }
    dst[i] = word;
  }
}

template <int RT, int NWV>
static void run(int steps, int n_wg) {
  const int rows = n_wg * NWV * 16 * RT;
  std::vector<float> W2(256 * 256), W3(48 * 256, 0.0f), b2(256), h1((size_t)rows * 256);
  srand(1234);
  auto rnd = [] { return (float)rand() / (float)RAND_MAX * 2.0f - 1.0f; };
  for (auto& v : W2) v = rnd() * 0.0625f;
  for (int o = 0; o < 40; ++o)
    for (int k = 0; k < 256; ++k) W3[o * 256 + k] = rnd() * 0.0625f;
  for (auto& v : b2) v = rnd() * 0.1f;
  for (auto& v : h1) { v = rnd(); v = v > 0.0f ? v : 0.0f; }
  std::vector<unsigned> p2, p3;
  pack_split(W2, 256, 256, 16, 8, p2);
  pack_split(W3, 256, 48, 3, 8, p3);
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
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_micro<RT, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  Args a{d2, d3, db2, dh1, dout, 2, rows};
  hipLaunchKernelGGL((k_micro<RT, NWV>), dim3(n_wg), dim3(64 * NWV), lds, 0, a);   // warm-up
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  a.steps = steps;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_micro<RT, NWV>), dim3(n_wg), dim3(64 * NWV), lds, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::vector<float> out((size_t)rows * 48);
  CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
  // fp64 reference on a sample of rows
  double maxerr = 0.0, maxref = 0.0, errj[3] = {0.0, 0.0, 0.0};
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
      const double got = out[(size_t)r * 48 + o];
      maxerr = fmax(maxerr, fabs(got - s));
      maxref = fmax(maxref, fabs(s));
      errj[o / 16] = fmax(errj[o / 16], fabs(got - s));
    }
  }
  const double rounds = (double)n_wg / 256.0;
  const double us = best * 1e3 / steps / (rounds < 1.0 ? 1.0 : rounds);
  const double mfmas = (1536.0 + 288.0) * RT / 4.0;   // per wave and tile-step
  printf("RT=%d, %d waves per workgroup (%d per SIMD)  %d workgroups x %d rows, %d steps: %.3f ms  ->  %.2f us per %d-row tile-step of layers 2+3 "
         "(%.2f us per 256 rows; %.2f ns per MFMA and wave)   max|err| %.3g (max|ref| %.3g)\n",
         RT, NWV, NWV / 4, n_wg, 16 * NWV * RT, steps, best, us, 16 * NWV * RT, us * 256.0 / (16 * NWV * RT), us * 1e3 / mfmas, maxerr, maxref);
  CHECK(hipFree(d2)); CHECK(hipFree(d3)); CHECK(hipFree(db2)); CHECK(hipFree(dh1)); CHECK(hipFree(dout));
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 200;
  run<4, 4>(steps, 256);
  run<2, 4>(steps, 256);
  run<2, 8>(steps, 256);
  run<2, 8>(steps, 512);
  run<4, 4>(steps, 512);
  return 0;
}
