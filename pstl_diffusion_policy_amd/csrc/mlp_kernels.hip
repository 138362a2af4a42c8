// mlp_kernels.hip -- the denoiser / RefineNet MLP chain on gfx950 MFMA (split-f16 by default: fp32-faithful; split-bf16
// or exact fp32 MFMA on request), the scene encoder, merge_net (forward + backward) and weight packing.
//
// Design:
//  * Both 3-layer MLPs of the hot path (policy_net 303->256->256->40, rect_net 271->256->256->40) see only 47
//    per-row input columns that change (x/fused 40 + highlevel 1 + stlp 6); the other 224 (+32 timestep) columns are
//    constant per scene (per step) and enter as the INITIAL VALUE of the layer-1 accumulator (base[scene] + tbias[t]).
//  * Weight-stationary: the 344 KB of per-step weights live in the REGISTER FILE of one workgroup (8 waves x 176
//    VGPRs, or 4 x 352) as MFMA A-operands for the whole launch.  Activations are the B operand
//    (v_mfma_f32_16x16x4_f32: D[f][row] += W[f][k] * X[k][row]), 16 rows per tile.
//  * The accumulator layout of one layer IS the B-operand layout of the next one if the k index is permuted
//    (lane group g, register r  <->  k = 16q + 4g + r), and the permutation is baked into the packed weights.  So layer
//    2's output feeds layer 3 straight from registers, and layer 1 -> 2 crosses waves through LDS as plain
//    lane-linear 16-byte reads/writes (conflict-free).
//  * A workgroup owns G tiles (192 rows = one scene at S = 64) for ALL reverse steps of a launch: x never leaves
//    LDS between steps; HBM traffic is the noise read (parity mode) and the emitted candidates.
//  * f32 MFMA is bit-for-bit a k-ordered fmaf chain, so results match an fp32 torch path to rounding (1e-4 gate).
//  * Default arithmetic (template parameter PT = 2): every fp32 operand is two IEEE-half pieces of a power-of-two
//    multiple of the value (2^-23 per operand), every product three v_mfma_f32_16x16x32_f16 products accumulated in fp32
//    -- 1.9e-6 from the reference after 99 chained steps, the same as the fp32 MFMA form, at 2.8x its speed.  PT = 1:
//    bf16 pieces (2^-17 per operand, 8e-6).  See the comment above k_chain.
#include <stdlib.h>

#include <type_traits>

#include "chain_args.hpp"
#include "pstl_common.hpp"
#include "rng.hpp"


namespace pstl {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kHid = PSTL_HID;     // 256
constexpr int kFeat = PSTL_FEAT;   // 224
constexpr int kCtrl = PSTL_CTRL;   // 40
constexpr int kKx = 48;            // per-row input columns of layer 1 (40 + 1 + 6, padded to 48)
constexpr int kTileRows = 16;
constexpr int kG = 12;             // tiles per workgroup (upper bound; small batches use fewer, see tiles_per_group)

// ---- packed weight buffer (float offsets) -----------------------------------------------------------------------
struct EncOff {        // one scene encoder (in -> 256 -> 256 -> 32), weights as fp32 MFMA A operands (k_pack_a layout)
  long a0;             // [16 T][enc_k16(e) q][4 r][64 lanes]   layer 0, input columns padded to 16 / 16 / 48
  long b0;             // [256]
  long a1;             // [16 T][16 q][4 r][64]
  long b1;             // [256]
  long a2;             // [2 T][16 q][4 r][64]
  long b2;             // [32]
};
struct MergeOff {
  long w0t, b0, w1t, b1, w2t, b2;  // [40][32],[32],[32][32],[32],[32][40],[40]
};
struct PackLayout {
  EncOff enc[3];
  ChainOff pol, rect;
  MergeOff mrg;
  long status;   // 16 words, see pstl_packed_status_offset() in the header
  long total;
};

__host__ __device__ constexpr int enc_in(int e) { return e == 0 ? 6 : e == 1 ? 7 : 45; }
__host__ __device__ constexpr int enc_k16(int e) { return e == 2 ? 3 : 1; }   // 16-column blocks of the padded token input

__host__ __device__ inline PackLayout make_layout() {
  PackLayout L;
  long o = 0;
  for (int e = 0; e < 3; ++e) {
    L.enc[e].a0 = o; o += 16L * enc_k16(e) * 4 * 64;
    L.enc[e].b0 = o; o += kHid;
    L.enc[e].a1 = o; o += 16L * 16 * 4 * 64;
    L.enc[e].b1 = o; o += kHid;
    L.enc[e].a2 = o; o += 2L * 16 * 4 * 64;
    L.enc[e].b2 = o; o += 32;
  }
  ChainOff* cs[2] = {&L.pol, &L.rect};
  for (int c = 0; c < 2; ++c) {
    cs[c]->w1f = o; o += (long)kFeat * kHid;
    cs[c]->b1 = o;  o += kHid;
    cs[c]->w1t = o; o += 32L * kHid;
    cs[c]->w1x = o; o += 16L * 3 * 4 * 64;
    cs[c]->w2 = o;  o += 16L * 16 * 4 * 64;
    cs[c]->b2 = o;  o += kHid;
    cs[c]->w3 = o;  o += 3L * 16 * 4 * 64;
    cs[c]->b3 = o;  o += 48;
    cs[c]->w1xb = o; o += 16L * 2 * 8 * 64;
    cs[c]->w2b = o;  o += 16L * 8 * 8 * 64;
    cs[c]->w3b = o;  o += 3L * 8 * 8 * 64;
    cs[c]->w1xh = o; o += 16L * 2 * 8 * 64;
    cs[c]->w2h = o;  o += 16L * 8 * 8 * 64;
    cs[c]->w3h = o;  o += 3L * 8 * 8 * 64;
  }
  L.mrg.w0t = o; o += 40 * 32;
  L.mrg.b0 = o;  o += 32;
  L.mrg.w1t = o; o += 32 * 32;
  L.mrg.b1 = o;  o += 32;
  L.mrg.w2t = o; o += 32 * 40;
  L.mrg.b2 = o;  o += 40;
  o = (o + 15) / 16 * 16;
  L.status = o;  o += 16;
  L.total = (o + 63) / 64 * 64;
  return L;
}

// ---- packing ----------------------------------------------------------------------------------------------------
// dst[k][o] = W[o][k]
__global__ void k_transpose(const float* W, int out, int in, int col0, int ncol, float* dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)ncol * out) return;
  const int k = (int)(i / out), o = (int)(i % out);
  dst[i] = W[(long)o * in + col0 + k];
}

__global__ void k_copy(const float* src, int n, int npad, float* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) dst[i] = i < n ? src[i] : 0.0f;
}

// A-operand layout: dst[((Tt*nq + q)*4 + r)*64 + lane] = W[16*Tt + (lane&15)][colmap(16q + 4(lane>>4) + r)]
// mode 0: identity columns; mode 1: policy K-ext (x 224.., hl 296, stlp 297..302); mode 2: rect K-ext
// (fused 231.., hl 224, stlp 225..230)
__global__ void k_pack_a(const float* W, int ld, int rows_valid, int n_tiles, int nq, int mode, float* dst,
                         int cols_valid = 1 << 30) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n_tiles * nq * 256) return;
  const int lane = (int)(i & 63), r = (int)((i >> 6) & 3);
  const long tq = i >> 8;
  const int q = (int)(tq % nq), Tt = (int)(tq / nq);
  const int row = 16 * Tt + (lane & 15);
  const int k = 16 * q + 4 * (lane >> 4) + r;
  int col = k;
  if (mode == 1) col = k < 40 ? 224 + k : k == 40 ? 296 : k < 47 ? 297 + (k - 41) : -1;
  if (mode == 2) col = k < 40 ? 231 + k : k == 40 ? 224 : k < 47 ? 225 + (k - 41) : -1;
  dst[i] = (row < rows_valid && col >= 0 && col < cols_valid) ? W[(long)row * ld + col] : 0.0f;
}

// Split A-operand layout (two 16-bit pieces per fp32 weight).  F16 = false: hi = bf16(w), lo = bf16(w - hi) (w = hi + lo to
// ~2^-17 relative).  F16 = true (the default arithmetic of the chains): the pieces are IEEE half, of ws = kSplitW * w:
// hi = f16(ws), lo = f16(ws - hi), so that ws = hi + lo to ~2^-23 relative -- an fp32 operand to within one bit.  The
// power-of-two factor keeps the pieces of ordinary weights (|w| ~ 1e-3 .. 1) inside half's normal range (|ws| >= 0.25
// leaves lo normal; below that the error is 2^-25 / kSplitW absolute); it is exact and is divided out of the
// accumulators by the kernel.  |w| must stay below 65504 / kSplitW = 64.
// Word m of lane l for block blk = Tt*nkb + kb (tile Tt, k-block kb) sits at
// dst[((blk*2 + (m>>2))*64 + l)*4 + (m&3)]: one 16-byte load per lane fetches the four hi words (m < 4: slots 2m, 2m+1,
// low half first), a second one the four lo words (m >= 4).
// Slot s of lane group g = l>>4 is column k = 32 kb + 16 (s>>2) + 4 g + (s&3): with this order the accumulator
// registers of two neighbouring 16-feature tiles ARE one k-block of the next layer's B operand (see k_chain).
__device__ __forceinline__ unsigned bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ unsigned f16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }

// wmax (F16 only): running maximum of |w| over the packed weights, as the bit pattern of a non-negative float (monotone in
// the value; a NaN weight ranks above everything) -- the domain check of the split-f16 arithmetic, read by the host.
template <bool F16>
__global__ void k_pack_a_split(const float* W, int ld, int rows_valid, int n_tiles, int nkb, int mode, unsigned* dst,
                               unsigned* wmax = nullptr) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n_tiles * nkb * 512) return;
  const int lane = (int)((i >> 2) & 63), m = (int)((i & 3) | (((i >> 8) & 1) << 2));
  const long tk = i >> 9;
  const int kb = (int)(tk % nkb), Tt = (int)(tk / nkb);
  const int row = 16 * Tt + (lane & 15), g = lane >> 4;
  unsigned word = 0;
  for (int e = 0; e < 2; ++e) {
    const int s = 2 * (m & 3) + e;
    const int k = 32 * kb + 16 * (s >> 2) + 4 * g + (s & 3);
    int col = k;
    if (mode == 1) col = k < 40 ? 224 + k : k == 40 ? 296 : k < 47 ? 297 + (k - 41) : -1;
    if (mode == 2) col = k < 40 ? 231 + k : k == 40 ? 224 : k < 47 ? 225 + (k - 41) : -1;
    float wv = (row < rows_valid && col >= 0) ? W[(long)row * ld + col] : 0.0f;
    if (F16) {
      if (wmax) {
        const unsigned bits = __builtin_bit_cast(unsigned, wv) & 0x7fffffffu;
        if (bits > *wmax) atomicMax(wmax, bits);
      }
      wv *= kSplitW;
      const _Float16 hi = (_Float16)wv;
      const float piece = m < 4 ? (float)hi : wv - (float)hi;
      word |= f16_bits(piece) << (16 * e);
    } else {
      const __bf16 hi = (__bf16)wv;
      const float piece = m < 4 ? (float)hi : wv - (float)hi;
      word |= bf16_bits(piece) << (16 * e);
    }
  }
  dst[i] = word;
}

// ---- timestep bias ----------------------------------------------------------------------------------------------
// tbias[t][h] = sum_k W1[h][264+k] * pe(t)[k], pe(t) = [sin(t f_j) | cos(t f_j)], f_j = 1/10000^(2j/32)
__global__ void k_time_bias(const float* w1t /* [32][256] */, int steps, float* tbias) {
  const int t = blockIdx.x, h = threadIdx.x;
  __shared__ float pe[32];
  if (h < 16) {
    const float inv = 1.0f / powf(10000.0f, (float)(2 * h) / 32.0f);
    const float arg = (float)t * inv;
    pe[h] = sinf(arg);
    pe[16 + h] = cosf(arg);
  }
  __syncthreads();
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 32; ++k) acc += w1t[k * kHid + h] * pe[k];
  tbias[(long)t * kHid + h] = acc;
}

// ---- the MLP chain kernel ---------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf(f16x8 a, f16x8 b, f32x4 c) {
#ifdef PSTL_ABL_NO_MFMA      // timing-only ablation (tools/dbg): the instruction stream without its matrix instructions
  asm volatile("" : "+v"(c) : "v"(a), "v"(b));
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// eight fp32 values -> their 16-bit hi pieces and the 16-bit rounding of what the hi pieces miss
// (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, RNE)
template <typename PV>
__device__ __forceinline__ void split8(const f32x4& u, const f32x4& v, PV& hi, PV& lo) {
  typedef std::remove_reference_t<decltype(hi[0])> E;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (E)u[i];
    hi[4 + i] = (E)v[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lo[i] = (E)(u[i] - (float)hi[i]);
    lo[4 + i] = (E)(v[i] - (float)hi[4 + i]);
  }
}


// Domain guard of the split-f16 arithmetic: the running maximum, per 16-bit half, of the hi pieces' bit patterns
// (v_pk_max_u16).  A layer input that left the half range shows up as the pattern of infinity (0x7c00) in its hi piece at
// the very conversion that overflowed -- before any NaN exists, so the test does not depend on NaNs surviving the ReLUs
// (max(NaN, 0) is 0, and so is the integer form used here for a NaN with the sign bit set).  `mask_sign`: the values are
// signed (the layer-1 input); hidden activations are >= 0 after the ReLU.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <typename PV>
__device__ __forceinline__ void note_pieces(unsigned& ovf, const PV& hi, bool mask_sign) {
  const u32x4 w = __builtin_bit_cast(u32x4, hi);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned v = mask_sign ? (w[r] & 0x7fff7fffu) : w[r];
    ovf = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, ovf), __builtin_bit_cast(u16x2, v)));
  }
}
__device__ __forceinline__ bool pieces_overflowed(unsigned ovf) { return (ovf & 0xffffu) >= 0x7c00u || (ovf >> 16) >= 0x7c00u; }

// Timing-only ablations of the fused tile-step loop (tools/dbg/build_variants.sh + time_variants.py; results are garbage):
// PSTL_ABL_SKIP bits: 1 epilogue, 2 noise, 4 the ReLU + half-piece conversions, 8 layer 2's B-operand reads (one k-block is
// read, the rest reuse it), 16 the input split (waves 3 / 7), 32 layer 1's constant rows, 64 the LDS writes (h1 pieces,
// partial sums), 128 / 256 only the conversion in front of layer 3 / only the one of layer 1's output (k-block 5).  PSTL_ABL_NO_MFMA / PSTL_ABL_NO_BARRIER: see mfma_bf and the end of the loop.
#ifndef PSTL_ABL_SKIP
#define PSTL_ABL_SKIP 0
#endif


// A value the optimiser must take as it comes at this point of the loop: stops it from hoisting `uniform pointer +
// lane offset` out of the tile-step loop as a per-lane 64-bit pointer (five of those were live across the loop, spilled,
// and reloaded in the epilogue behind a full vmcnt wait)
__device__ __forceinline__ unsigned here(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

// max(v, 0) as a signed-integer max on the bit patterns: one instruction per value (fmaxf costs two: the compiler first
// canonicalises an operand it cannot prove free of signalling NaNs); identical to fmaxf for every non-NaN input
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 b = __builtin_bit_cast(i32x4, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = b[i] > 0 ? b[i] : 0;
  return __builtin_bit_cast(f32x4, b);
}

// torch.clip semantics: a NaN stays a NaN (fminf / fmaxf would return the bound and hide an operand that left the
// split-f16 domain behind a plausible control); identical to fminf(fmaxf(v, -m), m) for every other input
__device__ __forceinline__ float clip_keep_nan(float v, float m) { return v < -m ? -m : (v > m ? m : v); }

// LDS address (in floats) of activation element k (0..47) of tile column c in the B-operand image [q][lane][r]
__device__ __forceinline__ int xs_addr(int k, int c) { return (((k >> 4) * 64) + (((k >> 2) & 3) * 16 + c)) * 4 + (k & 3); }

// ABL != 0 are timing-only diagnostic builds (selected with cfg->chain_waves = 8 | 16 + 100*ABL, used only to attribute
// the kernel's time): 1 = no epilogue and no noise (wrong results by construction), 7 = the full kernel with s_memtime
// stamps of workgroup 7, iterations 64..95, every wave, written as 64-bit cycle counts to the buffer passed as emit_out
// (n_emit must be 0): [it-64][wave][slot] (tools/dbg/chain_stamps.py names the slots).
//
// PT: 0 = fp32 MFMA; 1, 2 = the three layers on v_mfma_f32_16x16x32_{bf16,f16} with every fp32 operand split into two
// 16-bit pieces (x = hi + lo): W.X ~ Whi.Xhi + Wlo.Xhi + Whi.Xlo, three products accumulated in fp32.  One k-block of 32
// = the 2 x 16 features one wave produces, so the register-to-register hand-over between layers of the fp32 kernel
// carries over: lane (g, c) holds features 16 ot + 4 g + r of row c in acc[ot][r], which is slot s = 4 ot + r of its B
// operand.  PT = 2 (half pieces of 2^10 w and 2^4 x, see k_pack_a_split): an operand to 2^-23, 1.9e-6 from the reference
// after 99 chained steps like the fp32 kernel; PT = 1 (bfloat16 pieces): 2^-17, 8e-6.
// SAVE (REFINE, split forms): the training forward pass -- the hidden layers' fp32 outputs go to a.h1_save / a.h2_save (a
// template parameter, not a run-time test: the test alone cost the inference launch 6 %).
// PERSIST: one workgroup per CU walks the 12-tile groups blockIdx.x, blockIdx.x + gridDim.x, ... with the weights loaded
// into registers once (used for the single-step launches of the guided phase, where the 344 KB weight fetch and the
// workgroup turnover are ~10 % of a 12-iteration workgroup).
template <int NW, bool REFINE, int ABL = 0, bool UT = false, int PT = 0, bool PERSIST = false, bool SAVE = false,
          bool SPARSE = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void k_chain(ChainArgs a) {
#ifdef PSTL_WG_TIMES   // diagnostic build (tools/dbg/wg_finish_times.py): when and where every workgroup started and ended
  unsigned long long wg_t0;
  unsigned wg_hw, wg_xcc;
  asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
               : "=s"(wg_t0), "=s"(wg_hw), "=s"(wg_xcc));
#endif
  constexpr bool BF = PT != 0;      // the operands are split into two 16-bit pieces
  constexpr bool F16 = PT == 2;     // ... of IEEE half (scaled, see k_pack_a_split); PT == 1: bfloat16 pieces
  typedef std::conditional_t<F16, f16x8, bf16x8> pv8;
  // F16: layer inputs are held as kSX * x, weights as kSW * w, so every accumulator is kSW * kSX times its value; a
  // hidden layer's output relu(acc) / kSW is the next layer's kSX-scaled input.  All factors are powers of two (exact).
  constexpr float kSX = F16 ? kSplitX : 1.0f, kSW = F16 ? kSplitW : 1.0f;
  constexpr float kAcc = kSX * kSW, kInvSW = 1.0f / kSW, kInvAcc = 1.0f / kAcc, kInvSX = 1.0f / kSX;
  static_assert(!BF || NW == 8, "the split variants are 8-wave kernels");
  constexpr int OT = 16 / NW;       // 16-feature output tiles per wave
  constexpr int NT = NW * 64;
  // The first NCW waves also run the epilogue (8 waves: waves 0..3, the older wave of each SIMD pair; measured 1.8 %
  // faster than giving it to waves 4..7, and static s_setprio for either half changed nothing measurable).
  constexpr bool EPI_FIRST = true;
  constexpr int NCW = NW == 4 ? 3 : NW / 2;
  constexpr int NCT = NCW * 64;     // >= 160 epilogue threads are needed (4 outputs each)
  // wave that issues the direct-to-LDS loads of the constant rows: with 8 waves, wave 3 belongs to the epilogue group
  // but owns no epilogue rows (160 threads = 2.5 waves), so it has the slack
  constexpr int kStager = (EPI_FIRST && NCT - 160 >= 64) ? NCW - 1 : NW - 1;
  static_assert(NCT >= 160, "epilogue needs 160 threads");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                          // [kG][3][64][4]      activations of layer 1, B-operand order
  float* h1 = xs + kG * 768;                // [3][16][64][4]      layer-1 output (triple buffered: written two ahead)
  float* part = h1 + 3 * 16 * 256;          // [2][NW][3][64][4]   layer-3 partial sums (double buffered)
  float* b2s = part + 2 * NW * 768;         // [256] layer-2 bias, [48] layer-3 bias
  float* b3s = b2s + 256;
  float* coef = b3s + 48;                   // [kMaxLaunchSteps][4] c1, 1/sqrt(alpha), sqrt(beta) of step s_hi - n
  float* crow = coef + 4 * kMaxLaunchSteps; // [3][2][256] UT only: base[scene] row and tbias[step] row of a tile-step
  // BF: waves 4..6 (the SIMD partners of the epilogue waves 0..2) draw the noise and hand it over through LDS.  The
  // split-bf16 kernel is bound by the instructions its longest wave has to issue, not by the matrix pipe, so the ~130
  // instructions of Philox + Box-Muller are taken off the epilogue waves.  (In the fp32 kernel the same move was slower.)
  constexpr bool NOISE_SPLIT = BF;
  f32x4* zbuf = reinterpret_cast<f32x4*>(crow + 3 * 512);   // [2][192] noise quads of a tile-step (160 used)
  // split forms: [2][kb 2][hi | lo][64 lanes] B operands of layer 1 (pieces of a tile's input image), made by waves 7 / 3
  // one iteration before layer 1 of that tile-step reads them (XONCE: all eight waves used to split the same 48 x 16 values)
  u32x4* xpb = reinterpret_cast<u32x4*>(zbuf + 2 * 192);
  constexpr bool XONCE = BF;
#ifndef PSTL_ABL_NO_EMPTY_SLOT_FIX   // (tools/dbg: the state before the fix, to show that tests/ldspoison catches it)
  if constexpr (SPARSE) {
    // the layer 1 woven into an empty pipeline slot reads the piece buffer without anyone having written it for that slot
    // (with a single tile per workgroup: never): defined contents instead of whatever the LDS held.  (The first writer, the
    // prologue's split_x, comes after the next barrier; the constant-row ring is filled by the prologue for all three slots.)
    for (int i = threadIdx.x; i < 2 * 4 * 64; i += blockDim.x) xpb[i] = u32x4{0u, 0u, 0u, 0u};
  }
#endif
  // (declaring the wave index uniform -- readfirstlane -- turns the role branches into scalar branches and was measured
  // 30 % slower: that form of the loop spills inside it)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const long n_tiles = (a.N + kTileRows - 1) / kTileRows;
  const long n_groups = (n_tiles + a.tiles_per_group - 1) / a.tiles_per_group;

  // ---- weights -> registers (A operands), once per launch ----
  float w1x[BF ? 1 : OT][12], w2[BF ? 1 : OT][64], w3[3][BF ? 1 : OT][4];
  pv8 w1h[OT][2], w1l[OT][2], w2h[OT][8], w2l[OT][8], w3h[3], w3l[3];
  {
    const float* p1 = a.packed + a.off.w1x;
    const float* p2 = a.packed + a.off.w2;
    const float* p3 = a.packed + a.off.w3;
    if constexpr (BF) {
      const unsigned* q1 = reinterpret_cast<const unsigned*>(a.packed + (F16 ? a.off.w1xh : a.off.w1xb));
      const unsigned* q2 = reinterpret_cast<const unsigned*>(a.packed + (F16 ? a.off.w2h : a.off.w2b));
      const unsigned* q3 = reinterpret_cast<const unsigned*>(a.packed + (F16 ? a.off.w3h : a.off.w3b));
      auto load_pair = [&](const unsigned* q, long blk, pv8& hi, pv8& lo) {
        const u32x4* q4 = reinterpret_cast<const u32x4*>(q);
        hi = __builtin_bit_cast(pv8, q4[(blk * 2 + 0) * 64 + lane]);
        lo = __builtin_bit_cast(pv8, q4[(blk * 2 + 1) * 64 + lane]);
      };
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int T = w * OT + ot;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) load_pair(q1, (long)T * 2 + kb, w1h[ot][kb], w1l[ot][kb]);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) load_pair(q2, (long)T * 8 + kb, w2h[ot][kb], w2l[ot][kb]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) load_pair(q3, (long)j * 8 + w, w3h[j], w3l[j]);
    } else
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const int T = w * OT + ot;
#pragma unroll
      for (int m = 0; m < 12; ++m) w1x[BF ? 0 : ot][m] = p1[((long)T * 12 + m) * 64 + lane];
#pragma unroll
      for (int m = 0; m < 64; ++m) w2[BF ? 0 : ot][m] = p2[((long)T * 64 + m) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) w3[j][BF ? 0 : ot][r] = p3[(((long)j * 16 + T) * 4 + r) * 64 + lane];
    }
    if (tid < 256) b2s[tid] = a.packed[a.off.b2 + tid] * kAcc;
    if (tid < 48) b3s[tid] = a.packed[a.off.b3 + tid];
    if (!REFINE && tid <= a.step_hi - a.step_lo) {   // reverse-step coefficients (nusc_train.py:580-587), once per launch
      const int i = a.step_hi - tid;
      const float al = a.alpha[i], ah = a.alpha_hat[i], be = a.beta[i];
      coef[4 * tid + 0] = (1.0f - al) / sqrtf(1.0f - ah);
      coef[4 * tid + 1] = 1.0f / sqrtf(al);
      coef[4 * tid + 2] = sqrtf(be);
      coef[4 * tid + 3] = 0.0f;
    }
  }

  const unsigned long long seed = a.seed_dev ? uniform_u64(a.seed_dev) : a.seed;
  // split-f16 domain guard (see note_pieces): per-lane maximum of the hi pieces this lane made; tested once after the loop
  unsigned ovf = 0;
  if (F16 && tid == 0) {   // the weights themselves: max |w| as the packer recorded it (status words 0 = policy_net, 1 = rect_net)
    const float wm = reinterpret_cast<const float*>(a.status)[REFINE ? -1 : -2];
    if (!(wm < PSTL_SPLIT_F16_WMAX)) atomicOr(a.status, 1u);
  }

  // CONT (single-step launches of the split kernels, PERSIST): the workgroup's groups blockIdx.x, blockIdx.x + gridDim.x,
  // ... are ONE stream of tile-steps through the software pipeline -- no drain, prologue and refill between groups (7 of
  // the 31 us a group took).  A position's `n` then counts the rounds (groups), not reverse steps; the image slot of a
  // finished tile is refilled with the same tile of the next round by LDS-DMA (stage_x) two iterations after its
  // epilogue, >= 3 iterations before the splitter waves read it (tiles_per_group >= 8).
  const bool cont = PERSIST && XONCE && !REFINE && a.tiles_per_group >= 8 &&
                    !(a.n_emit >= a.steps && a.step_hi == a.steps - 1);
  for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {   // one pass unless PERSIST (gridDim.x == n_groups)
  if (PERSIST && grp != blockIdx.x) __syncthreads();   // the previous group's last epilogue has left xs / part
  const long tile0 = grp * a.tiles_per_group;
  auto t0 = [&](int n) { return cont ? ((long)blockIdx.x + (long)n * gridDim.x) * a.tiles_per_group : tile0; };
  int G = (int)((n_tiles - tile0) < a.tiles_per_group ? (n_tiles - tile0) : a.tiles_per_group);
  // SPARSE (latency layout for small batches, tiles_per_group 1..5): the workgroup owns Gr <= 5 tiles; the other slots of the
  // five-deep software pipeline are EMPTY rather than phantom tiles: an iteration does only the stages whose tile exists
  // (nothing at all -- just the barrier -- when neither layer 1's nor layer 2's position is a tile).  Same arithmetic per
  // row, so the same bits as the throughput layout.
  const int Gr = G;
  auto real = [&](int tl) { return !SPARSE || tl < Gr; };
  // Layer 1 runs two tile-steps ahead and the epilogue one behind, so >= 4 tiles must be in flight (see the hazard
  // notes below); short tail blocks process phantom tiles whose rows are clamped on load and masked on store.
  // (XONCE: the pieces of a tile's image are made one more iteration ahead: >= 5.)
  if (G < (XONCE ? 5 : 4)) G = XONCE ? 5 : 4;
  // (SPARSE with a single tile: its layer 1 always runs stand-alone -- layer 2's slot is empty then -- and splits its own
  // input, so neither the pieces nor layer 1 itself need the distance the woven form takes: THREE slots per reverse step,
  // layers 2 + 3 | epilogue | layer 1 of the next step (one tile-step ahead instead of two), no input split in the loop)
  const bool solo = SPARSE && Gr == 1;
  if (solo) G = 3;
  if (cont) G = a.tiles_per_group;   // every round walks all slots; tiles past the end are phantoms
  // ---- per-row constants and the initial state into the B-operand image ----
  // Four consecutive input columns k = 4j .. 4j+3 of one tile column sit in one 16-byte LDS slot of the B-operand image,
  // so a row is moved as 12 quads: 10 straight from the 160-byte state row (16-byte global loads), 2 assembled from
  // hl | stlp | 0.  (One scalar load per element made this prologue a visible part of the single-step launches.)
  for (int e = tid; e < G * kTileRows * (kKx / 4); e += NT) {
    const int tl = e / (kTileRows * (kKx / 4)), rem = e % (kTileRows * (kKx / 4));
    const int c = rem / (kKx / 4), j = rem % (kKx / 4);
    long row = (tile0 + tl) * kTileRows + c;
    if (row >= a.N) row = a.N - 1;
    f32x4 v;
    if (SPARSE && tl >= Gr) {   // an empty pipeline slot: zeros, never the rows of tiles other workgroups own (and rewrite)
      v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    } else if (j < kCtrl / 4) {
      if (REFINE) {
        v = *reinterpret_cast<const f32x4*>(a.init + row * kCtrl + 4 * j);
        if (a.pooled) {  // fused = init + pooled[scene][mode][shard]   (nusc_model.py:186-200)
          const long b = row / a.rows_per_scene;
          const int rr = (int)(row % a.rows_per_scene), s = rr / 3, m = rr % 3;
          const int sh = s / (a.S / a.n_shards);
          v += *reinterpret_cast<const f32x4*>(a.pooled + ((b * 3 + m) * a.n_shards + sh) * kCtrl + 4 * j);
        }
      } else {
        v = *reinterpret_cast<const f32x4*>(a.x_inout + row * kCtrl + 4 * j);
      }
    } else if (j == kCtrl / 4) {
      const float* sp = a.stlp + row * 6;
      v = f32x4{a.hl[row], sp[0], sp[1], sp[2]};
    } else {
      const float* sp = a.stlp + row * 6;
      v = f32x4{sp[3], sp[4], sp[5], 0.0f};
    }
    *reinterpret_cast<f32x4*>(xs + tl * 768 + xs_addr(4 * j, c)) = v;   // (unscaled: the split-f16 factor kSX is applied where the pieces are made)
  }
  if (!REFINE && a.n_emit >= a.steps && a.step_hi == a.steps - 1) {  // x_T itself is entry 0 of the full list
    for (int e = tid; e < (SPARSE ? Gr : G) * kTileRows * kCtrl; e += NT) {
      const long row = tile0 * kTileRows + e / kCtrl;
      const int f = e % kCtrl;
      if (row < a.N) {
        const float sc = (f & 1) ? a.a_max : a.w_max;
        float v = a.x_inout[row * kCtrl + f] * sc;
        if (a.clip) v = clip_keep_nan(v, sc);
        a.emit_out[((long)(a.n_emit - a.steps) * a.N + row) * kCtrl + f] = v;
      }
    }
  }
  const int s_hi = REFINE ? 1 : a.step_hi, s_lo = REFINE ? 1 : a.step_lo;
  // tile-steps of this workgroup; tile-step `it` = (step s_hi - it/G, tile it%G); CONT: (round it/G, tile it%G) of step s_hi
  const int rounds = (int)((n_groups - blockIdx.x + gridDim.x - 1) / gridDim.x);
  const int total = cont ? rounds * G : (s_hi - s_lo + 1) * G;
  auto step_of = [&](int n) { return cont ? s_hi : s_hi - n; };
  auto coef_of = [&](int n) { return cont ? 0 : n; };

  // ---- layer 1 of tile-step `it`: 48 -> 256, result (after ReLU) into h1[it % 3] --------------------------------
  // It is issued two iterations before layer 2 consumes it, at the START of an iteration: its LDS write has long
  // landed when the barrier comes, and its own operand latencies hide under the neighbouring MFMAs.
  // Hazards: it overwrites the buffer layer 2 read in the previous iteration (a barrier ago), and it reads xs[tile],
  // which the epilogue of tile-step it - G rewrote in iteration it - G + 1 <= it - 3 (G >= 4).
  // UT (all 16 rows of a tile share one scene: rows_per_scene % 16 == 0): the scene row of `base` and the step row of
  // `tbias` that layer 1 adds are brought into a 3-slot LDS ring by direct-to-LDS loads (global_load_lds, no
  // registers), issued by one wave a whole iteration before layer 1 needs them.  A plain global load here costs every
  // wave ~1400 stalled cycles per tile-step (measured with the stamp build), because nothing else of the wave can
  // issue while it waits.
  // A tile-step is named by its position (tile tl of the workgroup, n-th reverse step of the launch) and its ring slot
  // (tile-step index mod 3); both are advanced with counters -- divisions by G here cost every wave ~100 scalar
  // instructions per iteration, which the split-bf16 kernel (issue-bound, not MFMA-bound) cannot hide.
  struct Pos {
    int tl, n;
  };
  auto next_pos = [&](Pos p) {
    ++p.tl;
    if (p.tl == G) {
      p.tl = 0;
      ++p.n;
    }
    return p;
  };
  auto stage_cst = [&](Pos p, int slot3) {
    const int tl = p.tl, i = step_of(p.n);
    long row = (t0(p.n) + tl) * kTileRows;
    if (row >= a.N) row = a.N - 1;
    float* dst = crow + slot3 * 512;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const unsigned scene = (unsigned)row / (unsigned)a.rows_per_scene;   // N < 2^31 (checked by the host)
    // (uniform row pointer + 32-bit lane offset: no per-lane 64-bit pointer has to stay live across the loop)
    const unsigned lo4 = here((unsigned)lane * 4u);
    __builtin_amdgcn_global_load_lds((glb_ptr)((a.base + (long)scene * kHid) + lo4), (lds_ptr)dst, 16, 0, 0);
    if (!REFINE)
      __builtin_amdgcn_global_load_lds((glb_ptr)((a.tbias + (long)i * kHid) + lo4), (lds_ptr)(dst + 256), 16, 0, 0);
  };

  // CONT: the input image of the tile at position p (its 16 rows of x, hl | stlp) straight from global memory into its
  // LDS slot by direct-to-LDS loads (no registers), issued by the stager wave.  The image is three 1 KB blocks
  // [q][lane = g*16 + c][4 floats] holding columns 4 (4q + g) .. +3 of row c: blocks 0, 1 and the lower half of block 2
  // are 16-byte pieces of the state rows; columns 40..43 = hl | stlp[0..2] and 44..46 = stlp[3..5] are gathered word by
  // word (the last word of the image, column 47, stays the zero the prologue wrote).
  auto stage_x = [&](Pos p) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const long row0 = (t0(p.n) + p.tl) * kTileRows;
    float* dst = xs + p.tl * 768;
    const unsigned ln = here((unsigned)lane);
    const unsigned gq = ln >> 4, c16 = ln & 15;
    long r = row0 + c16;
    if (r >= a.N) r = a.N - 1;
    const float* xrow = a.x_inout + r * kCtrl + 4 * gq;
    __builtin_amdgcn_global_load_lds((glb_ptr)xrow, (lds_ptr)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(xrow + 16), (lds_ptr)(dst + 256), 16, 0, 0);
    if (gq < 2) __builtin_amdgcn_global_load_lds((glb_ptr)(xrow + 32), (lds_ptr)(dst + 512), 16, 0, 0);
    const unsigned c4 = ln >> 2, e4 = ln & 3;
    long r4 = row0 + c4;
    if (r4 >= a.N) r4 = a.N - 1;
    const float* w0 = e4 == 0 ? a.hl + r4 : a.stlp + r4 * 6 + (e4 - 1);
    __builtin_amdgcn_global_load_lds((glb_ptr)w0, (lds_ptr)(dst + 512 + 128), 4, 0, 0);
    if (e4 < 3) __builtin_amdgcn_global_load_lds((glb_ptr)(a.stlp + r4 * 6 + 3 + e4), (lds_ptr)(dst + 512 + 192), 4, 0, 0);
  };

  // the scene/timestep constant part of layer 1's pre-activation for this lane's 4*OT outputs
  auto l1_const = [&](Pos p, int buf, f32x4 (&cst)[OT]) {
    const int tl = p.tl;
    int i = step_of(p.n);
    if (i < 0) i = 0;   // a tile-step past the end of the launch (computed and discarded by the fused split-bf16 loop)
    if (UT) {
      const f32x4* cb = reinterpret_cast<const f32x4*>(crow + buf * 512);
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        cst[ot] = cb[4 * (w * OT + ot) + g];
        if (!REFINE) cst[ot] += cb[64 + 4 * (w * OT + ot) + g];
        if (F16) cst[ot] *= kAcc;
      }
    } else {
      long rowc = (t0(p.n) + tl) * kTileRows + col;
      if (rowc >= a.N) rowc = a.N - 1;
      const float* bp = a.base + (rowc / a.rows_per_scene) * kHid;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) {
        const int f0 = 16 * (w * OT + ot) + 4 * g;
        cst[ot] = *reinterpret_cast<const f32x4*>(bp + f0);
        if (!REFINE) cst[ot] += *reinterpret_cast<const f32x4*>(a.tbias + (long)i * kHid + f0);
        if (F16) cst[ot] *= kAcc;
      }
    }
  };

  // a hidden layer's output as the next layer's input: relu, (F16) the accumulator's weight factor divided out, pieces
  auto split_hidden = [&](const f32x4& a0, const f32x4& a1, pv8& hi, pv8& lo, int which = 0) {
    f32x4 h0 = relu4(a0), h1v = relu4(a1);
    if ((PSTL_ABL_SKIP & 4) || ((PSTL_ABL_SKIP & 128) && which == 1) || ((PSTL_ABL_SKIP & 256) && which == 2)) {
      hi = __builtin_bit_cast(pv8, a0);
      lo = __builtin_bit_cast(pv8, a1);
      return;
    }
    if (F16) h0 *= kInvSW, h1v *= kInvSW;
    split8(h0, h1v, hi, lo);
    if constexpr (F16) note_pieces(ovf, hi, false);
  };
  // training forward pass of the split forms (REFINE with activation buffers): a hidden layer's fp32 output relu(acc) /
  // kAcc, this lane's 4*OT features of row `col` of tile p.tl (uniform row pointer + 32-bit lane offset, see here())
  auto save_hidden = [&](float* dstbuf, Pos p, const f32x4 (&av)[OT]) {
    if constexpr (REFINE && BF && SAVE) {
      const long row0 = (t0(p.n) + p.tl) * kTileRows;
      const unsigned cc = here((unsigned)col);
      if (row0 + cc < a.N) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
          const unsigned off = cc * (unsigned)kHid + (unsigned)(16 * (w * OT + ot) + 4 * g);
          *reinterpret_cast<f32x4*>((dstbuf + row0 * kHid) + off) = relu4(av[ot]) * kInvAcc;
        }
      }
    }
  };
  auto layer1 = [&](Pos p, int buf) {   // buf = tile-step index mod 3: the h1 buffer written and the crow slot read
    const int tl = p.tl;
    // fetched now, added after the MFMAs
    f32x4 cst[OT];
    l1_const(p, buf, cst);
    f32x4 acc[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) acc[ot] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const f32x4* xb = reinterpret_cast<const f32x4*>(xs + tl * 768) + lane;
    if constexpr (BF) {
      // the fp32 image [q][lane][4] already holds slots 0..3 (q = 2 kb) and 4..7 (q = 2 kb + 1) of this lane's k-block
      // (same operation order as the fused loop below: results do not depend on which of the two computed a tile-step)
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) acc[ot] = cst[ot];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const f32x4 x0 = xb[2 * kb * 64];
        const f32x4 x1 = kb == 0 ? xb[64] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        pv8 bh, bl;
        split8(x0 * kSX, x1 * kSX, bh, bl);
        if constexpr (F16) note_pieces(ovf, bh, true);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma_bf(w1h[ot][kb], bh, acc[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma_bf(w1l[ot][kb], bh, acc[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma_bf(w1h[ot][kb], bl, acc[ot]);
      }
      pv8 hh, hl2;
      split_hidden(acc[0], acc[OT - 1], hh, hl2);
      if constexpr (SAVE) save_hidden(a.h1_save, p, acc);
      u32x4* hwb = reinterpret_cast<u32x4*>(h1 + buf * 4096);   // [kb = producing wave][hi | lo][lane]
      hwb[(w * 2 + 0) * 64 + lane] = __builtin_bit_cast(u32x4, hh);
      hwb[(w * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, hl2);
      return;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const f32x4 bq = xb[q * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma4(w1x[BF ? 0 : ot][q * 4 + r], bq[r], acc[ot]);
    }
    f32x4* hw = reinterpret_cast<f32x4*>(h1 + buf * 4096);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const f32x4 hv = relu4(acc[ot] + cst[ot]);
      hw[(w * OT + ot) * 64 + lane] = hv;
      if (REFINE && a.h1_save) {
        const long row = (t0(p.n) + tl) * kTileRows + col;
        if (row < a.N) *reinterpret_cast<f32x4*>(a.h1_save + row * kHid + 16 * (w * OT + ot) + 4 * g) = hv;
      }
    }
  };

  // ---- epilogue of tile-step `it` (run by the second half of the waves, one tile-step late) ---------------------
  // Hazards: it reads part[it & 1] (complete since the barrier that ended iteration `it`; rewritten only in iteration
  // it + 2, after another barrier) and rewrites xs[tile], which layer1 reads again G - 3 iterations later -- at least
  // one barrier later as long as G >= 4.
  auto epilogue = [&](Pos p, int par, const f32x4& z4) {   // par = tile-step index & 1
    const int tl = p.tl, i = step_of(p.n);
    const long row0 = (t0(p.n) + tl) * kTileRows;
    float c1 = 0.0f, inv_sa = 0.0f, sbeta = 0.0f;
    if (!REFINE) {
      const f32x4 cf = reinterpret_cast<const f32x4*>(coef)[coef_of(p.n)];
      c1 = cf.x;
      inv_sa = cf.y;
      sbeta = cf.z;
    }
    // thread et < 160 owns outputs f0..f0+3 of tile column c: 16-byte LDS accesses at consecutive slots (no bank
    // conflicts), 16-byte global accesses (4 lanes cover one 64-byte piece of a row)
    // (the per-lane constants of this role are re-derived from the thread index every time: kept in registers across
    // the loop beside those of the other roles they were spilled)
    const int et = (int)here((unsigned)(EPI_FIRST ? tid : tid - (NT - NCT)));
    if (et < 160) {
      const int qd = et >> 4, c = et & 15;
      const int j = qd >> 2, gq = qd & 3, slot = gq * 16 + c, f0 = 4 * qd;
      const f32x4* pp = reinterpret_cast<const f32x4*>(part + par * (NW * 768));
      f32x4 o = reinterpret_cast<const f32x4*>(b3s)[qd];
      if (F16) {   // the partial sums carry the factor kAcc
        f32x4 ps = pp[j * 64 + slot];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) ps += pp[(ww * 3 + j) * 64 + slot];
        o += ps * kInvAcc;
      } else {
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) o += pp[(ww * 3 + j) * 64 + slot];
      }
      const long row = row0 + c;
      // global addresses = uniform pointer of the tile's first row + a 32-bit lane offset (c, f0 are per lane, row0 is
      // uniform): per-lane 64-bit pointers kept across the loop were spilled and reloaded behind a full vmcnt wait
      const unsigned loff = (unsigned)(c * kCtrl + f0);
      const f32x4 sc = f32x4{a.w_max, a.a_max, a.w_max, a.a_max};
      if (!REFINE) {
        f32x4* xp = reinterpret_cast<f32x4*>(xs + tl * 768) + j * 64 + slot;
        const f32x4 x = *xp;
        const f32x4 eps = o + x;
        const f32x4 mu = inv_sa * (x - c1 * eps);
        const f32x4 xn = a.mu_only == 2 ? eps : a.mu_only ? mu : mu + sbeta * z4;
        if (!cont) *xp = xn;   // (CONT: one reverse step per launch, the slot is refilled from the next round's rows)
        if (row < a.N) {
          if (i == s_lo) {
            *reinterpret_cast<f32x4*>((a.x_inout + row0 * kCtrl) + loff) = xn;
            // split-f16 domain (|w| < 64, |x| < 4094 for every layer input): an operand outside it turns into inf in the
            // half cast and the state into NaN, which then stays NaN through the remaining steps -- one test per launch
            if (F16 && !(fabsf((xn[0] + xn[1]) + (xn[2] + xn[3])) <= 3.0e38f)) atomicOr(a.status, 1u);
          }
          if (i <= a.n_emit && !a.mu_only) {
            f32x4 v = xn * sc;
            if (a.clip) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = clip_keep_nan(v[r], sc[r]);
            }
            *reinterpret_cast<f32x4*>((a.emit_out + ((long)(a.n_emit - i) * a.N + row0) * kCtrl) + loff) = v;
          }
        }
      } else if (row < a.N) {
        // interval head (nusc_model.py:212-229): tanh output scales into the remaining headroom of init
        const f32x4 init = *reinterpret_cast<const f32x4*>((a.init + row0 * kCtrl) + loff);
        if (a.pre_save) *reinterpret_cast<f32x4*>((a.pre_save + row0 * kCtrl) + loff) = o;
        if (F16 && !(fabsf((o[0] + o[1]) + (o[2] + o[3])) <= 3.0e38f)) atomicOr(a.status, 1u);
        const float viol = (a.scores + row0)[(unsigned)c] < 0.0f ? 1.0f : 0.0f;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float raw = tanhf(o[r]);
          const float d = raw >= 0.0f ? raw * (sc[r] - init[r]) : raw * (init[r] - (-sc[r]));
          v[r] = init[r] + d * viol;
          if (a.clip) v[r] = clip_keep_nan(v[r], sc[r]);
        }
        *reinterpret_cast<f32x4*>((a.out + row0 * kCtrl) + loff) = v;
      }
    }
  };

  // Noise of tile-step `it`, fetched/drawn one iteration before its epilogue runs: by the epilogue waves themselves in
  // the fp32 kernel (drawing it on the partner waves was measured slower there: with "older wave first" arbitration of
  // both the matrix pipe and VALU issue their Philox work is starved behind the epilogue waves' MFMA streams), by the
  // partner waves 4..6 in the split-bf16 kernel (see zbuf above).
  auto fetch_noise = [&](Pos p, int et_, f32x4& z4) {
    const int et = (int)here((unsigned)et_);
    z4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (REFINE || (!a.noise && !a.rng) || a.mu_only) return;
    const int tl = p.tl, i = step_of(p.n);
    if (i <= 1) return;  // the reference adds zeros at the last step
    if (et < 160) {
      const long row0 = (t0(p.n) + tl) * kTileRows;
      const long row = row0 + (et & 15);
      if (row < a.N) {
        if (a.rng) {
          float z[4];
          normal4(seed, a.row_offset + row, et >> 4, i, z);
          z4 = f32x4{z[0], z[1], z[2], z[3]};
        } else {
          const unsigned loff = (unsigned)((et & 15) * kCtrl + 4 * (et >> 4));
          z4 = *reinterpret_cast<const f32x4*>((a.noise + ((long)(a.steps - 1 - i) * a.N + row0) * kCtrl) + loff);
        }
      }
    }
  };

  // XONCE: the half pieces of tile p.tl's input image as layer 1's B operands, into piece buffer `par`: wave 7 makes
  // k-block 0 (quads 0, 1), wave 3 k-block 1 (quad 2; its upper four slots, k = 48..63, are zero).  Both waves sit on
  // SIMD 3, whose waves carry neither the epilogue nor the noise.
  // Hazard: the image of tile p.tl was last written by the epilogue of tile-step (it + 3) - G in iteration it + 4 - G,
  // at least one barrier ago when G >= 5; the buffer written here was last read (by layer 1) in the previous iteration.
  auto split_x = [&](Pos p, int par) {
    if constexpr (XONCE) {
      if (w != 7 && w != 3) return;
      const unsigned ln = here((unsigned)lane);
      const f32x4* xb = reinterpret_cast<const f32x4*>(xs + p.tl * 768) + ln;
      u32x4* dst = xpb + par * 256 + ln;
      pv8 ph, pl;
      if (w == 7) {
        const f32x4 q0 = xb[0], q1 = xb[64];
        split8(q0 * kSX, q1 * kSX, ph, pl);
        if constexpr (F16) note_pieces(ovf, ph, true);
        dst[0] = __builtin_bit_cast(u32x4, ph);
        dst[64] = __builtin_bit_cast(u32x4, pl);
      } else {
        const f32x4 q2 = xb[128], zero = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        split8(q2 * kSX, zero, ph, pl);
        if constexpr (F16) note_pieces(ovf, ph, true);
        dst[128] = __builtin_bit_cast(u32x4, ph);
        dst[192] = __builtin_bit_cast(u32x4, pl);
      }
    }
  };

  Pos pm2{0, 0};                             // tile-step it - 2 (CONT: its slot is refilled now)
  Pos pm1{0, 0}, p0{0, 0};                   // tile-steps it - 1, it, it + 1, it + 2, it + 3
  Pos p1 = next_pos(p0), p2 = next_pos(p1), p3 = next_pos(p2);
  if (UT && w == kStager) {
    stage_cst(p0, 0);
    if (total > 1) stage_cst(p1, 1);
    if (total > 2) stage_cst(p2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();  // xs image, bias/coefficient tables and the first three constant rows are in LDS

  const bool epi_wave = EPI_FIRST ? (w < NCW) : (w >= NW - NCW);  // wave-uniform
  f32x4 zreg = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.emit_out);
  // Stamps are taken lazily: s_memtime into a scalar register pair, NO wait at the stamp (a wait for the scalar-memory
  // counter would also drain the LDS reads in flight and stall exactly the prefetches being measured); one wait at the
  // end of the iteration, then the eight values are written.
  unsigned long long stamp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PSTL_STAMP(slot)                                                                        \
  if (ABL == 7) {                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    asm volatile("s_memtime %0" : "=s"(stamp_t[slot]));                                         \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  }
#define PSTL_STAMP_FLUSH()                                                                      \
  if (ABL == 7) {                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                         \
                 : "+s"(stamp_t[0]), "+s"(stamp_t[1]), "+s"(stamp_t[2]), "+s"(stamp_t[3]),     \
                   "+s"(stamp_t[4]), "+s"(stamp_t[5]), "+s"(stamp_t[6]), "+s"(stamp_t[7])      \
                 :: "memory");                                                                  \
    if (blockIdx.x == 7 && it >= 64 && it < 96 && lane == 0)                                    \
      for (int k_ = 0; k_ < 8; ++k_) dbg[((it - 64) * NW + w) * 8 + k_] = stamp_t[k_];          \
  }

  // ABL == 8: the full kernel with four s_memtime reads per iteration (start, after the role work, before and after the
  // barrier), accumulated per wave: role / body / barrier-wait cycles of workgroup 7's waves, written after the loop to the
  // buffer passed as emit_out as [wave][role, body, barrier, iterations] (64-bit).  Perturbs the schedule far less than the
  // eight fenced stamps of ABL == 7.
  unsigned long long lt_role = 0, lt_body = 0, lt_bar = 0;
#define PSTL_LITE(var)                                                   \
  unsigned long long var = 0;                                            \
  if (ABL == 8) {                                                        \
    __builtin_amdgcn_sched_barrier(0);                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                                   \
  }
  // (SPARSE: empty slots get neither a stand-alone layer 1 nor pieces -- nothing reads them, and they stay out of the guard)
  layer1(p0, 0);
  if (total > 1 && real(p1.tl)) layer1(p1, 1);
  if (real(p2.tl) && !solo) split_x(p2, 0);      // pieces for the layer 1 woven into iteration 0
  __syncthreads();
  // BF, in-kernel noise: waves 4..6 draw it INSIDE their layer-2 MFMA stream (branch-free, one basic block) instead of in
  // a phase of their own in front of it
  bool woven_noise = BF && !REFINE && (ABL == 0 || ABL >= 7) && a.rng && !a.mu_only && w >= NW / 2 && w < NW / 2 + 3 &&
                     !(PSTL_ABL_SKIP & 2);
  // (stamp build: its scalar stamps must not cross a branch the compiler takes for divergent)
  if (ABL == 7) woven_noise = __builtin_amdgcn_readfirstlane((int)woven_noise) != 0;
  // (Tried in round 3, measured with tools/dbg/time_variants.py, results in profiles/r3/chain_variants_pin_prefetch.txt, code in
  // commit dbcba2c: pinning the MFMAs to source order -- accumulators strictly alternating instead of the scheduler's runs of up
  // to nine MFMAs on one accumulator -- +3.3 %; the first B operands of the next tile-step requested before the barrier and
  // carried across it behind a counted lgkmcnt wait, +16 % with the spills it caused and +1.1 % without them (layer 1's
  // half-empty second k-block on v_mfma_f32_16x16x16_f16 frees 8-12 registers; by itself +0.9 %, and it rounds differently).
  // The input split of waves 3 / 7 woven into their fused block as a third variant of it (loads at k-block 4, conversions and
  // stores at k-block 6, where the x pieces of this iteration are dead): 14 registers spilled in the multi-step kernel, dropped.
  // Per-wave phase stamps (chain_waves 816, profiles/r3/chain_phases_per_wave.txt): the noise waves 4-6 arrive last, the
  // epilogue waves 0-2 wait ~700 cycles at the barrier, yet removing the noise altogether gains 1.4 %: the two waves of a SIMD
  // share one issue port -- SQ_ACTIVE_INST_ANY of the pair covers 78 % of the wall time -- and what one sheds the other takes.
  // Wave priorities (s_setprio; profiles/r3/chain_variants_setprio_mix.txt): waves 4-7 above 0-3 +9.7 %, a wave in its fused
  // block above a partner in its role work +13.8 %, the role work above the fused block +1.8 %, waves 0-3 above 4-7 -0.1 %.
  // With the MFMAs in place, deleting the ReLU + piece conversions gains 12-13 % (each of the two sites ~5 %), layer 2's operand
  // reads 3.2 %, the LDS writes 4.5 %, layer 1's constant rows 1.7 %.  Both pieces
  // straight from v_fma_mixlo/hi_f16 (3 instructions per value with the ReLU instead of 4; inline asm, since the SLP vectoriser
  // turns the C++ form into v_pk_fma_f32 + conversions): same bits, +1.8 % -- the scheduler's issue groups do not see asm
  // statements as vector instructions and push them out of the MFMA shadow; the C++ form built with -fno-slp-vectorize does
  // select them, and is +3.7 % (the build flag alone: +3.5 %, the packed fp32 operations elsewhere in the loop are lost).)
  // (Tried in round 2: deferring layer 3 of every tile-step to the head of the next iteration -- accumulators kept across
  // the barrier, epilogue two iterations behind -- so that an iteration ends with layer 2's MFMAs instead of the serial
  // tail split -> layer 3 -> partial sums -> barrier.  Bit-identical, 5.8 % SLOWER (15.42 vs 14.57 ms): behind the barrier
  // the split has no MFMAs of its own wave to hide under.)
  int hbuf = 0;  // it % 3
  for (int it = 0; it < total; ++it) {
    PSTL_LITE(lt0)
    PSTL_STAMP(0)
    if (UT && w == kStager && it + 3 < total && !(PSTL_ABL_SKIP & 32) && real(p3.tl)) stage_cst(p3, hbuf);
    if (cont && w == kStager && it >= 2 && it - 2 + G < total) stage_x(Pos{pm2.tl, pm2.n + 1});
    if (!(PSTL_ABL_SKIP & 16) && real(p3.tl) && !solo) split_x(p3, (it + 1) & 1);   // pieces for the layer 1 woven into iteration it + 1
    if (NOISE_SPLIT && (ABL == 0 || ABL >= 7)) {
      if (w >= NW / 2 && w < NW / 2 + NCW && !woven_noise && !(PSTL_ABL_SKIP & 2) && real(p0.tl)) {
        const int nt = tid - NT / 2;
        f32x4 z;
        fetch_noise(p0, nt, z);
        if (nt < 160) zbuf[(it & 1) * 192 + nt] = z;
      }
      if (epi_wave && it > 0 && !(PSTL_ABL_SKIP & 1) && real(pm1.tl)) {
        const f32x4 z = tid < 160 ? zbuf[((it - 1) & 1) * 192 + tid] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        epilogue(pm1, (it - 1) & 1, z);
      }
    } else if (epi_wave && (ABL == 0 || ABL >= 7)) {
      // While this half finishes the previous tile-step, the partner wave on the same SIMD already issues MFMAs: the
      // matrix pipe never waits for the epilogue (the stagger of MI355X_MICROARCH.md "Two waves per SIMD", item 9).
      const int et = EPI_FIRST ? tid : tid - (NT - NCT);
      const f32x4 zprev = zreg;
      fetch_noise(p0, et, zreg);             // HBM read of this tile-step's noise first: a full iteration to land
      if (it > 0) epilogue(pm1, (it - 1) & 1, zprev);
    }
    PSTL_STAMP(1)
    PSTL_LITE(lt1)
    // ---------------- layer 1 of tile-step it + 2 (two ahead) ------------------------------------------------------
    // (BF: woven into layers 2 + 3 below)
    if (!BF && it + 2 < total) layer1(p2, hbuf == 0 ? 2 : hbuf - 1);
    PSTL_STAMP(2)
    // ---------------- layer 2: 256 -> 256 (B from LDS), layer 3: this wave's 16*OT features -> 48 ----------------
    if (SPARSE && BF && !real(p0.tl)) {   // an empty layer-2 slot: at most layer 1 of tile-step it + 2, stand-alone
      if (solo) {
        if (real(p1.tl) && it + 1 < total) layer1(p1, hbuf == 2 ? 0 : hbuf + 1);   // (solo: one ahead, behind the epilogue)
      } else if (real(p2.tl) && it + 2 < total) {
        layer1(p2, hbuf == 0 ? 2 : hbuf - 1);
      }
    } else {
    f32x4 acc[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) acc[ot] = reinterpret_cast<const f32x4*>(b2s)[(w * OT + ot) * 4 + g];
    const f32x4* hb = reinterpret_cast<const f32x4*>(h1 + hbuf * 4096) + lane;
    f32x4 acc3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc3[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (BF) {
      auto bf_block = [&](auto noise_tag) {
      constexpr bool NOISE = decltype(noise_tag)::value;
      // Layer 1 of tile-step it + 2 is woven into layers 2 + 3 of tile-step it: the kernel is bound by what one wave can
      // issue, and a wave cannot issue past its own MFMA while the matrix pipe is busy, so the conversions of layer 1
      // (independent work) go into the 8 issue cycles each 16-cycle MFMA leaves free.  Layer 1 is computed for the
      // two tile-steps past the end as well (results never read) to keep the loop body one basic block.
      const int b1 = hbuf == 0 ? 2 : hbuf - 1;
      const u32x4* hbb = reinterpret_cast<const u32x4*>(h1 + hbuf * 4096) + lane;
      u32x4 ch = hbb[0], cl = hbb[64];
      const u32x4* xr = xpb + (it & 1) * 256 + lane;
      const u32x4 xq0h = xr[0], xq0l = xr[64], xq1h = xr[128], xq1l = xr[192];
      f32x4 a1[OT];   // starts from the scene/timestep constant part (fetched at k-block 1, used from k-block 2 on)
      pv8 x0h, x0l, x1h, x1l, hh, hl2;
      f32x4 zv = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        u32x4 nh = ch, nl = cl;
        if (kb < 7) {
          if (!(PSTL_ABL_SKIP & 8)) {
            nh = hbb[(2 * kb + 2) * 64];
            nl = hbb[(2 * kb + 3) * 64];
          }
        }
        const pv8 bh = __builtin_bit_cast(pv8, ch), bl = __builtin_bit_cast(pv8, cl);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma_bf(w2h[ot][kb], bh, acc[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma_bf(w2l[ot][kb], bh, acc[ot]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma_bf(w2h[ot][kb], bl, acc[ot]);
        if (kb == 0) x0h = __builtin_bit_cast(pv8, xq0h), x0l = __builtin_bit_cast(pv8, xq0l);
        if (kb == 1) x1h = __builtin_bit_cast(pv8, xq1h), x1l = __builtin_bit_cast(pv8, xq1l);
        if (kb == 2 || kb == 3) {
          const pv8 vh = kb == 2 ? x0h : x1h, vl = kb == 2 ? x0l : x1l;
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) a1[ot] = mfma_bf(w1h[ot][kb - 2], vh, a1[ot]);
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) a1[ot] = mfma_bf(w1l[ot][kb - 2], vh, a1[ot]);
#pragma unroll
          for (int ot = 0; ot < OT; ++ot) a1[ot] = mfma_bf(w1h[ot][kb - 2], vl, a1[ot]);
        }
        if (kb == 1) {
          if (PSTL_ABL_SKIP & 32) {
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) a1[ot] = acc[ot];
          } else {
            l1_const(p2, b1, a1);
          }
        }
        if (kb == 5) {
          const unsigned ovf_before = ovf;
          split_hidden(a1[0], a1[OT - 1], hh, hl2, 2);   // layer 1's output, in the shadow of layer 2's MFMAs
          // (SPARSE: the layer 1 woven into an EMPTY slot ran on whatever the piece buffer and the constant-row ring held --
          // possibly LDS never written in this launch; its result is never read, and it must not reach the domain guard)
#ifndef PSTL_ABL_NO_EMPTY_SLOT_FIX
          if (SPARSE && !real(p2.tl)) ovf = ovf_before;
#endif
          if constexpr (SAVE)
            if (it + 2 < total) save_hidden(a.h1_save, p2, a1);
        }
        if (NOISE && kb == 6) {   // this wave's share of the noise of tile-step it (rows past N and the quads 10, 11 are never read)
          const int nt = tid - NT / 2;
          const int i = step_of(p0.n);
          float z[4];
          normal4(seed, a.row_offset + (t0(p0.n) + p0.tl) * kTileRows + (nt & 15), nt >> 4, i, z);
          zv = i > 1 ? f32x4{z[0], z[1], z[2], z[3]} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        ch = nh;
        cl = nl;
        // issue order of this k-block: the two LDS reads of the next one, then its MFMAs with the conversions between
        if (kb < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        if (kb == 0) {
#pragma unroll
          for (int m = 0; m < 3 * OT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
          }
        } else if (kb == 1) {
          if (UT) __builtin_amdgcn_sched_group_barrier(0x100, 2 * OT, 0);
#pragma unroll
          for (int m = 0; m < 3 * OT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          }
        } else if (kb == 2 || kb == 3) {
          __builtin_amdgcn_sched_group_barrier(0x008, 6 * OT, 0);
        } else if (kb == 5) {
#pragma unroll
          for (int m = 0; m < 3 * OT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          }
        } else if (NOISE && (kb == 6 || kb == 7)) {
#pragma unroll
          for (int m = 0; m < 3 * OT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4-5 per MFMA measured best (2: +3 %, 6: +1.5 %)
          }
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 3 * OT, 0);
        }
        if (ABL == 7 && kb == 1) { PSTL_STAMP(6) }
        if (ABL == 7 && kb == 4) { PSTL_STAMP(7) }
        if (ABL == 7 && kb == 6) { PSTL_STAMP(2) }
      }
      // (no scheduling fence here: left free, the compiler starts layer 3's conversions under layer 2's last MFMAs, -4 %)
      PSTL_STAMP(3)
      // layer 3 of tile-step it; the ReLU + split of layer 1's output sits between its MFMAs
      pv8 bh, bl;
      split_hidden(acc[0], acc[OT - 1], bh, bl, 1);
      if constexpr (SAVE) save_hidden(a.h2_save, p0, acc);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc3[j] = mfma_bf(w3h[j], bh, acc3[j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc3[j] = mfma_bf(w3l[j], bh, acc3[j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc3[j] = mfma_bf(w3h[j], bl, acc3[j]);
      u32x4* hwb = reinterpret_cast<u32x4*>(h1 + b1 * 4096);
      if (!(PSTL_ABL_SKIP & 64)) {
        hwb[(w * 2 + 0) * 64 + lane] = __builtin_bit_cast(u32x4, hh);
        hwb[(w * 2 + 1) * 64 + lane] = __builtin_bit_cast(u32x4, hl2);
      } else {
        asm volatile("" :: "v"(hh), "v"(hl2));
      }
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
      if (NOISE) zbuf[(it & 1) * 192 + (tid - NT / 2)] = zv;
      };
      if (woven_noise) bf_block(std::true_type{});
      else bf_block(std::false_type{});
    } else {
    f32x4 bq = hb[0];
    __builtin_amdgcn_sched_barrier(0);  // the pipelined region starts here
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      f32x4 bn = bq;
      if (q < 15) bn = hb[(q + 1) * 64];  // next B fragment in flight while this one feeds 4*OT MFMAs
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) acc[ot] = mfma4(w2[BF ? 0 : ot][q * 4 + r], bq[r], acc[ot]);
      bq = bn;
      // issue order: the LDS read of fragment q+1, then the 4*OT MFMAs of fragment q (left alone, the scheduler puts
      // the read behind the MFMAs and exposes its latency)
      if (q < 15) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * OT, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    PSTL_STAMP(3)
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const f32x4 h = relu4(acc[ot]);
      if (REFINE && a.h2_save) {
        const long row = (tile0 + p0.tl) * kTileRows + col;
        if (row < a.N) *reinterpret_cast<f32x4*>(a.h2_save + row * kHid + 16 * (w * OT + ot) + 4 * g) = h;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc3[j] = mfma4(w3[j][BF ? 0 : ot][r], h[r], acc3[j]);
    }
    }
    f32x4* pw = reinterpret_cast<f32x4*>(part + (it & 1) * (NW * 768));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (PSTL_ABL_SKIP & 64) asm volatile("" :: "v"(acc3[j]));
      else pw[(w * 3 + j) * 64 + lane] = acc3[j];
    }
    }
    PSTL_STAMP(4)
    if (UT && w == kStager) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staged rows have landed
    PSTL_LITE(lt2)
#ifdef PSTL_ABL_NO_BARRIER   // timing-only ablation: no workgroup barrier in the tile-step loop (results are garbage)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    __syncthreads();
#endif
    PSTL_LITE(lt3)
    if (ABL == 8) lt_role += lt1 - lt0, lt_body += lt2 - lt1, lt_bar += lt3 - lt2;
    PSTL_STAMP(5)
    PSTL_STAMP_FLUSH()
    hbuf = hbuf == 2 ? 0 : hbuf + 1;
    pm2 = pm1, pm1 = p0, p0 = p1, p1 = p2, p2 = p3, p3 = next_pos(p3);
  }
  if (ABL == 8 && blockIdx.x == 7 && lane == 0) {
    dbg[w * 4 + 0] = lt_role, dbg[w * 4 + 1] = lt_body, dbg[w * 4 + 2] = lt_bar, dbg[w * 4 + 3] = (unsigned long long)total;
  }
  if (NOISE_SPLIT && epi_wave && tid < 160) zreg = zbuf[((total - 1) & 1) * 192 + tid];
  if (epi_wave && (ABL == 0 || ABL >= 7) && real(pm1.tl)) epilogue(pm1, (total - 1) & 1, zreg);
  if (ABL != 0 && a.N < 0) epilogue(p0, 0, zreg);  // keep the code reachable for the compiler, never executed
  if (!PERSIST || cont) break;
  }
  if constexpr (F16)
    if (pieces_overflowed(ovf)) atomicOr(a.status, 1u);   // a layer input left |x| < 4094 somewhere in this launch
#ifdef PSTL_WG_TIMES
  if (!REFINE && a.n_emit == 0 && a.step_hi > a.step_lo) {   // (the multi-step launch; emit_out is the caller's debug buffer)
    __syncthreads();
    unsigned long long wg_t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t1));
    if (threadIdx.x == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.emit_out) + 4 * (long)blockIdx.x;
      o[0] = wg_t0, o[1] = wg_t1, o[2] = wg_hw, o[3] = wg_xcc;
    }
  }
#endif
}

// ---- scene encoder (A1) -----------------------------------------------------------------------------------------
// Three 3-layer MLPs (ego 6 -> 256 -> 256 -> 32, neighbour 7 -> ..., lane 45 -> ...) over the tokens of the whole batch,
// ordered [bs ego | bs*K neighbours (scene-major) | 3*bs lanes (scene-major)] so that every encoder is three plain GEMMs
// over contiguous rows:  k_tokens (ego-frame token inputs)  ->  k_enc_gemm x 3 layers (fp32 MFMA)  ->  k_feature_base
// (min / mean / max over the neighbours, the 224-wide feature, and the scene-constant part of layer 1 of policy_net /
// rect_net).  The activations between the layers are row-major (T,256) buffers in global memory (25 MB at 4096 scenes:
// L2 / Infinity-Cache traffic); they are exactly what the encoders' backward pass needs (pstl_encode_scene_saved).
// Round 1-2 had one fused VALU kernel with the activations in LDS (12 tokens per workgroup): every workgroup streamed all
// three encoders' weights and both 224 x 256 layer-1 blocks from L2, 1.4 MB per 2 scenes -- 0.58 ms per 4096 scenes.

// token inputs (normalize_xyth, nusc_model.py:238-263; lanes: first waypoint, then differences, :73-76), zero padded to 48
__global__ void k_tokens(int bs, int K, const float* ego0, const float* neighbors, const float* lane0, const float* lane1,
                         const float* lane2, const float* id0, const float* id1, const float* id2, float* tok_in) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long T = (long)bs * (K + 4);
  if (i >= T) return;
  float o[48];
#pragma unroll
  for (int c = 0; c < 48; ++c) o[c] = 0.0f;
  if (i < bs) {
    const float* ego = ego0 + i * 6;
    o[3] = ego[3], o[4] = ego[4], o[5] = ego[5];
  } else if (i < (long)bs * (K + 1)) {
    const long j = i - bs, sc = j / K;
    const float* ego = ego0 + sc * 6;
    const float bx = ego[0], by = ego[1], bth = ego[2];
    const float cb = cosf(bth), sb = sinf(bth);
    const float* n = neighbors + j * 7;
    const float v = n[0];
    const float xt = n[1] - bx * v, yt = n[2] - by * v;
    o[0] = v;
    o[1] = xt * cb + yt * sb;
    o[2] = -xt * sb + yt * cb;
    o[3] = n[3] - bth * v;
    o[4] = n[4];
    o[5] = n[5];
    o[6] = n[6];
  } else {
    const long j = i - (long)bs * (K + 1), sc = j / 3;
    const int m = (int)(j % 3);
    const float* ego = ego0 + sc * 6;
    const float bx = ego[0], by = ego[1], bth = ego[2];
    const float cb = cosf(bth), sb = sinf(bth);
    const float* lp = (m == 0 ? lane0 : m == 1 ? lane1 : lane2) + sc * 45;
    const float v = (m == 0 ? id0 : m == 1 ? id1 : id2)[sc];
    float px = 0.0f, py = 0.0f, pt = 0.0f;
#pragma unroll
    for (int w = 0; w < 15; ++w) {
      const float xt = lp[3 * w] - bx * v, yt = lp[3 * w + 1] - by * v;
      const float lx = xt * cb + yt * sb, ly = -xt * sb + yt * cb, lt = lp[3 * w + 2] - bth * v;
      o[3 * w] = w == 0 ? lx : lx - px;
      o[3 * w + 1] = w == 0 ? ly : ly - py;
      o[3 * w + 2] = w == 0 ? lt : lt - pt;
      px = lx, py = ly, pt = lt;
    }
  }
  f32x4* dst = reinterpret_cast<f32x4*>(tok_in + i * 48);
#pragma unroll
  for (int q = 0; q < 12; ++q) dst[q] = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
}

// out[row][f] = act(bias[f] + sum_k X[row][k] W[f][k]) on v_mfma_f32_16x16x4_f32, up to three problems per launch
// (workgroup ranges).  Weights register-stationary as A operands (k_pack_a layout, k index permuted k = 16q + 4g + r as
// in the chain kernel), a tile of 16 rows is the B operand, loaded straight from global memory: lane (g, c) takes the
// 16-byte quad X[row c][16q + 4g .. +3] = its B values of the four k-steps (q, 0..3).  No LDS, no barrier: a wave is
// independent of the others.
//   ROWSPLIT = false (256 outputs): wave w owns features [32 w, 32 w + 32) of every tile its workgroup walks (the eight
//                                   waves read the same rows: L1 hits).
//   ROWSPLIT = true  (32 outputs):  every wave holds all of W and walks its own tiles.
struct EncGemmProb {
  const float* X;      // (rows, ldx), 16 K16 readable columns
  const float* A;      // packed weights
  const float* bias;
  float* out;          // (rows, ldo)
  long rows;
  int ldx, ldo, relu;
  int blk0, nblk;      // workgroups [blk0, blk0 + nblk)
};
struct EncGemmArgs {
  EncGemmProb p[3];
  int np;
};

template <int K16, bool ROWSPLIT>
__global__ __launch_bounds__(512) void k_enc_gemm(EncGemmArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int pi = 0;
  if (a.np > 1 && (int)blockIdx.x >= a.p[1].blk0) pi = 1;
  if (a.np > 2 && (int)blockIdx.x >= a.p[2].blk0) pi = 2;
  const EncGemmProb P = a.p[pi];
  const int lb = blockIdx.x - P.blk0;
  float wt[2][K16 * 4];
  f32x4 bq4[2];
#pragma unroll
  for (int ot = 0; ot < 2; ++ot) {
    const int T = ROWSPLIT ? ot : 2 * w + ot;
#pragma unroll
    for (int m = 0; m < K16 * 4; ++m) wt[ot][m] = P.A[((long)T * K16 * 4 + m) * 64 + lane];
    bq4[ot] = *reinterpret_cast<const f32x4*>(P.bias + 16 * T + 4 * g);
  }
  const long n_tiles = (P.rows + 15) / 16;
  const long first = ROWSPLIT ? (long)lb * 8 + w : lb, stride = ROWSPLIT ? (long)P.nblk * 8 : P.nblk;
  for (long t = first; t < n_tiles; t += stride) {
    const long row = t * 16 + c;
    const bool in = row < P.rows;
    f32x4 bq[K16];
#pragma unroll
    for (int q = 0; q < K16; ++q)
      bq[q] = in ? *reinterpret_cast<const f32x4*>(P.X + row * P.ldx + 16 * q + 4 * g) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 acc[2] = {bq4[0], bq4[1]};
#pragma unroll
    for (int q = 0; q < K16; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) acc[ot] = mfma4(wt[ot][q * 4 + r], bq[q][r], acc[ot]);
    if (in) {
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const int T = ROWSPLIT ? ot : 2 * w + ot;
        *reinterpret_cast<f32x4*>(P.out + row * P.ldo + 16 * T + 4 * g) = P.relu ? relu4(acc[ot]) : acc[ot];
      }
    }
  }
}

// feature = [ego 32 | nei min 32 | nei mean 32 | nei max 32 | lanes 3x32]  (nusc_model.py:82-93) and the scene-constant
// 224 columns of layer 1 of policy_net / rect_net: base_x[b][h] = bias1[h] + sum_k W1[h][k] feature[b][k].  16 scenes per
// workgroup share every layer-1 weight they stream from L2.
constexpr int kFbScn = 16;
__global__ __launch_bounds__(256) void k_feature_base(int bs, int K, const float* tok_out, const float* packed, ChainOff pol,
                                                      ChainOff rect, float* feature, float* base_policy, float* base_rect) {
  __shared__ float feat_s[kFbScn][kFeat];
  const int tid = threadIdx.x;
  const long b0 = (long)blockIdx.x * kFbScn;
  const int ns = (bs - b0) < kFbScn ? (int)(bs - b0) : kFbScn;
  for (int i = tid; i < ns * 32; i += 256) {
    const int sc = i >> 5, o = i & 31;
    const long b = b0 + sc;
    feat_s[sc][o] = tok_out[b * 32 + o];
    float mn = INFINITY, mx = -INFINITY, sm = 0.0f;
    for (int k = 0; k < K; ++k) {
      const float v = tok_out[((long)bs + b * K + k) * 32 + o];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
      sm += v;
    }
    feat_s[sc][32 + o] = mn;
    feat_s[sc][64 + o] = sm / (float)K;
    feat_s[sc][96 + o] = mx;
    for (int m = 0; m < 3; ++m) feat_s[sc][128 + 32 * m + o] = tok_out[((long)bs * (K + 1) + b * 3 + m) * 32 + o];
  }
  __syncthreads();
  const int which = blockIdx.y;   // 0: policy_net (and the feature itself), 1: rect_net
  if (which == 0)
    for (int i = tid; i < ns * kFeat; i += 256) feature[b0 * kFeat + i] = feat_s[i / kFeat][i % kFeat];
  {
    float* dst = which == 0 ? base_policy : base_rect;
    if (!dst) return;
    const ChainOff& co = which == 0 ? pol : rect;
    const float bias = packed[co.b1 + tid];
    const float* wp = packed + co.w1f;
    float acc[kFbScn];
#pragma unroll
    for (int u = 0; u < kFbScn; ++u) acc[u] = bias;
#pragma unroll 4
    for (int k = 0; k < kFeat; ++k) {
      const float wv = wp[k * kHid + tid];
#pragma unroll
      for (int u = 0; u < kFbScn; ++u) acc[u] += wv * feat_s[u][k];
    }
#pragma unroll
    for (int u = 0; u < kFbScn; ++u)
      if (u < ns) dst[(b0 + u) * kHid + tid] = acc[u];
  }
}

// ---- merge_net + shard max-pool (nusc_model.py:186-196) -----------------------------------------------------------
struct MergeArgs {
  int bs, S, n_shards;
  MergeOff off;
  const float* packed;
  const float* init;   // (N,40)
  float* pooled;       // (bs,3,n_shards,40)
};

// LDS copy of the packed merge_net block: w0t [40][32] | b0 [32] | w1t [32][32] | b1 [32] | w2t [32][40] | b2 [40]
constexpr int kMrgW0 = 0, kMrgB0 = 40 * 32, kMrgW1 = kMrgB0 + 32, kMrgB1 = kMrgW1 + 32 * 32, kMrgW2 = kMrgB1 + 32,
              kMrgB2 = kMrgW2 + 32 * 40, kMrgFloats = kMrgB2 + 40;   // 3688

// merge_net on one row, lane-private: h0 / h1v = the two hidden layers BEFORE their ReLU, out[0..40) the output (LDS).
// One k-ordered fma chain per output: the forward (k_merge_pool) and the backward's recomputation (k_merge_bwd) see the
// same bits, so the backward finds the maximum the forward pooled.
// WP: where the weights are read from -- an LDS copy (const float*) or the packed buffer through the constant address
// space (kconst_f32: uniform indices become scalar loads, the weights reach the FMAs as SGPR operands and no LDS
// bandwidth is spent on them; same operations, same bits).
typedef const __attribute__((address_space(4))) float* kconst_f32;
template <typename WP>
__device__ __forceinline__ void merge_row(const float* x, WP wt, float (&h0)[32], float (&h1v)[32], float* out) {
  const WP w0t = wt + kMrgW0;
  const WP b0 = wt + kMrgB0;
  const WP w1t = wt + kMrgW1;
  const WP b1 = wt + kMrgB1;
  const WP w2t = wt + kMrgW2;
  const WP b2 = wt + kMrgB2;
  // Every accumulation is an EXPLICIT fused multiply-add in k order: left to the compiler's contraction the same source
  // came out partly as v_pk_mul + v_add and partly as v_fmac, differently in each instantiation -- forward and backward
  // would then disagree on near-ties of the max-pool.
#pragma unroll
  for (int o = 0; o < 32; ++o) h0[o] = b0[o];
  f32x4 xq[10];   // the row as ten 16-byte loads (a lane reads its own 160-byte row: 40 scalar loads touched 64 lines each)
#pragma unroll
  for (int q = 0; q < 10; ++q) xq[q] = reinterpret_cast<const f32x4*>(x)[q];
#pragma unroll
  for (int k = 0; k < 40; ++k) {
    const float xv = xq[k >> 2][k & 3];
#pragma unroll
    for (int o = 0; o < 32; ++o) h0[o] = __builtin_fmaf(w0t[k * 32 + o], xv, h0[o]);
  }
#pragma unroll
  for (int o = 0; o < 32; ++o) h1v[o] = b1[o];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const float hv = fmaxf(h0[k], 0.0f);
#pragma unroll
    for (int o = 0; o < 32; ++o) h1v[o] = __builtin_fmaf(w1t[k * 32 + o], hv, h1v[o]);
  }
  float o2[40];
#pragma unroll
  for (int o = 0; o < 40; ++o) o2[o] = b2[o];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const float hv = fmaxf(h1v[k], 0.0f);
#pragma unroll
    for (int o = 0; o < 40; ++o) o2[o] = __builtin_fmaf(w2t[k * 40 + o], hv, o2[o]);
  }
#pragma unroll
  for (int o = 0; o < 40; ++o) out[o] = o2[o];
}

// (256 registers = two waves per SIMD is the best point: pinned to 1, 3, 4 or 6 waves per SIMD it ran 10 % slower)
__global__ __launch_bounds__(64) void k_merge_pool(MergeArgs a) {
  __shared__ float outs[64][41];
  const int tid = threadIdx.x;
  const kconst_f32 wt = (kconst_f32)(a.packed + a.off.w0t);
  // one workgroup per (scene, mode); the S samples are walked 64 at a time (all lanes busy when S >= 64); the running
  // maximum of every shard is kept by the 40 x n_shards threads that own one (shard, output) pair each
  const long bm = blockIdx.x;
  const int m = (int)(bm % 3);
  const long b = bm / 3;
  const int sps = a.S / a.n_shards;
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // up to 4 (shard, output) pairs per thread
  __syncthreads();
  for (int s0 = 0; s0 < a.S; s0 += 64) {
    const int s = s0 + tid;
    if (s < a.S) {
      const long row = (b * a.S + s) * 3 + m;
      float h0[32], h1v[32];
      merge_row(a.init + row * kCtrl, wt, h0, h1v, outs[tid]);
    }
    __syncthreads();
    const int n = (a.S - s0) < 64 ? (a.S - s0) : 64;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = tid + 64 * u;               // (shard, output) pair
      if (p < a.n_shards * 40) {
        const int sh = p / 40, o = p % 40;
        for (int s2 = 0; s2 < n; ++s2)
          if ((s0 + s2) / sps == sh) best[u] = fmaxf(best[u], outs[s2][o]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = tid + 64 * u;
    if (p < a.n_shards * 40) a.pooled[bm * a.n_shards * kCtrl + p] = best[u];
  }
}

// ---- merge_net backward (training with --joint) ------------------------------------------------------------------
// fused = init + pooled[scene, mode, shard] feeds rect_net's last 40 input columns, so d pooled[shard][c] is the sum of
// d fused[row][c] over the shard's rows and goes to the ONE row whose merge_net output c was the shard's maximum
// (torch.max(dim): the first such row).  One wave per (scene, mode), lane = sample s (S <= 64, as the DPP loss kernel
// requires): recompute the row's merge_net forward (merge_row), find the winners, back-propagate the sparse d out through
// the two hidden layers lane-privately, and accumulate the weight gradients -- 58 entries per lane, held in registers
// over all the (scene, mode) pairs this workgroup walks -- from per-layer (d output | layer input) row vectors staged in
// LDS; rows without a winner are skipped.  Every workgroup writes one slab; k_merge_reduce adds the slabs in order.
struct MergeBwdArgs {
  int bs, S, n_shards;
  MergeOff off;
  const float* packed;
  const float* init;     // (N,40)
  const float* dfused;   // (N,ldf): d loss / d rect_net input columns 231..270
  int ldf;
  float* slabs;          // (gridDim.x, kMrgFloats): dW0 (32,40) | db0 | dW1 (32,32) | db1 | dW2 (40,32) | db2, reference layout
};

__global__ __launch_bounds__(64) void k_merge_bwd(MergeBwdArgs a) {
  const kconst_f32 wt = (kconst_f32)(a.packed + a.off.w0t);   // weights through scalar loads (see merge_row)
  __shared__ float outs[64][41];
  __shared__ float rowv[64][73];          // per layer: d output (<= 40) | layer input (<= 40)
  __shared__ float dpl[256];              // d pooled[(shard, output)]
  __shared__ int win[256];                // winning sample of (shard, output)
  __shared__ int live[64];                // row holds a winner
  const int tid = threadIdx.x;
  const int sps = a.S / a.n_shards;
  float g2[20], g1[16], g0[20], gb[2];    // dW2[c = 2j + (tid>>5)][k = tid&31], dW1[o = 2j + (tid>>5)][k], dW0[o = e/40][k = e%40], biases
#pragma unroll
  for (int j = 0; j < 20; ++j) g2[j] = g0[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) g1[j] = 0.0f;
  gb[0] = gb[1] = 0.0f;
  __syncthreads();
  for (long bm = blockIdx.x; bm < (long)a.bs * 3; bm += gridDim.x) {
    const int m = (int)(bm % 3);
    const long b = bm / 3;
    const int s = tid;
    const bool has = s < a.S;
    const long row = (b * a.S + (has ? s : 0)) * 3 + m;
    float h0[32], h1v[32], xin[40];
    if (has) {
#pragma unroll
      for (int k = 0; k < 40; ++k) xin[k] = a.init[row * kCtrl + k];
      merge_row(a.init + row * kCtrl, wt, h0, h1v, outs[tid]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = tid + 64 * u;
      if (p < a.n_shards * 40) {
        const int sh = p / 40, o = p % 40;
        float best = -INFINITY, dsum = 0.0f;
        int bi = sh * sps;
        for (int s2 = sh * sps; s2 < (sh + 1) * sps; ++s2) {
          const float v = outs[s2][o];
          if (v > best) best = v, bi = s2;
          dsum += a.dfused[((b * a.S + s2) * 3 + m) * a.ldf + o];
        }
        win[p] = bi;
        dpl[p] = dsum;
      }
    }
    __syncthreads();
    // d out of this row, then the two hidden layers (lane-private)
    float dout[40], dh1[32], dh0[32];
    bool any = false;
    const int sh = has ? s / sps : 0;
#pragma unroll
    for (int o = 0; o < 40; ++o) {
      const bool mine = has && win[sh * 40 + o] == s;
      dout[o] = mine ? dpl[sh * 40 + o] : 0.0f;
      any |= mine;
    }
    live[tid] = any ? 1 : 0;
    if (any) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        float acc = 0.0f;
#pragma unroll
        for (int o = 0; o < 40; ++o) acc += wt[kMrgW2 + k * 40 + o] * dout[o];
        dh1[k] = h1v[k] > 0.0f ? acc : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        float acc = 0.0f;
#pragma unroll
        for (int o = 0; o < 32; ++o) acc += wt[kMrgW1 + k * 32 + o] * dh1[o];
        dh0[k] = h0[k] > 0.0f ? acc : 0.0f;
      }
    }
    // layer 2: dW2[c][k] += dout[c] * relu(h1v[k]);  db2[c] += dout[c]
    if (any) {
#pragma unroll
      for (int o = 0; o < 40; ++o) rowv[tid][o] = dout[o];
#pragma unroll
      for (int k = 0; k < 32; ++k) rowv[tid][40 + k] = fmaxf(h1v[k], 0.0f);
    }
    __syncthreads();
    for (int r = 0; r < a.S; ++r) {
      if (!live[r]) continue;
      const float hv = rowv[r][40 + (tid & 31)];
#pragma unroll
      for (int j = 0; j < 20; ++j) g2[j] += rowv[r][2 * j + (tid >> 5)] * hv;
      if (tid < 40) gb[0] += rowv[r][tid];
    }
    __syncthreads();
    // layer 1: dW1[o][k] += dh1[o] * relu(h0[k]);  db1[o] += dh1[o]
    if (any) {
#pragma unroll
      for (int o = 0; o < 32; ++o) rowv[tid][o] = dh1[o];
#pragma unroll
      for (int k = 0; k < 32; ++k) rowv[tid][40 + k] = fmaxf(h0[k], 0.0f);
    }
    __syncthreads();
    for (int r = 0; r < a.S; ++r) {
      if (!live[r]) continue;
      const float hv = rowv[r][40 + (tid & 31)];
#pragma unroll
      for (int j = 0; j < 16; ++j) g1[j] += rowv[r][2 * j + (tid >> 5)] * hv;
      if (tid >= 40) gb[0] += rowv[r][tid - 40];   // lanes 40..63 hold db1[0..23] in gb[0], lanes 0..7 db1[24..31] in gb[1]
      if (tid < 8) gb[1] += rowv[r][24 + tid];
    }
    __syncthreads();
    // layer 0: dW0[o][k] += dh0[o] * x[k];  db0[o] += dh0[o]   (entry e = tid + 64 j of the 32 x 40 matrix)
    if (any) {
#pragma unroll
      for (int o = 0; o < 32; ++o) rowv[tid][o] = dh0[o];
#pragma unroll
      for (int k = 0; k < 40; ++k) rowv[tid][32 + k] = xin[k];
    }
    __syncthreads();
    for (int r = 0; r < a.S; ++r) {
      if (!live[r]) continue;
#pragma unroll
      for (int j = 0; j < 20; ++j) {
        const int e = tid + 64 * j;
        g0[j] += rowv[r][e / 40] * rowv[r][32 + e % 40];
      }
      if (tid >= 8 && tid < 40) gb[1] += rowv[r][tid - 8];   // lanes 8..39: db0[0..31]
    }
    __syncthreads();
  }
  float* out = a.slabs + (long)blockIdx.x * kMrgFloats;
  // slab layout: dW0 (32,40) at 0 | db0 at 1280 | dW1 (32,32) at 1312 | db1 at 2336 | dW2 (40,32) at 2368 | db2 at 3648
#pragma unroll
  for (int j = 0; j < 20; ++j) out[tid + 64 * j] = g0[j];
#pragma unroll
  for (int j = 0; j < 16; ++j) out[1312 + (2 * j + (tid >> 5)) * 32 + (tid & 31)] = g1[j];
#pragma unroll
  for (int j = 0; j < 20; ++j) out[2368 + (2 * j + (tid >> 5)) * 32 + (tid & 31)] = g2[j];
  if (tid < 40) out[3648 + tid] = gb[0];            // db2
  if (tid >= 40) out[2336 + tid - 40] = gb[0];      // db1[0..23]
  if (tid < 8) out[2336 + 24 + tid] = gb[1];        // db1[24..31]
  if (tid >= 8 && tid < 40) out[1280 + tid - 8] = gb[1];   // db0
}

// gradient tensors (reference layout) = sum of the slabs, in slab order
__global__ void k_merge_reduce(int nslabs, const float* slabs, float* dw0, float* db0, float* dw1, float* db1, float* dw2,
                               float* db2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kMrgFloats) return;
  float acc = 0.0f;
  int sl = 0;
  for (; sl + 8 <= nslabs; sl += 8) {   // eight loads in flight, added in slab order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slabs[(long)(sl + u) * kMrgFloats + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; sl < nslabs; ++sl) acc += slabs[(long)sl * kMrgFloats + i];
  if (i < 1280) dw0[i] = acc;
  else if (i < 1312) db0[i - 1280] = acc;
  else if (i < 2336) dw1[i - 1312] = acc;
  else if (i < 2368) db1[i - 2336] = acc;
  else if (i < 3648) dw2[i - 2368] = acc;
  else db2[i - 3648] = acc;
}

__global__ void k_fill_normal(long N, unsigned long long seed, const pstl_dyn* dyn, long row_offset, int step, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (row, quad)
  if (i >= N * 10) return;
  if (dyn) seed = uniform_u64(&dyn->seed);
  const long row = i / 10;
  const int quad = (int)(i % 10);
  float z[4];
  normal4(seed, row_offset + row, quad, step, z);
  *reinterpret_cast<f32x4*>(out + row * kCtrl + 4 * quad) = f32x4{z[0], z[1], z[2], z[3]};
}

// Tiles per workgroup: 12 (192 rows) when there is enough work for every CU; small batches (the closed-loop caller
// runs 192 rows per simulation step, reference nusc_sim.py) are spread over more workgroups, down to the 5 tiles the
// software pipeline needs, which cuts the latency of one reverse step from 12 to 5 tile-iterations.
inline int cu_count();
inline int tiles_per_group(long N) {
  const long n_tiles = (N + kTileRows - 1) / kTileRows;
  long g = n_tiles / 256;
  if (g < 5) g = 5;
  if (g > kG) g = kG;
  return (int)g;
}
// Multi-step launches (one workgroup per CU and round, every workgroup alive for the whole launch): the fewest rounds of
// <= 12-tile workgroups, and the tiles spread evenly over rounds x CUs workgroups -- so that a batch just past a multiple
// of 12 tiles per CU (3 200 tiles on 256 CUs: 267 twelve-tile groups = a second round for 11 workgroups, 24 tile-steps per
// reverse step where 2 x 7 do) does not pay a whole extra round.  Results do not depend on the grouping (row-wise
// arithmetic, noise keyed by the global row).
inline int tiles_per_group_balanced(long N) {
  const long n_tiles = (N + kTileRows - 1) / kTileRows;
  const long cus = cu_count();
  const long rounds = (n_tiles + cus * kG - 1) / (cus * kG);
  long g = (n_tiles + cus * rounds - 1) / (cus * rounds);
  if (g < 5) g = 5;
  if (g > kG) g = kG;
  return (int)g;
}

// The latency layout: fewer than five tiles per CU -> 1..5 tiles per workgroup, spread over as many CUs as there are tiles
// (0 = enough work for the throughput layout).  (Five -- more than four but fewer than five tiles per CU -- fills every
// pipeline slot with a real tile: the SPARSE instantiation then skips nothing, and unlike 5-tile groups of the throughput
// layout it computes no phantom tiles; tests: test_latency_layout_equals_throughput_layout[100-64-6].)
inline int sparse_tiles_per_group(long N) {
  const long n_tiles = (N + kTileRows - 1) / kTileRows;
  const long cus = cu_count();
  if (n_tiles >= 5 * cus) return 0;
  return (int)((n_tiles + cus - 1) / cus);
}

template <int NW>
size_t chain_lds_bytes() {
  return (size_t)(kG * 768 + 3 * 16 * 256 + 2 * NW * 768 + 256 + 48 + 4 * kMaxLaunchSteps + 3 * 512 + 2 * 192 * 4 +
                  2 * 4 * 64 * 4) * sizeof(float);
}

inline int cu_count() { return device_cus(); }   // (of the CURRENT device: pstl_common.hpp's per-device table)

template <int NW, bool REFINE, int ABL = 0, bool UT = false, int PT = 0, bool PERSIST = false, bool SAVE = false,
          bool SPARSE = false>
int launch_chain(const ChainArgs& a, hipStream_t st) {
  const long n_tiles = (a.N + kTileRows - 1) / kTileRows;
  const long n_groups = (n_tiles + a.tiles_per_group - 1) / a.tiles_per_group;
  const dim3 grid((unsigned)(PERSIST && n_groups > cu_count() ? cu_count() : n_groups));
  const size_t lds = chain_lds_bytes<NW>();
  auto fn = k_chain<NW, REFINE, ABL, UT, PT, PERSIST, SAVE, SPARSE>;
  // (once per instantiation and device: the attribute call costs host time on every launch of a latency-bound caller)
  static DeviceOnce allowed;
  const int dev = current_device();
  if (dev < 0) return PSTL_ERR_LAUNCH;
  if (!allowed.done(dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return PSTL_ERR_LAUNCH;
    allowed.set(dev);
  }
  hipLaunchKernelGGL(fn, grid, dim3(NW * 64), lds, st, a);
  return launch_status();
}

// Batches that k_chain2 runs faster than k_chain.  Multi-step launches: its workgroups own 256 (or 192) rows for the whole launch,
// one per CU at a time, so the launch takes ceil(workgroups / CUs) rounds of its tile-step (chain2_step_cost: in per cent of the
// 256-row tile-step, 25.6 us) against k_chain's rounds x tiles per group x 1.91 us (7.46 per cent per tile) -- it pays where
// its rounds are (nearly) full: from 45 072 rows (2 817 tiles: 235 workgroups of 192 rows against twelve-tile groups of k_chain) and at every size from 196 608 rows up, not at
// 24 576 ... 40 960 rows nor where a round would be mostly empty (profiles/r5/chain2_192_row_workgroups.txt).  The single-step
// launches of the guided phase: one workgroup per CU walks the tiles, no rounds -- measured faster than k_chain's single-step
// layout at every size from 24 576 rows up, so it takes every batch the latency layout does not
// (profiles/r5/chain2_single_step_sizes.txt).
inline bool chain2_pays(long N, bool single_step) {
  if (single_step) return sparse_tiles_per_group(N) == 0;
  const long cus = cu_count(), n_tiles = (N + kTileRows - 1) / kTileRows;
  const long g = tiles_per_group_balanced(N), rk = ((n_tiles + g - 1) / g + cus - 1) / cus;
  ChainArgs a = {};
  a.N = N;
  return chain2_step_cost(a) * 100 < rk * g * 746 * 97 / 100;   // (both sides in 0.01 % of a 256-row tile-step; 3 % in hand for k_chain)
}

// cfg->chain_waves: 0, 16 = both networks on split-f16 MFMA: every fp32 operand as two half pieces, three
//                          v_mfma_f32_16x16x32_f16 products per fp32 product, fp32 accumulation (operands good to 2^-23:
//                          results sit as close to the reference as an fp32 fmaf chain in another summation order does),
//                   32   = policy_net on split-bf16 MFMA (operands good to 2^-17; the round-1 default, kept for
//                          comparison), rect_net on fp32 MFMA,
//                   8    = eight waves x 32 output features on fp32 MFMA (2 waves/SIMD, <=256 registers each),
//                   4    = four waves x 64 output features on fp32 MFMA (1 wave/SIMD, weights partly in AGPRs)
template <bool REFINE>
int launch_chain_nw(int chain_waves, const ChainArgs& a, hipStream_t st) {
  const bool ut = (a.rows_per_scene % kTileRows == 0);  // every 16-row tile lies inside one scene
  // 0: the latency layout for batches that cannot give every CU a full five-tile pipeline (SPARSE, see k_chain); 16: the same
  // arithmetic in the throughput layout whatever the batch size (what large batches get either way; bit-identical results)
  const bool latency = chain_waves == 0;
  // 2: the row-stationary kernel (k_chain2, chain2_kernels.hip) for every launch it can take, whatever the batch size; 0:
  // for the multi-step denoiser launches of batches that fill whole rounds of its 256-row workgroups (chain2_pays)
  // (which of the two kernels: by the job's row count where the caller names one -- cfg->plan_rows, so that every shard of a
  // job runs the same kernel whatever its own size --, by this call's otherwise)
  const long plan = a.plan_N > 0 ? a.plan_N : a.N;
  if constexpr (!REFINE) {
    if ((chain_waves == 2 || (chain_waves == 0 && chain2_pays(plan, a.step_hi == a.step_lo))) && chain2_eligible(a))
      return launch_chain2(a, st);
  }
  if constexpr (REFINE) {   // RefineNet's inference pass: k_chain2's tile-walking form, for the batches its single-step form takes
    if ((chain_waves == 2 || (chain_waves == 0 && chain2_pays(plan, true))) && chain2_refine_eligible(a)) return launch_chain2_refine(a, st);
  }
  if (chain_waves == 16 || chain_waves == 2) chain_waves = 0;
  if (REFINE && chain_waves == 32) chain_waves = 8;    // bf16 pieces cost the interval head up to 9e-5 per pass
  if (REFINE && a.h1_save && chain_waves != 0) chain_waves = chain_waves == 4 ? 4 : 8;   // (no bf16-piece training forward)
  if (chain_waves == 0) {
    if constexpr (!REFINE)
      if (ut && latency) {   // small batch: the latency layout, for multi-step and single-step launches alike
        // (a shard much smaller than its job's plan, which chose k_chain for it: the layouts of ONE kernel give the same bits)
        const int g = sparse_tiles_per_group(a.N);
        if (g > 0) {
          ChainArgs b = a;
          b.tiles_per_group = g;
          return launch_chain<8, false, 0, true, 2, false, false, true>(b, st);
        }
      }
    if constexpr (!REFINE)
      if (ut && a.step_hi == a.step_lo) return launch_chain<8, false, 0, true, 2, true>(a, st);   // single step: persistent
    if constexpr (REFINE)
      if (a.h1_save && a.h2_save)   // training forward pass
        return ut ? launch_chain<8, true, 0, true, 2, false, true>(a, st) : launch_chain<8, true, 0, false, 2, false, true>(a, st);
    if constexpr (!REFINE)
      if (a.step_hi > a.step_lo) {   // multi-step: whole rounds of evenly sized workgroups
        ChainArgs b = a;
        b.tiles_per_group = tiles_per_group_balanced(a.N);
        return ut ? launch_chain<8, false, 0, true, 2>(b, st) : launch_chain<8, false, 0, false, 2>(b, st);
      }
    return ut ? launch_chain<8, REFINE, 0, true, 2>(a, st) : launch_chain<8, REFINE, 0, false, 2>(a, st);
  }
  if (chain_waves == 32) {
    if constexpr (!REFINE) {
      if (ut && a.step_hi == a.step_lo) return launch_chain<8, false, 0, true, 1, true>(a, st);
      return ut ? launch_chain<8, false, 0, true, 1>(a, st) : launch_chain<8, false, 0, false, 1>(a, st);
    }
  }
  if (chain_waves == 8) return ut ? launch_chain<8, REFINE, 0, true>(a, st) : launch_chain<8, REFINE>(a, st);
  if (chain_waves == 4) return launch_chain<4, REFINE>(a, st);
#ifdef PSTL_DIAG
  // Timing-only diagnostic instantiations (see the comment above k_chain; results are wrong by construction).  They exist
  // only in builds made with -DPSTL_DIAG (tools/dbg); the shipped library answers PSTL_ERR_SHAPE to these values.
  if (REFINE) return chain_waves > 100 ? launch_chain<8, true>(a, st) : PSTL_ERR_SHAPE;
  switch (chain_waves) {
    case 108: return launch_chain<8, false, 1>(a, st);
    case 708: return ut ? launch_chain<8, false, 7, true>(a, st) : launch_chain<8, false, 7>(a, st);
    case 1008: return launch_chain<8, false>(a, st);   // force the general (non-uniform-tile) path
    case 116: return launch_chain<8, false, 1, true, 2>(a, st);
    case 716: return launch_chain<8, false, 7, true, 2>(a, st);
    case 816: return launch_chain<8, false, 8, true, 2>(a, st);   // role / body / barrier-wait cycles per wave
    default: return PSTL_ERR_SHAPE;
  }
#else
  return PSTL_ERR_SHAPE;   // unknown chain_waves
#endif
}

}  // namespace
}  // namespace pstl

using namespace pstl;

extern "C" int pstl_version(void) { return PSTL_ABI_VERSION; }

extern "C" const char* pstl_error_string(int code) {
  switch (code) {
    case PSTL_OK: return "ok";
    case PSTL_ERR_ARG: return "bad argument (null pointer or size)";
    case PSTL_ERR_SHAPE: return "shape not supported by this build";
    case PSTL_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}

extern "C" size_t pstl_packed_weight_floats(void) { return (size_t)make_layout().total; }
extern "C" size_t pstl_packed_status_offset(void) { return (size_t)make_layout().status; }

static int pack_mlp_t(const pstl_mlp3& m, int in, int hid, int out, const long* off6, float* packed, hipStream_t st) {
  // off6: w0t, b0, w1t, b1, w2t, b2
  auto tr = [&](const float* W, int o, int i, long dst) {
    const long n = (long)o * i;
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, o, i, 0, i, packed + dst);
  };
  auto cp = [&](const float* b, int n, long dst) {
    hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, st, b, n, n, packed + dst);
  };
  tr(m.w0, hid, in, off6[0]);
  cp(m.b0, hid, off6[1]);
  tr(m.w1, hid, hid, off6[2]);
  cp(m.b1, hid, off6[3]);
  tr(m.w2, out, hid, off6[4]);
  cp(m.b2, out, off6[5]);
  return launch_status();
}

static bool mlp_ok(const pstl_mlp3& m) { return m.w0 && m.b0 && m.w1 && m.b1 && m.w2 && m.b2; }

static int pack_chain(const pstl_mlp3& m, int in, int kext_mode, bool with_time, const ChainOff& o, float* packed,
                      unsigned* wmax, hipStream_t st) {
  long n = (long)kFeat * kHid;
  hipLaunchKernelGGL(k_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m.w0, kHid, in, 0, kFeat,
                     packed + o.w1f);
  hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, st, m.b0, kHid, kHid, packed + o.b1);
  if (with_time) {
    n = 32L * kHid;
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m.w0, kHid, in, 264, 32,
                       packed + o.w1t);
  }
  hipLaunchKernelGGL(k_pack_a, dim3(16 * 3), dim3(256), 0, st, m.w0, in, kHid, 16, 3, kext_mode, packed + o.w1x);
  hipLaunchKernelGGL(k_pack_a, dim3(16 * 16), dim3(256), 0, st, m.w1, kHid, kHid, 16, 16, 0, packed + o.w2);
  hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, st, m.b1, kHid, kHid, packed + o.b2);
  hipLaunchKernelGGL(k_pack_a, dim3(3 * 16), dim3(256), 0, st, m.w2, kHid, kCtrl, 3, 16, 0, packed + o.w3);
  hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, st, m.b2, kCtrl, 48, packed + o.b3);
  unsigned* pw = reinterpret_cast<unsigned*>(packed);
  hipLaunchKernelGGL(k_pack_a_split<false>, dim3(16 * 2 * 2), dim3(256), 0, st, m.w0, in, kHid, 16, 2, kext_mode, pw + o.w1xb);
  hipLaunchKernelGGL(k_pack_a_split<false>, dim3(16 * 8 * 2), dim3(256), 0, st, m.w1, kHid, kHid, 16, 8, 0, pw + o.w2b);
  hipLaunchKernelGGL(k_pack_a_split<false>, dim3(3 * 8 * 2), dim3(256), 0, st, m.w2, kHid, kCtrl, 3, 8, 0, pw + o.w3b);
  hipLaunchKernelGGL(k_pack_a_split<true>, dim3(16 * 2 * 2), dim3(256), 0, st, m.w0, in, kHid, 16, 2, kext_mode, pw + o.w1xh, wmax);
  hipLaunchKernelGGL(k_pack_a_split<true>, dim3(16 * 8 * 2), dim3(256), 0, st, m.w1, kHid, kHid, 16, 8, 0, pw + o.w2h, wmax);
  hipLaunchKernelGGL(k_pack_a_split<true>, dim3(3 * 8 * 2), dim3(256), 0, st, m.w2, kHid, kCtrl, 3, 8, 0, pw + o.w3h, wmax);
  return launch_status();
}

// partial = false: the whole buffer (zeroed first; the four networks every checkpoint has are required).
// partial = true:  only the networks whose six pointers are all there are packed again, in place (an optimiser step changed
//                  them); the max |w| word of a re-packed chain is recomputed, the sticky domain word is left alone.
static int pack_impl(const pstl_weight_ptrs* w, float* packed, hipStream_t st, bool partial) {
  const PackLayout L = make_layout();
  if (!partial && hipMemsetAsync(packed, 0, L.total * sizeof(float), st) != hipSuccess) return PSTL_ERR_LAUNCH;
  const pstl_mlp3* encs[3] = {&w->ego_encoder, &w->neighbor_encoder, &w->lane_encoder};
  for (int e = 0; e < 3; ++e) {   // fp32 MFMA A operands (k_enc_gemm)
    const pstl_mlp3& m = *encs[e];
    if (!mlp_ok(m)) continue;
    const EncOff& o = L.enc[e];
    hipLaunchKernelGGL(k_pack_a, dim3(16 * enc_k16(e)), dim3(256), 0, st, m.w0, enc_in(e), kHid, 16, enc_k16(e), 0,
                       packed + o.a0, enc_in(e));
    hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, st, m.b0, kHid, kHid, packed + o.b0);
    hipLaunchKernelGGL(k_pack_a, dim3(16 * 16), dim3(256), 0, st, m.w1, kHid, kHid, 16, 16, 0, packed + o.a1, kHid);
    hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, st, m.b1, kHid, kHid, packed + o.b1);
    hipLaunchKernelGGL(k_pack_a, dim3(2 * 16), dim3(256), 0, st, m.w2, kHid, 32, 2, 16, 0, packed + o.a2, kHid);
    hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, st, m.b2, 32, 32, packed + o.b2);
    if (int err = launch_status()) return err;
  }
  unsigned* status = reinterpret_cast<unsigned*>(packed + L.status);   // (zeroed by the memset above when !partial)
  auto zero_word = [&](int i) {   // (k_copy with no source elements writes the padding value: 0)
    hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, st, (const float*)packed, 0, 1, packed + L.status + i);
  };
  if (mlp_ok(w->policy_net)) {
    if (partial) zero_word(0);
    if (int err = pack_chain(w->policy_net, 303, 1, true, L.pol, packed, status + 0, st)) return err;
  }
  if (mlp_ok(w->rect_net)) {
    if (partial) zero_word(1);
    if (int err = pack_chain(w->rect_net, 271, 2, false, L.rect, packed, status + 1, st)) return err;
  }
  if (mlp_ok(w->merge_net)) {
    const long off6[6] = {L.mrg.w0t, L.mrg.b0, L.mrg.w1t, L.mrg.b1, L.mrg.w2t, L.mrg.b2};
    if (int err = pack_mlp_t(w->merge_net, 40, 32, 40, off6, packed, st)) return err;
  }
  return PSTL_OK;
}

extern "C" int pstl_pack_weights(const pstl_weight_ptrs* w, float* packed, void* stream) {
  if (!w || !packed) return PSTL_ERR_ARG;
  if (!mlp_ok(w->ego_encoder) || !mlp_ok(w->neighbor_encoder) || !mlp_ok(w->lane_encoder) || !mlp_ok(w->policy_net))
    return PSTL_ERR_ARG;
  return pack_impl(w, packed, as_stream(stream), false);
}

extern "C" int pstl_repack_weights(const pstl_weight_ptrs* w, float* packed, void* stream) {
  if (!w || !packed) return PSTL_ERR_ARG;
  return pack_impl(w, packed, as_stream(stream), true);
}

extern "C" int pstl_fill_normal(const pstl_cfg* cfg, int step, float* out, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!out || step < 0) return PSTL_ERR_ARG;
  const long N = n_rows(cfg);
  hipLaunchKernelGGL(k_fill_normal, dim3((unsigned)((N * 10 + 255) / 256)), dim3(256), 0, as_stream(stream), N,
                     (unsigned long long)cfg->seed, cfg->dyn, (long)cfg->row_offset, step, out);
  return launch_status();
}

extern "C" int pstl_time_bias(const float* packed, int steps, float* tbias, void* stream) {
  if (!packed || !tbias || steps < 1) return PSTL_ERR_ARG;
  const PackLayout L = make_layout();
  hipLaunchKernelGGL(k_time_bias, dim3(steps), dim3(kHid), 0, as_stream(stream), packed + L.pol.w1t, steps, tbias);
  return launch_status();
}

extern "C" size_t pstl_encode_scene_work_floats(const pstl_cfg* cfg) {
  if (check_cfg(cfg)) return 0;
  return (size_t)cfg->bs * (cfg->K + 4) * (48 + 2 * kHid + 32);
}

static int encode_scene(const pstl_cfg* cfg, const float* packed, const float* ego0, const float* neighbors,
                        const float* currlane, const float* leftlane, const float* rightlane, const float* curr_id,
                        const float* left_id, const float* right_id, float* feature, float* base_policy, float* base_rect,
                        float* tok_in, float* tok_h1, float* tok_h2, float* tok_out, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!packed || !ego0 || !neighbors || !currlane || !leftlane || !rightlane || !curr_id || !left_id || !right_id ||
      !feature || !base_policy || !tok_in || !tok_h1 || !tok_h2 || !tok_out)
    return PSTL_ERR_ARG;
  hipStream_t st = as_stream(stream);
  const PackLayout L = make_layout();
  const long bs = cfg->bs, K = cfg->K, T = bs * (K + 4);
  hipLaunchKernelGGL(k_tokens, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, (int)bs, (int)K, ego0, neighbors, currlane,
                     leftlane, rightlane, curr_id, left_id, right_id, tok_in);
  const long t0[3] = {0, bs, bs * (K + 1)}, tn[3] = {bs, bs * K, 3 * bs};
  const int cus = cu_count();
  // one launch per layer and input width; the workgroups of a launch are shared out by tile count
  auto launch = [&](int layer, int e_lo, int e_hi) {
    EncGemmArgs a = {};
    const bool rowsplit = layer == 2;
    long unit[3] = {0, 0, 0}, total = 0;   // tiles (layer 2: batches of 8 tiles, one per wave) to hand out
    for (int e = e_lo; e < e_hi; ++e) {
      const long tiles = (tn[e] + 15) / 16;
      unit[e - e_lo] = rowsplit ? (tiles + 7) / 8 : tiles;
      total += unit[e - e_lo];
    }
    int blk = 0;
    for (int e = e_lo; e < e_hi; ++e) {
      EncGemmProb& p = a.p[e - e_lo];
      const long units = unit[e - e_lo];
      long nb = units * cus / total;     // rounded down: one workgroup more than there are CUs would run as a second wave
      if (nb > units) nb = units;
      if (nb < 1) nb = 1;
      const EncOff& o = L.enc[e];
      p.X = (layer == 0 ? tok_in + t0[e] * 48 : layer == 1 ? tok_h1 + t0[e] * kHid : tok_h2 + t0[e] * kHid);
      p.ldx = layer == 0 ? 48 : kHid;
      p.A = packed + (layer == 0 ? o.a0 : layer == 1 ? o.a1 : o.a2);
      p.bias = packed + (layer == 0 ? o.b0 : layer == 1 ? o.b1 : o.b2);
      p.out = (layer == 0 ? tok_h1 + t0[e] * kHid : layer == 1 ? tok_h2 + t0[e] * kHid : tok_out + t0[e] * 32);
      p.ldo = layer == 2 ? 32 : kHid;
      p.rows = tn[e];
      p.relu = layer < 2;
      p.blk0 = blk;
      p.nblk = (int)nb;
      blk += (int)nb;
    }
    a.np = e_hi - e_lo;
    if (layer == 0 && e_lo == 2)
      hipLaunchKernelGGL((k_enc_gemm<3, false>), dim3(blk), dim3(512), 0, st, a);
    else if (layer == 0)
      hipLaunchKernelGGL((k_enc_gemm<1, false>), dim3(blk), dim3(512), 0, st, a);
    else if (layer == 1)
      hipLaunchKernelGGL((k_enc_gemm<16, false>), dim3(blk), dim3(512), 0, st, a);
    else
      hipLaunchKernelGGL((k_enc_gemm<16, true>), dim3(blk), dim3(512), 0, st, a);
  };
  launch(0, 0, 2);   // ego + neighbours: 16 padded input columns
  launch(0, 2, 3);   // lanes: 48
  launch(1, 0, 3);
  launch(2, 0, 3);
  hipLaunchKernelGGL(k_feature_base, dim3((unsigned)((bs + kFbScn - 1) / kFbScn), base_rect ? 2 : 1), dim3(256), 0, st, (int)bs, (int)K,
                     (const float*)tok_out, packed, L.pol, L.rect, feature, base_policy, base_rect);
  return launch_status();
}

extern "C" int pstl_encode_scene(const pstl_cfg* cfg, const float* packed, const float* ego0, const float* neighbors,
                                 const float* currlane, const float* leftlane, const float* rightlane,
                                 const float* curr_id, const float* left_id, const float* right_id, float* work,
                                 float* feature, float* base_policy, float* base_rect, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!work) return PSTL_ERR_ARG;
  const long T = (long)cfg->bs * (cfg->K + 4);
  float* tok_in = work;
  float* tok_h1 = tok_in + T * 48;
  float* tok_h2 = tok_h1 + T * kHid;
  float* tok_out = tok_h2 + T * kHid;
  return encode_scene(cfg, packed, ego0, neighbors, currlane, leftlane, rightlane, curr_id, left_id, right_id, feature,
                      base_policy, base_rect, tok_in, tok_h1, tok_h2, tok_out, stream);
}

extern "C" int pstl_encode_scene_saved(const pstl_cfg* cfg, const float* packed, const float* ego0, const float* neighbors,
                                       const float* currlane, const float* leftlane, const float* rightlane,
                                       const float* curr_id, const float* left_id, const float* right_id, float* feature,
                                       float* base_policy, float* base_rect, float* tok_in, float* tok_h1, float* tok_h2,
                                       float* tok_out, void* stream) {
  return encode_scene(cfg, packed, ego0, neighbors, currlane, leftlane, rightlane, curr_id, left_id, right_id, feature,
                      base_policy, base_rect, tok_in, tok_h1, tok_h2, tok_out, stream);
}

extern "C" int pstl_rollout(const pstl_cfg* cfg, float* packed, const float* base_policy, const float* tbias,
                            const float* stlp, const float* hl, const float* beta, const float* alpha,
                            const float* alpha_hat, const float* noise, int step_hi, int step_lo, int mu_only,
                            float* x_inout, float* emit_out, int n_emit, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!packed || !base_policy || !tbias || !stlp || !hl || !beta || !alpha || !alpha_hat || !x_inout) return PSTL_ERR_ARG;
  if (step_lo < 1 || step_hi < step_lo || step_hi > cfg->steps - 1) return PSTL_ERR_ARG;
  if (mu_only && step_hi != step_lo) return PSTL_ERR_ARG;
  if (n_emit < 0 || n_emit > cfg->steps || (n_emit > 0 && !emit_out)) return PSTL_ERR_ARG;
  ChainArgs a = {};
  a.N = n_rows(cfg);
  a.plan_N = cfg->plan_rows > 0 ? (long)cfg->plan_rows : 0;
  if (a.N >= (1L << 31)) return PSTL_ERR_SHAPE;   // row indices are 32-bit inside the kernel (a shard of 2^31 rows is 344 GB)
  a.tiles_per_group = tiles_per_group(a.N);
  a.rows_per_scene = cfg->rows_per_scene;
  a.steps = cfg->steps;
  a.step_hi = step_hi;
  a.step_lo = step_lo;
  a.mu_only = mu_only;
  a.n_emit = n_emit;
  a.clip = (cfg->flags & PSTL_FLAG_CLIP) ? 1 : 0;
  a.w_max = cfg->w_max;
  a.a_max = cfg->a_max;
  a.off = make_layout().pol;
  a.packed = packed;
  a.base = base_policy;
  a.tbias = tbias;
  a.stlp = stlp;
  a.hl = hl;
  a.beta = beta;
  a.alpha = alpha;
  a.alpha_hat = alpha_hat;
  a.noise = noise;
  a.rng = (cfg->flags & PSTL_FLAG_RNG) ? 1 : 0;
  a.seed = cfg->seed;
  a.seed_dev = cfg->dyn ? reinterpret_cast<const unsigned long long*>(&cfg->dyn->seed) : nullptr;
  a.row_offset = (long)cfg->row_offset;
  a.x_inout = x_inout;
  a.emit_out = emit_out;
  a.status = reinterpret_cast<unsigned*>(packed + make_layout().status) + 2;
  // one launch covers at most kMaxLaunchSteps reverse steps (the per-step coefficients sit in LDS)
  for (int hi = step_hi; hi >= step_lo; hi -= kMaxLaunchSteps) {
    a.step_hi = hi;
    a.step_lo = (hi - kMaxLaunchSteps + 1 > step_lo) ? hi - kMaxLaunchSteps + 1 : step_lo;
    if (int e = launch_chain_nw<false>(cfg->chain_waves, a, as_stream(stream))) return e;
  }
  return PSTL_OK;
}

extern "C" int pstl_rollout_layout(const pstl_cfg* cfg, int multi_step, int* kernel, int* tiles, int* rounds) {
  if (int e = check_cfg(cfg)) return e;
  if (!kernel || !tiles || !rounds) return PSTL_ERR_ARG;
  const long N = n_rows(cfg), n_tiles = (N + kTileRows - 1) / kTileRows, cus = cu_count();
  const int cw = cfg->chain_waves;
  ChainArgs a = {};
  a.N = N, a.rows_per_scene = cfg->rows_per_scene, a.step_hi = multi_step ? 2 : 1, a.step_lo = 1, a.mu_only = multi_step ? 0 : 1;
  const bool ut = cfg->rows_per_scene % kTileRows == 0;
  int k = 3, g = tiles_per_group(N);
  const long plan = cfg->plan_rows > 0 ? (long)cfg->plan_rows : N;
  if ((cw == 2 || (cw == 0 && chain2_pays(plan, !multi_step))) && chain2_eligible(a)) {
    k = 2, g = chain2_wg_rows(a) / kTileRows;
  } else if (cw == 0 || cw == 16 || cw == 2) {
    k = 1;
    const int sg = (cw == 0 && ut) ? sparse_tiles_per_group(N) : 0;
    if (sg > 0) k = 0, g = sg;
    else if (multi_step) g = tiles_per_group_balanced(N);
  }
  const long n_groups = (n_tiles + g - 1) / g;
  *kernel = k, *tiles = g, *rounds = (int)((n_groups + cus - 1) / cus);
  return PSTL_OK;
}

static int refine_impl(const pstl_cfg* cfg, float* packed, const float* base_rect, const float* stlp,
                       const float* hl, const float* init_controls, const float* scores, float* pooled_work,
                       float* out_controls, float* h1_save, float* h2_save, float* pre_save, void* stream);

extern "C" int pstl_refine(const pstl_cfg* cfg, float* packed, const float* base_rect, const float* stlp,
                           const float* hl, const float* init_controls, const float* scores, float* pooled_work,
                           float* out_controls, void* stream) {
  return refine_impl(cfg, packed, base_rect, stlp, hl, init_controls, scores, pooled_work, out_controls, nullptr, nullptr,
                     nullptr, stream);
}

extern "C" int pstl_refine_train_forward(const pstl_cfg* cfg, float* packed, const float* base_rect,
                                         const float* stlp, const float* hl, const float* init_controls,
                                         const float* scores, float* pooled_work, float* out_controls, float* h1_save,
                                         float* h2_save, float* pre_save, void* stream) {
  if (!h1_save || !h2_save || !pre_save) return PSTL_ERR_ARG;
  return refine_impl(cfg, packed, base_rect, stlp, hl, init_controls, scores, pooled_work, out_controls, h1_save, h2_save,
                     pre_save, stream);
}

static int refine_impl(const pstl_cfg* cfg, float* packed, const float* base_rect, const float* stlp,
                       const float* hl, const float* init_controls, const float* scores, float* pooled_work,
                       float* out_controls, float* h1_save, float* h2_save, float* pre_save, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!packed || !base_rect || !stlp || !hl || !init_controls || !scores || !out_controls) return PSTL_ERR_ARG;
  const bool merge = !(cfg->flags & PSTL_FLAG_NO_MERGE);
  if (merge) {
    if (!pooled_work) return PSTL_ERR_ARG;
    if (cfg->rows_per_scene != 3 * cfg->S || cfg->n_shards < 1 || cfg->S % cfg->n_shards != 0 || cfg->n_shards * 40 > 256)
      return PSTL_ERR_SHAPE;
  }
  hipStream_t st = as_stream(stream);
  const PackLayout L = make_layout();
  if (merge) {
    MergeArgs m;
    m.bs = cfg->bs;
    m.S = cfg->S;
    m.n_shards = cfg->n_shards;
    m.off = L.mrg;
    m.packed = packed;
    m.init = init_controls;
    m.pooled = pooled_work;
    hipLaunchKernelGGL(k_merge_pool, dim3((unsigned)((long)cfg->bs * 3)), dim3(64), 0, st, m);
    if (int e = launch_status()) return e;
  }
  ChainArgs a = {};
  a.N = n_rows(cfg);
  a.plan_N = cfg->plan_rows > 0 ? (long)cfg->plan_rows : 0;
  if (a.N >= (1L << 31)) return PSTL_ERR_SHAPE;   // row indices are 32-bit inside the kernel (a shard of 2^31 rows is 344 GB)
  a.tiles_per_group = tiles_per_group(a.N);
  a.rows_per_scene = cfg->rows_per_scene;
  a.steps = cfg->steps;
  a.clip = (cfg->flags & PSTL_FLAG_CLIP_RECT) ? 1 : 0;
  a.w_max = cfg->w_max;
  a.a_max = cfg->a_max;
  a.off = L.rect;
  a.packed = packed;
  a.base = base_rect;
  a.stlp = stlp;
  a.hl = hl;
  a.init = init_controls;
  a.pooled = merge ? pooled_work : nullptr;
  a.scores = scores;
  a.out = out_controls;
  a.S = cfg->S;
  a.n_shards = cfg->n_shards;
  a.h1_save = h1_save;
  a.h2_save = h2_save;
  a.pre_save = pre_save;
  a.status = reinterpret_cast<unsigned*>(packed + L.status) + 2;
  return launch_chain_nw<true>(cfg->chain_waves, a, st);
}

// ---- merge_net backward (training with --joint) ------------------------------------------------------------------
constexpr int kMergeBwdBlocks = 1024;

extern "C" size_t pstl_merge_backward_work_floats(const pstl_cfg* cfg) {
  if (check_cfg(cfg)) return 0;
  return (size_t)kMergeBwdBlocks * kMrgFloats;
}

extern "C" int pstl_merge_backward(const pstl_cfg* cfg, const float* packed, const float* init_controls,
                                   const float* dfused, int ldf, float* work, float* dw0, float* db0, float* dw1,
                                   float* db1, float* dw2, float* db2, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!packed || !init_controls || !dfused || !work || !dw0 || !db0 || !dw1 || !db1 || !dw2 || !db2) return PSTL_ERR_ARG;
  if ((cfg->flags & PSTL_FLAG_NO_MERGE) || ldf < kCtrl) return PSTL_ERR_ARG;
  if (cfg->rows_per_scene != 3 * cfg->S || cfg->n_shards < 1 || cfg->S % cfg->n_shards != 0 || cfg->n_shards * 40 > 256 ||
      cfg->S > 64)
    return PSTL_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  MergeBwdArgs m;
  m.bs = cfg->bs;
  m.S = cfg->S;
  m.n_shards = cfg->n_shards;
  m.off = make_layout().mrg;
  m.packed = packed;
  m.init = init_controls;
  m.dfused = dfused;
  m.ldf = ldf;
  m.slabs = work;
  const long pairs = (long)cfg->bs * 3;
  const int nb = (int)(pairs < kMergeBwdBlocks ? pairs : kMergeBwdBlocks);
  hipLaunchKernelGGL(k_merge_bwd, dim3(nb), dim3(64), 0, st, m);
  hipLaunchKernelGGL(k_merge_reduce, dim3((kMrgFloats + 255) / 256), dim3(256), 0, st, nb, (const float*)work, dw0, db0, dw1,
                     db1, dw2, db2);
  return launch_status();
}
