// chain2_kernels.hip -- k_chain2<RNG, MODE, RT>: the MLP chain of large batches with the ROWS stationary and the WEIGHTS
// streamed -- MODE 0: the multi-step denoiser launch (policy_net, reverse diffusion A3-A5); 1: the guided phase's single-step
// launches (mu only), one workgroup per CU walking the tiles; 2: RefineNet's inference pass (A11) in that tile-walking form.
// Same arithmetic as k_chain's default form (mlp_kernels.hip: every fp32 operand as two IEEE-half pieces, three
// v_mfma_f32_16x16x32_f16 products per fp32 product, fp32 accumulation), other dataflow (described for RT = 4 row tiles per
// wave = 256 rows per workgroup; RT = 3 and 2 are built for batches those would leave CUs or rounds part empty):
//
//  * One workgroup = 4 waves, one per SIMD (up to 512 registers each), owns 256 rows for ALL reverse steps of the launch.
//    A wave owns 64 rows (four 16-row tiles) and all 256 hidden features.  The ReLU'd, split layer-1 output of its rows --
//    8 k-blocks x 4 row tiles x (hi | lo) = 256 registers -- stays in the accumulation half of its register file as the B
//    operands of layer 2; a layer's accumulators, ReLU'd and split, ARE the next layer's B operands (the permuted-k
//    hand-over of k_chain), so no activation crosses LDS and no barrier sits between the layers.  Layer 2 is walked in 8
//    chunks of 32 output features; a finished chunk is one k-block of layer 3; layer 3's output (+ the DDPM update) is, in
//    the same lane layout, the next step's layer-1 input.  x never leaves the wave.
//  * The 368 KB of split weights stream L2 -> LDS by LDS-DMA (1 KB per wave-instruction = one (tile, k-block, piece) block of
//    the packed buffer) through a ring of three 22 KB slots, 20 phases per tile-step (4 of layer 1, 16 of layers 2 + 3), ONE
//    barrier per phase placed in the middle of it.  Every A operand is read once per wave and feeds 12 MFMAs.
//  * The instruction stream is laid out by hand: one slot = one MFMA + what goes into its shadow, closed by
//    sched_barrier(0) (the compiler allocates registers, counts LDS waits and pads hazards; left to choose the order it
//    puts a chunk's ~170 conversion instructions in front of the MFMAs they should hide under).
//
// Reference: nusc_model.py:97-180 (Net.forward, diffusion branch), nusc_train.py:557-587,628-655 (diffusion_rollout),
// nusc_model.py:182-235 (Net.rect_forward).
// Built with -mllvm -amdgpu-mfma-vgpr-form (accumulators in the architectural half: the accumulation half is full).
#include <type_traits>
#include <utility>

#include "chain_args.hpp"
#include "pstl_common.hpp"
#include "rng.hpp"

namespace pstl {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int uw4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// RT = 16-row tiles per wave (template parameter of the kernel): 4 -- 256 rows per workgroup -- unless fewer fill the chip better
// (chain2_wg_rows: 3 for some multi-step and tile-walking launches, 2 for tile-walking launches of 20 480 ... 40 960 rows)
constexpr int kCtrl2 = 40, kHid2 = 256;
constexpr int kSlotBytes = 22 * 1024, kRing = 3;
constexpr int kMaxScn = 7;                 // scenes a workgroup's 256 rows may touch (rows_per_scene >= 48)
// LDS carve (bytes)
template <int RT>
struct Carve {
  static constexpr int kWgRows = 4 * 16 * RT;                           // rows per workgroup
  static constexpr int kOffXq = kRing * kSlotBytes;                     // [4 waves][RT][3][64] f32x4: the state x, lane-private
  static constexpr int kOffCrow = kOffXq + 4 * RT * 3 * 64 * 16;        // [kMaxScn][256] (base[scene] + tbias[step]) x kAcc
  static constexpr int kOffBrow = kOffCrow + kMaxScn * 1024;            // [kMaxScn][256] base[scene]
  static constexpr int kOffB2 = kOffBrow + kMaxScn * 1024;              // [256] b2 x kAcc
  static constexpr int kOffB3 = kOffB2 + 1024;                          // [48]  b3 x kAcc
  static constexpr int kOffCoef = kOffB3 + 192;                         // [kMaxLaunchSteps][4] kk, a, sb, 0
  static constexpr int kOffRc = kOffCoef + kMaxLaunchSteps * 16;        // MU: [4 waves][RT][4][64] the next tile's hl | stlp words (lanes g >= 2)
  static constexpr int kLdsBytes = kOffRc + 4 * RT * 4 * 64 * 4;
};

constexpr float kSX = kSplitX, kSW = kSplitW, kAcc = kSX * kSW, kInvSW = 1.0f / kSW, kInvAcc = 1.0f / kAcc;

#define FENCE() __builtin_amdgcn_sched_barrier(0)
// Timing-only ablations (tools/dbg/build_variants2.sh + time_variants.py; the results are garbage): bits 1 noise steps,
// 2 layer-1 conversions + pins, 4 layer-2 conversions, 8 epilogue, 16 LDS-DMA, 32 phase barriers, 64 the tail's conversion,
// 128 the domain guard's running maximum, 256 layer 1 altogether, 512 layer 3, 1024 the wait for the LDS-DMA in front of
// the phase barriers (the DMA is still issued)
#ifndef PSTL_C2_ABL
#define PSTL_C2_ABL 0
#endif
// -DPSTL_C2_STAMP: a diagnostic build that accumulates, per wave of workgroup 7, the shader cycles of the sections of a
// tile-step (layer 1 | chunk 0 | chunk 1 | first halves of chunks 2..5 | second halves | chunks 6, 7 | tail | epilogue) and the launch's realtime span, written as 64-bit words
// to the buffer passed as emit_out (n_emit must be 0): [wave][8 sections, cycles total, realtime ticks (100 MHz), steps]
// (tools/dbg/chain2_stamps.py).  The stamps' waits drain the LDS reads in flight: read the SHARES, not the length.
#ifdef PSTL_C2_STAMP
#define C2_STAMP(k)                                                                                       \
  {                                                                                                       \
    FENCE();                                                                                              \
    unsigned long long t_;                                                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
    st_sum[k] += t_ - st_prev;                                                                            \
    st_prev = t_;                                                                                         \
    FENCE();                                                                                              \
  }
#else
#define C2_STAMP(k)
#endif
// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  Every index is a
// constant expression by construction (a run-time loop the unroller gives up on puts the arrays it indexes in scratch).
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

// one 1 KB LDS-DMA piece: lane l moves 16 bytes from sbase + voff to LDS byte address lds_dst + 16 l.  Issued from inline
// assembly: invisible to the compiler's wait-count pass (a tracked LDS-DMA puts s_waitcnt vmcnt(0) in front of the wave's
// next ds_read), waited for by the vmcnt(0) in front of each phase barrier.
__device__ __forceinline__ void dma16x2(const void* sbase, unsigned voff, unsigned lds_dst) {   // two adjacent pieces
  if (PSTL_C2_ABL & 16) return;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff),
               "s"(sbase), "s"(lds_dst)
               : "memory");
}

// four adjacent pieces behind ONE M0 write: issuing a vector memory instruction costs the wave ~40 cycles, the M0 write ~15 more,
// each further instruction of a burst next to nothing (tools/dbg/dma_issue_cost.hip, profiles/r5/dma_issue_cost.txt)
__device__ __forceinline__ void dma16x4(const void* sbase, unsigned voff, unsigned lds_dst) {
  if (PSTL_C2_ABL & 16) return;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff),
               "s"(sbase), "s"(lds_dst)
               : "memory");
}
// MU: single pieces of the next tile's state -- 16 bytes per lane from sbase + voff, 4 bytes per lane from a per-lane pointer
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const float* p, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(p), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// (The domain guard of the split-f16 arithmetic -- see note_pieces in mlp_kernels.hip -- is a running maximum of the hi pieces,
// kept by an asm statement in conv_step: written as a max() chain the optimiser re-associates it and sinks the whole chain
// behind the phases, keeping -- spilling -- every hi word until then.)
// the lo pieces of two values: f16(v k - hi) for both halves of the packed hi word (inline assembly: hipcc selects
// v_cvt_f32_f16 x 2 + v_pk_fma_f32 + v_cvt_pk_f16_f32 for the C++ form; consumers are many instructions away, no hazard)
__device__ __forceinline__ unsigned lo_word(float v0, float v1, float k, unsigned hi) {
  unsigned lw;
  asm volatile("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
               : "=&v"(lw)
               : "v"(v0), "v"(v1), "s"(k), "v"(hi));
  return lw;
}
__device__ __forceinline__ bool pieces_overflowed(unsigned ovf) { return (ovf & 0xffffu) >= 0x7c00u || (ovf >> 16) >= 0x7c00u; }
// A value the optimiser must take as it comes at this point: stops it from hoisting `uniform pointer + lane offset` out of the
// step loop as per-lane 64-bit pointers (twelve of them were spilled and reloaded from scratch in every epilogue, each reload a
// memory round trip in a wave that has nothing else to issue)
__device__ __forceinline__ unsigned here(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ float clip_keep_nan(float v, float m) { return v < -m ? -m : (v > m ? m : v); }

// Phases of a tile-step and what their ring slot holds (1 KB pieces; the A operand of tile t, piece hl of the phase's
// k-block kq is piece pidx(kq, 2 t + hl) = (2 t + (kq >> 1)) * 4 + (kq & 1) * 2 + hl: the four pieces a wave fetches -- those
// of one tile and two k-blocks, adjacent in the packed buffer -- are adjacent in the slot as well, one burst behind one M0):
//   P0..P3   layer 1, phase q: chunks 2q, 2q+1 (kq = 2 cl + kb); P0 also carries W3's k-block 7 (pieces 16..21) for the
//            tail of the step before
//   P4+2c    layers 2/3, chunk c, k-blocks 0..3; for c >= 1 also W3's k-block c-1 (pieces 16..21)
//   P5+2c    chunk c, k-blocks 4..7
//
// MU (mu_only = 1, one reverse step per launch: the guided phase): the same tile-step, but the loop walks TILES instead of
// reverse steps -- one workgroup per CU takes tiles blockIdx.x, + gridDim.x, ... so that the weight stream, the tables and the
// launch are paid once per CU and not once per 256 rows.  The next tile's state arrives by LDS-DMA straight into the wave's
// lane-private image (issued in the tail, when the current tile's image has been read), its scene rows in the middle of
// chunk 6's phases.
//
// MODE 2 (REF): RefineNet's inference pass (Net.rect_forward, nusc_model.py:182-235) in the same tile-walking form -- rect_net's
// weights, the scene rows without a timestep row, the input init + pooled[scene][mode][shard] (merge_net's max-pool, k_merge_pool),
// and the tanh interval head instead of the DDPM update; the image keeps init, which the head needs again.
template <bool RNG, int MODE, int RT>
__global__ __launch_bounds__(256) void k_chain2(ChainArgs a) {
  constexpr bool MU = MODE != 0, REF = MODE >= 2, SAVE = MODE == 3;   // (3: the training forward pass -- REF + saved activations)
  static_assert(RT == 4 || RT == 3 || (RT == 2 && MU && !RNG), "see noise_l3 / noise_b: chunk pair p draws row tile p's noise");
  typedef Carve<RT> C;
  constexpr int kWgRows = C::kWgRows, kOffXq = C::kOffXq, kOffCrow = C::kOffCrow, kOffBrow = C::kOffBrow, kOffB2 = C::kOffB2,
                kOffB3 = C::kOffB3, kOffCoef = C::kOffCoef, kOffRc = C::kOffRc;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, col = lane & 15;
  constexpr int NM = 6 * RT;          // MFMAs per k-block
  constexpr int NCONV = 24 * RT;      // conversion micro-steps per chunk (4 RT pairs x 6)
  constexpr int kLead = 4;            // slots between an LDS read of a constant part and the MFMA that takes it as C
  long wg_row0 = (long)blockIdx.x * kWgRows;
  long row0 = wg_row0 + (long)w * (16 * RT);                 // first row of this wave
  const int nsteps = MU ? 1 : a.step_hi - a.step_lo + 1;
  const long n_tiles = (a.N + kWgRows - 1) / kWgRows;
  const int n_iter = MU ? (int)((n_tiles - 1 - (long)blockIdx.x) / (long)gridDim.x) + 1 : nsteps;   // tile-steps of this workgroup

  float* crow = reinterpret_cast<float*>(smem + kOffCrow);
  float* brow = reinterpret_cast<float*>(smem + kOffBrow);
  float* b2s = reinterpret_cast<float*>(smem + kOffB2);
  float* b3s = reinterpret_cast<float*>(smem + kOffB3);
  f32x4* coef = reinterpret_cast<f32x4*>(smem + kOffCoef);
  float* tbs = reinterpret_cast<float*>(smem + kOffCoef + 1024);   // MU (one coefficient entry): the launch's timestep row
  f32x4* xq = reinterpret_cast<f32x4*>(smem + kOffXq) + (w * RT * 3) * 64 + lane;   // [rt][j] at (rt*3 + j)*64

  // ---- scenes of this workgroup's rows; per-row-tile scene slot (a 16-row tile lies in one scene: rows_per_scene % 16 == 0) ----
  const long last_row = a.N - 1;
  long scene_first = wg_row0 / a.rows_per_scene;
  int scn[RT];
  auto set_scn = [&] {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      long r = row0 + 16 * rt;
      if (r > last_row) r = last_row;
      scn[rt] = (int)(r / a.rows_per_scene - scene_first);
    }
  };
  set_scn();

  // ---- tables ----
  {
    b2s[tid] = a.packed[a.off.b2 + tid] * kAcc;
    if (tid < 48) b3s[tid] = a.packed[a.off.b3 + tid] * kAcc;
    if (!REF && tid < nsteps) {   // reverse-step coefficients (nusc_train.py:580-587): x' = a x + sb z - kk (eps_net + b3),  a = (1 - c1) / sqrt(alpha)
      const int i = a.step_hi - tid;
      const float al = a.alpha[i], ah = a.alpha_hat[i], be = a.beta[i];
      const float c1 = (1.0f - al) / sqrtf(1.0f - ah), inv_sa = 1.0f / sqrtf(al);
      const float kk = inv_sa * c1;
      const bool noisy = !MU && i > 1 && (RNG || a.noise);   // the reference adds zeros at the last step
      coef[tid] = f32x4{kk, inv_sa - kk, noisy ? sqrtf(be) : 0.0f, 0.0f};
    }
    const float tb0 = REF ? 0.0f : a.tbias[(long)a.step_hi * kHid2 + tid];   // (rect_net has no timestep columns)
    if (MU) tbs[tid] = tb0;
    const long nscn_rows = (wg_row0 + kWgRows - 1 > last_row ? last_row : wg_row0 + kWgRows - 1);
    const int nscn = (int)(nscn_rows / a.rows_per_scene - scene_first) + 1;
    for (int s = 0; s < kMaxScn; ++s) {
      const int ss = s < nscn ? s : nscn - 1;
      const float bvv = a.base[(scene_first + ss) * kHid2 + tid];
      brow[s * 256 + tid] = bvv;
      crow[s * 256 + tid] = (bvv + tb0) * kAcc;
    }
  }
  const unsigned long long seed = a.seed_dev ? uniform_u64(a.seed_dev) : a.seed;
  unsigned ovf = 0;
  if (tid == 0 && blockIdx.x == 0) {   // the weights themselves: max |w| as the packer recorded it (status word 0 = policy_net)
    const float wm = reinterpret_cast<const float*>(a.status)[REF ? -1 : -2];
    if (!(wm < PSTL_SPLIT_F16_WMAX)) atomicOr(a.status, 1u);
  }

  // ---- state ----
  f16x8 bh[8][RT], bl[8][RT];         // h1 pieces: B operands of layer 2 (accumulation registers)
  f16x8 xh[2][RT], xl[2][RT];         // x pieces: B operands of layer 1
  f32x4 acc3[3][RT];
  f32x4 accA[2][RT], accB[2][RT];     // chunk accumulators (layer 1 and layer 2), even / odd chunk
  unsigned phw[RT][4], plw[RT][4];    // pieces of the finished chunk, as packed words
  f16x8 ah[2], al[2], nh[2], nl[2];   // A operands of the current / the next k-block
  f16x8 w3h[2], w3l[2];               // layer 3's A operands, double buffered over j
  // (What a chunk's accumulators start from -- layer 1: the scene / timestep part, layer 2: the bias, layer 3: b3 -- is read
  // from LDS STRAIGHT INTO the accumulator registers, kLead slots or more ahead, and every MFMA accumulates in place.  With a
  // separate C operand the registers it dies in are free the moment the MFMA has issued, while the MFMA is still reading them:
  // an inline-assembly statement with a fresh output register right behind it -- the conversions' -- could be given exactly
  // those, and the compiler pads that hazard for its own instructions only.)
  f32x2 cm[2], cf[2];                 // conversion state of the two pairs in flight
  unsigned chw[2];

  // the layer-1 B operands of row tile rt from its state quads: k-block 0 = output tiles j = 0, 1; k-block 1 = tile j = 2
  // (x 32..39 | hl | stlp) and zeros.  Per pair of values: v_pk_mul_f32 (x 16), v_cvt_pk_f16_f32, the running maximum of
  // |16 x| for the domain guard (x is signed: v_max3_f32 with |.| modifiers), and the fused lo word.
  float ovfx = 0.0f;
  auto x_pair = [&](float v0, float v1, unsigned& hw, unsigned& lw) {
    const f32x2 m = f32x2{v0, v1} * kSX;
    hw = __builtin_bit_cast(unsigned, __builtin_convertvector(m, f16x2));
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ovfx) : "v"(m[0]), "v"(m[1]));
    lw = lo_word(v0, v1, kSX, hw);
  };
  auto make_x_pieces = [&](int rt, const f32x4& q0, const f32x4& q1, const f32x4& q2) {
    unsigned h0[4], l0[4], h1[2], l1[2];
    x_pair(q0[0], q0[1], h0[0], l0[0]);
    x_pair(q0[2], q0[3], h0[1], l0[1]);
    x_pair(q1[0], q1[1], h0[2], l0[2]);
    x_pair(q1[2], q1[3], h0[3], l0[3]);
    x_pair(q2[0], q2[1], h1[0], l1[0]);
    x_pair(q2[2], q2[3], h1[1], l1[1]);
    xh[0][rt] = __builtin_bit_cast(f16x8, uw4{h0[0], h0[1], h0[2], h0[3]}), xl[0][rt] = __builtin_bit_cast(f16x8, uw4{l0[0], l0[1], l0[2], l0[3]});
    xh[1][rt] = __builtin_bit_cast(f16x8, uw4{h1[0], h1[1], 0u, 0u}), xl[1][rt] = __builtin_bit_cast(f16x8, uw4{l1[0], l1[1], 0u, 0u});
  };

  // REF: the input is init + pooled[scene][mode][shard] (nusc_model.py:186-200); r = this lane's row
  auto add_pooled = [&](long r, f32x4& q0, f32x4& q1, f32x4& q2) {
    if constexpr (REF) {
      if (a.pooled) {
        const unsigned ru = (unsigned)r, rps = (unsigned)a.rows_per_scene;
        const unsigned b = ru / rps, rr = ru - b * rps, sm = rr / 3u, m = rr - 3u * sm, sh = sm / (unsigned)(a.S / a.n_shards);
        const float* pp = a.pooled + ((long)(b * 3u + m) * a.n_shards + sh) * kCtrl2 + 4 * g;
        q0 += *reinterpret_cast<const f32x4*>(pp);
        q1 += *reinterpret_cast<const f32x4*>(pp + 16);
        if (g < 2) q2 += *reinterpret_cast<const f32x4*>(pp + 32);
      }
    }
  };

  // ---- the initial state: lane (g, col) of row tile rt holds columns 16 j + 4 g .. + 3 of row 16 rt + col ----
  const bool own2 = g < 2;            // tile j = 2: lanes g < 2 hold x 32..39, lanes g >= 2 the row constants hl | stlp | 0
  {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      long r = row0 + 16 * rt + col;
      if (r > last_row) r = last_row;
      const float* xr = (REF ? a.init : a.x_inout) + r * kCtrl2;
      f32x4 q0 = *reinterpret_cast<const f32x4*>(xr + 4 * g);
      f32x4 q1 = *reinterpret_cast<const f32x4*>(xr + 16 + 4 * g);
      f32x4 q2;
      if (own2) {
        q2 = *reinterpret_cast<const f32x4*>(xr + 32 + 4 * g);
      } else {
        const float* sp = a.stlp + r * 6;
        q2 = g == 2 ? f32x4{a.hl[r], sp[0], sp[1], sp[2]} : f32x4{sp[3], sp[4], sp[5], 0.0f};
      }
      xq[(rt * 3 + 0) * 64] = q0;
      xq[(rt * 3 + 1) * 64] = q1;
      xq[(rt * 3 + 2) * 64] = q2;
      add_pooled(r, q0, q1, q2);
      make_x_pieces(rt, q0, q1, q2);
      if (!MU && a.n_emit >= a.steps && a.step_hi == a.steps - 1 && row0 + 16 * rt + col <= last_row) {   // x_T is entry 0 of the full list
        const f32x4 sc = f32x4{a.w_max, a.a_max, a.w_max, a.a_max};
        float* er = a.emit_out + ((long)(a.n_emit - a.steps) * a.N + r) * kCtrl2;
        f32x4 v0 = q0 * sc, v1 = q1 * sc, v2 = q2 * sc;
        if (a.clip) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v0[e] = clip_keep_nan(v0[e], sc[e]), v1[e] = clip_keep_nan(v1[e], sc[e]), v2[e] = clip_keep_nan(v2[e], sc[e]);
        }
        *reinterpret_cast<f32x4*>(er + 4 * g) = v0;
        *reinterpret_cast<f32x4*>(er + 16 + 4 * g) = v1;
        if (own2) *reinterpret_cast<f32x4*>(er + 32 + 4 * g) = v2;
      }
    }
  }

  // ---- the weight stream ----
  const unsigned* w1p = reinterpret_cast<const unsigned*>(a.packed + a.off.w1xh);
  const unsigned* w2p = reinterpret_cast<const unsigned*>(a.packed + a.off.w2h);
  const unsigned* w3p = reinterpret_cast<const unsigned*>(a.packed + a.off.w3h);
  const unsigned lane16 = (unsigned)lane * 16u;
  // this wave's share of a phase's W1 / W2 pieces: wave w = 2 t + kh fetches tile t of the phase, k-blocks 2 kh, 2 kh + 1, both
  // pieces each -- 4 KB that are contiguous in the packed buffer ([tile][k-block][hi | lo]) and in the slot.  KIND 0: layer-1
  // phase q = c (its four tiles 4 q + 2 cl + t, k-blocks kq = 2 cl + kb: wave w fetches tile 4 q + 2 (w & 1) + (w >> 1)); 1 / 2:
  // first / second half of layer-2 chunk c (tile 2 c + t, k-blocks 4 half + kq)
  auto issue_w = [&](auto kind_tag, int c, unsigned slot_byte) {
    constexpr int KIND = decltype(kind_tag)::value;
    const unsigned dst = slot_byte + (unsigned)w * 4096u;
    if constexpr (KIND == 0) {
      dma16x4(w1p + (long)(4 * c + 2 * (w & 1) + (w >> 1)) * 4 * 256, lane16, dst);
    } else {
      dma16x4(w2p + ((long)(2 * c + (w >> 1)) * 8 + 4 * (KIND - 1) + 2 * (w & 1)) * 2 * 256, lane16, dst);
    }
  };
  // pair w of the 3 (hi | lo) pairs of the W3 k-block c3 that rides with the phase (wave 3 repeats pair 0)
  auto issue_w3 = [&](int c3, unsigned slot_byte) {
    const int j = w < 3 ? w : 0;
    dma16x2(w3p + (long)(j * 8 + c3) * 2 * 256, lane16, slot_byte + 16384u + (unsigned)j * 2048u);
  };
  constexpr auto pidx = [](int kq, int m) { return (2 * (m >> 1) + (kq >> 1)) * 4 + (kq & 1) * 2 + (m & 1); };

  // prologue: phases P0 and P1 of the first tile-step
  issue_w(Ic<0>{}, 0, 0u);
  issue_w3(7, 0u);
  issue_w(Ic<0>{}, 1, (unsigned)kSlotBytes);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  unsigned s_cur = 0u, s_nxt = (unsigned)kSlotBytes, s_nn = 2u * (unsigned)kSlotBytes;   // slots of phases p, p + 1, p + 2
  const uw4* lbase = reinterpret_cast<const uw4*>(smem) + lane;
  auto rdA = [&](unsigned slot_byte, int piece) { return __builtin_bit_cast(f16x8, lbase[(slot_byte >> 4) + piece * 64]); };
  auto rd_bias = [&](int c, int t) { return *reinterpret_cast<const f32x4*>(b2s + 16 * (2 * c + t) + 4 * g); };
  auto rd_cst = [&](int c1, int t, int rt) { return *reinterpret_cast<const f32x4*>(crow + scn[rt] * 256 + 16 * (2 * c1 + t) + 4 * g); };

#pragma unroll
  for (int t = 0; t < 2; ++t) ah[t] = rdA(s_cur, pidx(0, t * 2)), al[t] = rdA(s_cur, pidx(0, t * 2 + 1));
#pragma unroll
  for (int u = 0; u < kLead; ++u) accA[u / RT][u % RT] = rd_cst(0, u / RT, u % RT);

  // ReLU + split of the finished chunk's accumulators S into the packed pieces, as 6 steps of 1-2 instructions per pair of
  // values, two pairs in flight: step i = 12 grp + 2 stage + which.  hi = f16(relu(v) / kSW) (v_pk_mul_f32 + v_cvt_pk_f16_f32),
  // lo = f16(relu(v) / kSW - hi) as ONE fused multiply-add per half (v_fma_mixlo / mixhi_f16: the product with a power of two
  // is exact, so this is the value the separate subtraction gives).
  // (the domain guard takes the hi words of both pairs in flight in one v_pk_maximum3_f16: halves >= +0, their order is the
  // integers'; infinity and NaN come out on top)
  auto conv_step = [&](auto& S, auto i_tag) {
    constexpr int i = decltype(i_tag)::value;
    constexpr int which = i & 1, stage = (i % 12) >> 1, p = 2 * (i / 12) + which;   // pair p = 4 rt + q
    constexpr int rt = p >> 2, q = p & 3, t = q >> 1, e = (q & 1) * 2, word = 2 * t + (q & 1);
    if constexpr (stage == 0) {
      const float v0 = S[t][rt][e];   // (a copy: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0)
      cm[which][0] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v0), 0));
    } else if constexpr (stage == 1) {
      const float v1 = S[t][rt][e + 1];
      cm[which][1] = __builtin_bit_cast(float, max(__builtin_bit_cast(int, v1), 0));
    } else if constexpr (stage == 2) {
      cf[which] = cm[which] * kInvSW;   // (behind an MFMA the compiler's late peephole takes two of three of these packed multiplies
                                        // apart into scalar ones; forcing them packed -- 100 instructions fewer -- changes nothing: measured)
    } else if constexpr (stage == 3) {
      chw[which] = __builtin_bit_cast(unsigned, __builtin_convertvector(cf[which], f16x2));
      phw[rt][word] = chw[which];
    } else if constexpr (stage == 4) {
      if constexpr (which == 1 && !(PSTL_C2_ABL & 128)) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(ovf) : "v"(chw[0]), "v"(chw[1]));
    } else {
      plw[rt][word] = lo_word(cm[which][0], cm[which][1], kInvSW, chw[which]);
    }
  };
  auto pieces_h = [&](int rt) { return __builtin_bit_cast(f16x8, uw4{phw[rt][0], phw[rt][1], phw[rt][2], phw[rt][3]}); };
  auto pieces_l = [&](int rt) { return __builtin_bit_cast(f16x8, uw4{plw[rt][0], plw[rt][1], plw[rt][2], plw[rt][3]}); };
  // the finished pieces of row tile rt become k-block KB of layer 2's B operand, in the accumulation half of the register file
  auto pin_h1 = [&](auto kb_tag, auto rt_tag, auto hl_tag) {
    constexpr int KB = decltype(kb_tag)::value, rt = decltype(rt_tag)::value, HL = decltype(hl_tag)::value;
    if constexpr (HL == 0) {
      bh[KB][rt] = pieces_h(rt);
      asm volatile("" : "+a"(bh[KB][rt]));
    } else {
      bl[KB][rt] = pieces_l(rt);
      asm volatile("" : "+a"(bl[KB][rt]));
    }
  };
  // conversion of a layer-1 chunk in the shadow of 48 slots: 3 steps in each of the first four slots, 2 in the others (96 in
  // 46 slots), and its pieces pinned as soon as a row tile's words are done (hi words: step 24 rt + 19, lo: 24 rt + 23)
  // SAVE: quad g = 2 rt + t of chunk c -- the fp32 outputs relu(acc) / kAcc of features 16 (2 c + t) + 4 g .. + 3 of this lane's
  // row of row tile rt -- goes to the activation buffer the backward pass reads (k_chain's save_hidden), in the slot in which its
  // conversion starts (conversion group g = steps 12 g ..: the accumulators are whole until the next chunk's preloads)
  auto save_quad = [&](float* buf, auto& S, int c, auto g_tag) {
    if constexpr (SAVE) {
      constexpr int gq = decltype(g_tag)::value, t = gq & 1, rt = gq >> 1;
      const unsigned ln = here((unsigned)lane);
      const unsigned trow0 = (unsigned)row0 + 16u * (unsigned)rt;
      if (trow0 + (ln & 15u) <= (unsigned)last_row) {
        const f32x4 v = S[t][rt];
        f32x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = fmaxf(v[e], 0.0f) * kInvAcc;
        *reinterpret_cast<f32x4*>((buf + (long)trow0 * kHid2) + ((ln & 15u) * (unsigned)kHid2 + (unsigned)(16 * (2 * c + t)) + 4u * (ln >> 4))) = h;
      }
    }
  };
  auto conv_h1_slot = [&](auto& S, auto kb_tag, auto slot_tag) {
    constexpr int slot = decltype(slot_tag)::value;
    constexpr int first = slot < 4 ? 3 * slot : 12 + 2 * (slot - 4), cnt = slot < 4 ? 3 : 2;
    static_for<cnt>([&](auto u) {
      constexpr int i = first + decltype(u)::value;
      if constexpr (SAVE && i < NCONV && i % 12 == 0) save_quad(a.h1_save, S, decltype(kb_tag)::value, Ic<i / 12>{});
      if constexpr (i < NCONV) conv_step(S, Ic<i>{});
    });
    static_for<RT>([&](auto rt_tag) {
      constexpr int rt = decltype(rt_tag)::value;
      constexpr int eh = 24 * rt + 19, el = 24 * rt + 23;
      constexpr int sh = eh < 12 ? eh / 3 : 4 + (eh - 12) / 2, sl = el < 12 ? el / 3 : 4 + (el - 12) / 2;
      if constexpr (slot == sh + 1) pin_h1(kb_tag, rt_tag, Ic<0>{});
      if constexpr (slot == sl + 1) pin_h1(kb_tag, rt_tag, Ic<1>{});
    });
  };

  // ---- in-kernel noise (RNG): Philox4x32 (kPhiloxRounds) + Box-Muller of rng.hpp (normal4: same operations, same bits), and
  // Q = a x + sb z written over x in the wave's LDS image, as single-instruction-sized steps that ride beside the MFMAs of
  // layer 2 where no conversion runs (the second halves of its chunks and layer 3's k-blocks: noise_l3 / noise_b below): 64
  // steps per quad, 1.5 quads = 96 steps per chunk; chunk pair p does the three quads j = 0, 1, 2 of row tile p.
  unsigned qx = 0, qy = 0, qz = 0, qw = 0, qtx = 0;
  unsigned long long qp0 = 0, qp1 = 0;
  float qu0 = 0.0f, qu1 = 0.0f, qu2 = 0.0f, qu3 = 0.0f, qr0 = 0.0f, qr1 = 0.0f;
  f32x4 qxv = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, qzv = qxv;
  float n_ca = 0.0f, n_sb = 0.0f;     // this step's a and sb
  int n_step = 0;
  auto noise_sub = [&](int rt, auto j_tag, auto sub_tag) {
#pragma clang fp contract(off)
    constexpr int j = decltype(j_tag)::value, sub = decltype(sub_tag)::value;
    constexpr int R4 = 4 * kPhiloxRounds;   // four single-instruction steps per Philox round (rng.hpp), then 11 of Box-Muller and Q
    static_assert(R4 + 11 < 64, "a quad's steps fit its 64 slots");
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    const float k32 = 2.3283064365386963e-10f, c2 = -1.3862943611198906f;
    if constexpr (sub == 0) {
      const unsigned ln = here((unsigned)lane);
      unsigned r = (unsigned)row0 + 16u * (unsigned)rt + (ln & 15u);   // (N < 2^31: checked by the host)
      if (r > (unsigned)last_row) r = (unsigned)last_row;
      const unsigned long long e = ((unsigned long long)a.row_offset + r) * 10u + (unsigned long long)(4 * j + (int)(ln >> 4));
      qx = (unsigned)e, qy = (unsigned)(e >> 32), qz = (unsigned)n_step, qw = 0x5053544Cu;
      qxv = xq[(rt * 3 + j) * 64];
    } else if constexpr (sub <= R4) {
      constexpr int r = (sub - 1) >> 2, ph = (sub - 1) & 3;
      const unsigned k0 = (unsigned)seed + (unsigned)r * W0, k1 = (unsigned)(seed >> 32) + (unsigned)r * W1;
      if constexpr (ph == 0) qp0 = (unsigned long long)M0 * qx;
      else if constexpr (ph == 1) qp1 = (unsigned long long)M1 * qz;
      else if constexpr (ph == 2) qtx = (unsigned)(qp1 >> 32) ^ qy ^ k0;
      else {
        qz = (unsigned)(qp0 >> 32) ^ qw ^ k1;
        qx = qtx, qy = (unsigned)qp1, qw = (unsigned)qp0;
      }
    } else if constexpr (sub == R4 + 1) {
      qu0 = ((float)qx + 1.0f) * k32;
    } else if constexpr (sub == R4 + 2) {
      qu1 = (float)qy * k32;
    } else if constexpr (sub == R4 + 3) {
      qu2 = ((float)qz + 1.0f) * k32;
    } else if constexpr (sub == R4 + 4) {
      qu3 = (float)qw * k32;
    } else if constexpr (sub == R4 + 5) {
      qr0 = __builtin_amdgcn_sqrtf(c2 * __builtin_amdgcn_logf(qu0));
    } else if constexpr (sub == R4 + 6) {
      qr1 = __builtin_amdgcn_sqrtf(c2 * __builtin_amdgcn_logf(qu2));
    } else if constexpr (sub == R4 + 7) {
      qzv[0] = qr0 * __builtin_amdgcn_cosf(qu1);
      qzv[1] = qr0 * __builtin_amdgcn_sinf(qu1);
    } else if constexpr (sub == R4 + 8) {
      qzv[2] = qr1 * __builtin_amdgcn_cosf(qu3);
      qzv[3] = qr1 * __builtin_amdgcn_sinf(qu3);
    } else if constexpr (sub == R4 + 9) {
      const float lsb = (j < 2 || own2) ? n_sb : 0.0f;
      qzv *= lsb;
    } else if constexpr (sub == R4 + 10) {
      const float la = (j < 2 || own2) ? n_ca : 1.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) qzv[e] = __builtin_fmaf(la, qxv[e], qzv[e]);
    } else if constexpr (sub == R4 + 11) {
      xq[(rt * 3 + j) * 64] = qzv;
    }
  };
  // step s (0..95) of the sequence of chunk 2 p (EVEN) / 2 p + 1 (odd)
  auto noise_slot = [&](int p, auto odd_tag, auto s_tag) {
    constexpr int s = decltype(s_tag)::value;
    constexpr bool ODD = decltype(odd_tag)::value;
    if constexpr (RNG && !(PSTL_C2_ABL & 1)) {
      if constexpr (!ODD) {
        if constexpr (s < 64) noise_sub(p, Ic<0>{}, Ic<s>{});
        else noise_sub(p, Ic<1>{}, Ic<s - 64>{});
      } else {
        if constexpr (s < 32) noise_sub(p, Ic<1>{}, Ic<s + 32>{});
        else noise_sub(p, Ic<2>{}, Ic<s - 32>{});
      }
    }
  };

  // Where a chunk's 96 steps ride (an MFMA leaves ~8 cycles of issue beside it: profiles/r5/valu_beside_mfma.txt; 96 steps are
  // ~1 000 cycles, a phase's 96 slots have 770): the first 24 in two of every three slots of layer 3's k-block behind the chunk's
  // first-half phase -- slots that carry nothing else --, the other 72 in three of every four slots of its second-half phase.
  // Chunk 0 has no layer-3 block in front of it: all 96 in its second half.
  // (RT = 3: a phase has 72 slots, the layer-3 block 27 -- the first 24 steps in the block's first 24 slots, the other 72 one per
  // slot of the phase; chunk 0 doubles up in its first 24 slots; chunk pair 3 has no row tile to draw for.)
  auto noise_l3 = [&](int p, auto odd_tag, auto k_tag) {
    constexpr int k = decltype(k_tag)::value;
    if constexpr (RT == 4) {
      if constexpr (k % 3 != 2) noise_slot(p, odd_tag, Ic<k - k / 3>{});
    } else if constexpr (RT == 3) {
      if constexpr (k < 24) noise_slot(p, odd_tag, Ic<k>{});
    }
  };
  auto noise_b = [&](int p, auto odd_tag, auto k_tag) {
    constexpr int k = decltype(k_tag)::value;
    if constexpr (RT == 4) {
      if constexpr (k % 4 != 3) noise_slot(p, odd_tag, Ic<24 + k - k / 4>{});
    } else if constexpr (RT == 3) {
      noise_slot(p, odd_tag, Ic<24 + k>{});
    }
  };
  auto noise_b0 = [&](auto k_tag) {   // chunk 0: all 96 steps in its second-half phase
    constexpr int k = decltype(k_tag)::value;
    if constexpr (RT == 4) {
      noise_slot(0, std::false_type{}, k_tag);
    } else if constexpr (RT == 3) {
      if constexpr (k < 24) {
        noise_slot(0, std::false_type{}, Ic<2 * k>{});
        noise_slot(0, std::false_type{}, Ic<2 * k + 1>{});
      } else {
        noise_slot(0, std::false_type{}, Ic<24 + k>{});
      }
    }
  };

  // one k-block of layer 3: the pieces of a finished chunk against its W3 blocks (pieces 16..21 of `slot`; the first pair
  // w3h[0] / w3l[0] was read by the caller); fn(k): what else rides in slot k
  auto layer3 = [&](unsigned slot, auto&& fn) {
    static_for<9 * RT>([&](auto m_tag) {
      constexpr int m = decltype(m_tag)::value;
      constexpr int j = m / (3 * RT), pr = (m / RT) % 3, rt = m % RT;
      const f16x8 wa = pr == 1 ? w3l[j & 1] : w3h[j & 1];
      const f16x8 pb = pr == 2 ? pieces_l(rt) : pieces_h(rt);
      acc3[j][rt] = mfma(wa, pb, acc3[j][rt]);   // (FIRST: the accumulators hold b3, read into them behind B(0))
      if constexpr (j < 2 && pr == 0 && rt == 0) w3h[(j + 1) & 1] = rdA(slot, 16 + (j + 1) * 2);
      if constexpr (j < 2 && pr == 0 && rt == 1) w3l[(j + 1) & 1] = rdA(slot, 16 + (j + 1) * 2 + 1);
      fn(m_tag);
      FENCE();
    });
  };

  // what every phase has in common: slot (kq, m) after its MFMA -- A operands of the next k-block, the DMA of the phase
  // after next in the third k-block, the barrier after the second
  auto common_fill = [&](auto kq_tag, auto m_tag, auto kind_tag, auto np_tag, int c_issue, int c3_issue) {
    constexpr int kq = decltype(kq_tag)::value, m = decltype(m_tag)::value, NP = decltype(np_tag)::value;
    if constexpr (m < 4) {   // (kq == 3: the next phase's first k-block; its slot was published by this phase's barrier)
      const unsigned sl = kq < 3 ? s_cur : s_nxt;
      constexpr int pc = pidx(kq < 3 ? kq + 1 : 0, m);
      if constexpr (m & 1) nl[m >> 1] = rdA(sl, pc);
      else nh[m >> 1] = rdA(sl, pc);
    }
    // this wave's DMA share of the phase after next (NP = 4: its four W1 / W2 pieces; 6: + the W3 pair), in the third k-block
    if constexpr (kq == 2 && m == 2) issue_w(kind_tag, c_issue, s_nn);
    if constexpr (kq == 2 && m == NM / 2 && NP == 6) issue_w3(c3_issue, s_nn);
  };
  auto end_kq = [&](auto kq_tag, auto&& mid) {
    constexpr int kq = decltype(kq_tag)::value;
    if constexpr (kq == 1) {
      // every wave's share of phase p + 1 has landed (issued a whole phase ago), every wave is done with phase p - 1
      if (PSTL_C2_ABL & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (PSTL_C2_ABL & 1024) asm volatile("s_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      FENCE();
      mid();   // (nothing of this wave is in flight here: a compiler-counted global load consumed now waits for nothing)
      FENCE();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) ah[t] = nh[t], al[t] = nl[t];
  };
  auto rotate = [&] {
    const unsigned t_ = s_cur;
    s_cur = s_nxt, s_nxt = s_nn, s_nn = t_;
  };

  // ---- a layer-1 phase: chunks 2 Q (accumulators D0) and 2 Q + 1 (D1); the chunk before each is converted in its shadow
  // (S0: the chunk before 2 Q -- none when Q == 0; D0 under the second chunk).  Issues phase p + 2 (kind, c_issue).
  auto l1_phase = [&](auto q_tag, auto& D0, auto& D1, auto kind_tag, auto np_tag, int c_issue, int c3_issue) {
    constexpr int Q = decltype(q_tag)::value;
    FENCE();
    static_for<4>([&](auto kq_tag) {
      constexpr int kq = decltype(kq_tag)::value;
      constexpr int cl = kq >> 1, kb = kq & 1, c1 = 2 * Q + cl;
      static_for<NM>([&](auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
        constexpr int pr = m / (2 * RT), t = (m / RT) % 2, rt = m % RT;
        const f16x8 wa = pr == 1 ? al[t] : ah[t];
        const f16x8 xb = pr == 2 ? xl[kb][rt] : xh[kb][rt];
        auto& D = cl == 0 ? D0 : D1;
        auto& Dn = cl == 0 ? D1 : D0;   // the next chunk's accumulators = the chunk before's, converted in this chunk's shadow
        D[t][rt] = mfma(wa, xb, D[t][rt]);   // (the first 2 RT start from the scene / timestep part, read into D: below)
        common_fill(kq_tag, m_tag, kind_tag, np_tag, c_issue, c3_issue);
        // the constant parts a chunk's first 2 RT MFMAs (m = t RT + rt) start from, kLead slots ahead, into the accumulators
        // themselves: the last kLead slots of the chunk before (k-block 1) fetch those of MFMAs 0 .. kLead-1 (the chunk before
        // THAT one's accumulators (t = 0) are converted by slot 42), slot m of k-block 0 fetches that of MFMA m + kLead
        if constexpr (kb == 1 && m >= NM - kLead && c1 < 7)
          Dn[(m - (NM - kLead)) / RT][(m - (NM - kLead)) % RT] = rd_cst(c1 + 1, (m - (NM - kLead)) / RT, (m - (NM - kLead)) % RT);
        if constexpr (kb == 0 && m + kLead < 2 * RT) D[(m + kLead) / RT][(m + kLead) % RT] = rd_cst(c1, (m + kLead) / RT, (m + kLead) % RT);
        // conversion of the chunk before this one, over this chunk's two k-blocks
        if constexpr (c1 > 0 && !(PSTL_C2_ABL & 2)) {
          auto& S = cl == 0 ? D1 : D0;
          conv_h1_slot(S, Ic<c1 - 1>{}, Ic<kb * NM + m>{});
        }
        FENCE();
      });
      end_kq(kq_tag, [] {});
    });
    rotate();
  };

  // ---- a layer-2 phase on D (chunk c, half HALF).  CONV 1: the layer-2 chunk before (S) is converted in its shadow and its
  // layer 3 follows the fourth k-block; CONV 2: S is layer 1's last chunk (pieces -> k-block 7 of h1).
  auto l2_phase = [&](auto& D, auto& S, auto half_tag, auto conv_tag, auto&& l3_fn, auto kind_tag, auto np_tag, int c_issue,
                      int c3_issue, int c_bias, auto&& mid, auto&& slot_fn, int c_conv = 0) {
    constexpr int HALF = decltype(half_tag)::value;
    constexpr int CONV = decltype(conv_tag)::value;
    FENCE();
    static_for<4>([&](auto kq_tag) {
      constexpr int kq = decltype(kq_tag)::value;
      constexpr int kb = 4 * HALF + kq;
      static_for<NM>([&](auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
        constexpr int pr = m / (2 * RT), t = (m / RT) % 2, rt = m % RT;
        const f16x8 wa = pr == 1 ? al[t] : ah[t];
        const f16x8 xb = pr == 2 ? bl[kb][rt] : bh[kb][rt];
        D[t][rt] = mfma(wa, xb, D[t][rt]);   // (a chunk starts from its bias, read into D by the phase before)
        common_fill(kq_tag, m_tag, kind_tag, np_tag, c_issue, c3_issue);
        if constexpr (CONV == 1 && !(PSTL_C2_ABL & 4)) {
          constexpr int i0 = 2 * (kq * NM + m);
          if constexpr (SAVE && i0 < NCONV && i0 % 12 == 0) save_quad(a.h2_save, S, c_conv, Ic<i0 / 12>{});
          if constexpr (i0 < NCONV) conv_step(S, Ic<i0>{});
          if constexpr (i0 + 1 < NCONV) conv_step(S, Ic<i0 + 1>{});
        }
        if constexpr (CONV == 2 && !(PSTL_C2_ABL & 2)) {
          if constexpr (kq * NM + m < 2 * NM) conv_h1_slot(S, Ic<7>{}, Ic<kq * NM + m>{});
        }
        // bias of the next chunk, into its accumulators (= the chunk before's: converted in the first half's shadow)
        if constexpr (HALF == 1 && kq == 3 && m < 2 * RT) S[m / RT][m % RT] = rd_bias(c_bias, m / RT);
        if constexpr (CONV == 1 && kq == 3 && m == 6) w3h[0] = rdA(s_cur, 16);
        if constexpr (CONV == 1 && kq == 3 && m == 7) w3l[0] = rdA(s_cur, 17);
        slot_fn(Ic<kq * NM + m>{});
        FENCE();
      });
      end_kq(kq_tag, mid);
    });
    if constexpr (CONV == 1 && !(PSTL_C2_ABL & 512)) layer3(s_cur, l3_fn);
    rotate();
  };
  using Yes = std::true_type;
  using No = std::false_type;

  const f32x4 sc = f32x4{a.w_max, a.a_max, a.w_max, a.a_max};
#ifdef PSTL_C2_STAMP
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev, st_rt0, st_t0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev), "=s"(st_rt0)::"memory");
  st_t0 = st_prev;
#endif
  float* rcw = reinterpret_cast<float*>(smem + kOffRc) + (w * RT * 4) * 64 + lane;   // [rt][e] at (rt*4 + e)*64
  if constexpr (MU) {   // (a workgroup's last tile makes pieces of what lies here: they are not used, but the domain guard sees them)
#pragma unroll
    for (int u = 0; u < RT * 4; ++u) rcw[u * 64] = 0.0f;
  }
#pragma unroll 1
  for (int n = 0;; ++n) {   // (left by the break behind the last epilogue: n_iter >= 1)
    const int i = MU ? a.step_hi : a.step_hi - n;
    const f32x4 cf4 = coef[MU ? 0 : n];
    const float kk = cf4[0], ca = cf4[1], sb = cf4[2];
    n_ca = ca, n_sb = sb, n_step = i;
    const bool more = n + 1 < n_iter;
    // the timestep row of the NEXT step, on its way while this one computes
    float tb_next = 0.0f;
    if (!MU && more) tb_next = a.tbias[(long)(i - 1) * kHid2 + tid];
    auto none = [] {};
    auto noslot = [](auto) {};
    // the constant rows of the next step, written right behind the barrier of B(0): every wave is past layer 1's reads of
    // this step's rows (two barriers ago), the next reads come after the barriers of the remaining phases
    auto write_crow = [&] {
      if (!MU && more) {
#pragma unroll
        for (int s = 0; s < kMaxScn; ++s) crow[s * 256 + tid] = (brow[s * 256 + tid] + tb_next) * kAcc;
      }
    };
    // MU: the next tile of this workgroup, its scenes' rows requested behind the barrier of A(6) and written behind B(6)'s
    const long nwg_row0 = wg_row0 + (long)gridDim.x * kWgRows;
    const long nscene_first = (MU && more) ? nwg_row0 / a.rows_per_scene : 0;
    // (by LDS-DMA into the table of scene rows, which this mode does not use otherwise: each wave moves its own 64 columns and
    // reads them back itself, behind the wait in front of the next barrier)
    auto load_nbase = [&] {
      if constexpr (MU) {
        if (more) {
          const long lr = nwg_row0 + kWgRows - 1 > last_row ? last_row : nwg_row0 + kWgRows - 1;
          const int nscn = (int)(lr / a.rows_per_scene - nscene_first) + 1;
          const unsigned voff = here((unsigned)tid) * 4u;
#pragma unroll
          for (int s = 0; s < kMaxScn; ++s)
            dma4(a.base + (nscene_first + (s < nscn ? s : nscn - 1)) * kHid2, voff, (unsigned)kOffBrow + (unsigned)(s * 1024 + w * 256));
        }
      }
    };
    auto write_ncrow = [&] {
      if constexpr (MU) {
        if (more) {
          const float tbc = tbs[tid];
#pragma unroll
          for (int s = 0; s < kMaxScn; ++s) crow[s * 256 + tid] = (brow[s * 256 + tid] + tbc) * kAcc;
        }
      }
    };

    // ---- layer 1: 47 -> 256 (four phases), its pieces into the accumulation registers ----
    if (PSTL_C2_ABL & 256) {
      rotate(), rotate(), rotate(), rotate();
    } else {
    l1_phase(Ic<0>{}, accA, accB, Ic<0>{}, Ic<4>{}, 2, 0);          // issues P2
    l1_phase(Ic<1>{}, accA, accB, Ic<0>{}, Ic<4>{}, 3, 0);          // issues P3
    l1_phase(Ic<2>{}, accA, accB, Ic<1>{}, Ic<4>{}, 0, 0);          // issues A(0)
    l1_phase(Ic<3>{}, accA, accB, Ic<2>{}, Ic<4>{}, 0, 0);          // issues B(0)
    }
    C2_STAMP(0)
    // ---- layers 2 + 3 ----
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) accA[t][rt] = rd_bias(0, t);   // (layer 1's chunk 6 left these registers a phase ago)
    l2_phase(accA, accB, Ic<0>{}, Ic<2>{}, noslot, Ic<1>{}, Ic<6>{}, 1, 0, 0, none, noslot);    // A(0): converts layer 1's chunk 7; issues A(1) + W3[0]
    l2_phase(accA, accB, Ic<1>{}, Ic<0>{}, noslot, Ic<2>{}, Ic<4>{}, 1, 0, 1, write_crow, [&](auto s_) { noise_b0(s_); });    // B(0); issues B(1)
    C2_STAMP(1)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc3[j][rt] = *reinterpret_cast<const f32x4*>(b3s + 16 * j + 4 * g);
    l2_phase(accB, accA, Ic<0>{}, Ic<1>{}, [&](auto k_) { noise_l3(0, Yes{}, k_); }, Ic<1>{}, Ic<6>{}, 2, 1, 0, none, noslot, 0);   // A(1): layer 3 of chunk 0
    l2_phase(accB, accA, Ic<1>{}, Ic<0>{}, noslot, Ic<2>{}, Ic<4>{}, 2, 0, 2, none, [&](auto s_) { noise_b(0, Yes{}, s_); });
    C2_STAMP(2)
#pragma unroll 1
    for (int cc = 2; cc < 6; cc += 2) {
      l2_phase(accA, accB, Ic<0>{}, Ic<1>{}, [&](auto k_) { noise_l3(cc >> 1, No{}, k_); }, Ic<1>{}, Ic<6>{}, cc + 1, cc, 0, none, noslot, cc - 1);
      C2_STAMP(3)
      l2_phase(accA, accB, Ic<1>{}, Ic<0>{}, noslot, Ic<2>{}, Ic<4>{}, cc + 1, 0, cc + 1, none, [&](auto s_) { noise_b(cc >> 1, No{}, s_); });
      C2_STAMP(4)
      l2_phase(accB, accA, Ic<0>{}, Ic<1>{}, [&](auto k_) { noise_l3(cc >> 1, Yes{}, k_); }, Ic<1>{}, Ic<6>{}, cc + 2, cc + 1, 0, none, noslot, cc);
      C2_STAMP(3)
      l2_phase(accB, accA, Ic<1>{}, Ic<0>{}, noslot, Ic<2>{}, Ic<4>{}, cc + 2, 0, cc + 2, none, [&](auto s_) { noise_b(cc >> 1, Yes{}, s_); });
      C2_STAMP(4)
    }
    l2_phase(accA, accB, Ic<0>{}, Ic<1>{}, [&](auto k_) { if constexpr (RT > 3) noise_l3(3, No{}, k_); }, Ic<1>{}, Ic<6>{}, 7, 6, 0, load_nbase, noslot, 5);    // A(6); issues A(7) + W3[6]
    l2_phase(accA, accB, Ic<1>{}, Ic<0>{}, noslot, Ic<2>{}, Ic<4>{}, 7, 0, 7, write_ncrow, [&](auto s_) { if constexpr (RT > 3) noise_b(3, No{}, s_); });
    l2_phase(accB, accA, Ic<0>{}, Ic<1>{}, [&](auto k_) { if constexpr (RT > 3) noise_l3(3, Yes{}, k_); }, Ic<0>{}, Ic<6>{}, 0, 7, 0, none, noslot, 6);    // A(7); issues P0 of the next step + W3[7]
    l2_phase(accB, accA, Ic<1>{}, Ic<0>{}, noslot, Ic<0>{}, Ic<4>{}, 1, 0, 0, none, [&](auto s_) { if constexpr (RT > 3) noise_b(3, Yes{}, s_); });    // B(7); issues P1 of the next step
    C2_STAMP(5)
    // tail: layer 3 of chunk 7.  Its W3 blocks sit in the slot of the NEXT tile-step's first phase (s_cur now): landed and
    // published by the barrier in the middle of the phase just finished.
    w3h[0] = rdA(s_cur, 16);
    w3l[0] = rdA(s_cur, 17);
    f32x4 qv[RT][3];   // RNG: Q = a x + sb z, made in the shadow of layer 2; otherwise x -- on their way while the tail computes
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 3; ++j) qv[rt][j] = xq[(rt * 3 + j) * 64];
    float vio[RT];   // REF: 1 where the row's score says "violated" (only those rows are refined), on its way like qv
    if constexpr (REF) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        long r = row0 + 16 * rt + col;
        if (r > last_row) r = last_row;
        vio[rt] = a.scores[r];
      }
    }
    if constexpr (SAVE) static_for<2 * RT>([&](auto g_tag) { save_quad(a.h2_save, accB, 7, g_tag); });
    if (!(PSTL_C2_ABL & 64)) static_for<NCONV>([&](auto i_tag) { conv_step(accB, i_tag); });
    if constexpr (MU) {
      // the next tile's state, straight into this wave's image (its reads above have returned): per row tile three 16-byte
      // pieces (lane (g, col): columns 16 j + 4 g .. of row col; tile j = 2 holds x only in lanes g < 2) and four 4-byte
      // pieces with the row constants of lanes g >= 2 (g = 2: hl, stlp 0..2; g = 3: stlp 3..5 and a word that is dropped)
      if (more) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned ln = here((unsigned)lane);
        const unsigned lc = ln & 15u, lg = ln >> 4;
        const long nrow0 = nwg_row0 + (long)w * (16 * RT);
        static_for<RT>([&](auto rt_tag) {
          constexpr int rt = decltype(rt_tag)::value;
          long tb0 = nrow0 + 16 * rt;                                  // (uniform)
          if (tb0 > last_row) tb0 = last_row;
          const unsigned rem = (unsigned)(last_row - tb0);
          const unsigned lr = lc < rem ? lc : rem;                     // rows behind the end repeat the last one
          const float* xb = (REF ? a.init : a.x_inout) + tb0 * kCtrl2;
          const unsigned voff = lr * (unsigned)(kCtrl2 * 4) + lg * 16u;
          const unsigned dst = (unsigned)kOffXq + (unsigned)((w * RT * 3 + rt * 3) * 64) * 16u;
          dma16(xb, voff, dst);
          dma16(xb + 16, voff, dst + 1024u);
          dma16(xb + 32, lr * (unsigned)(kCtrl2 * 4) + (lg & 1u) * 16u, dst + 2048u);
          const float* spb = a.stlp + tb0 * 6;
          const float* p0 = lg == 2u ? (a.hl + tb0) + lr : spb + (lr * 6u + 3u);
          const unsigned o1 = lr * 24u + (lg == 2u ? 0u : 16u);
          const unsigned rdst = (unsigned)kOffRc + (unsigned)((w * RT * 4 + rt * 4) * 64) * 4u;
          dma4(p0, rdst);
          dma4(spb, o1, rdst + 256u);
          dma4(spb, o1 + 4u, rdst + 512u);
          dma4(spb, o1 + (lg == 2u ? 8u : 4u), rdst + 768u);
          FENCE();
        });
      }
    }
    FENCE();
    layer3(s_cur, noslot);

    C2_STAMP(6)
    // ---- epilogue: eps = layer 3 + b3 (already in the accumulators); x' = a x + sb z - kk eps; candidates; next pieces ----
    const bool emit = !MU && i <= a.n_emit, last = MU || i == a.step_lo;
    {
    // (global addresses = uniform pointer of the tile's first row + a 32-bit lane offset, re-derived here: see here())
    const unsigned ln = here((unsigned)lane);
    const unsigned lc = ln & 15u, lg = ln >> 4;
    const float nk = -kk * kInvAcc;            // the accumulators carry the factor kAcc
    const float nk2 = own2 ? nk : 0.0f;        // (tile j = 2, lanes g >= 2: the row constants stay what they are)
    f32x4 xn[RT][3];
#pragma unroll
    for (int rt = 0; rt < ((PSTL_C2_ABL & 8) ? 0 : RT); ++rt) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        f32x4 q = qv[rt][j];
        if constexpr (REF) {
          // interval head (nusc_model.py:212-229): the tanh of the output scales into the headroom init leaves
          const float viol = vio[rt] < 0.0f ? 1.0f : 0.0f;
          f32x4 o, v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = acc3[j][rt][e] * kInvAcc;
            const float raw = tanhf(o[e]);
            const float d = raw >= 0.0f ? raw * (sc[e] - q[e]) : raw * (q[e] - (-sc[e]));
            v[e] = q[e] + d * viol;
            if (a.clip) v[e] = clip_keep_nan(v[e], sc[e]);
          }
          if ((j < 2 || own2) && (unsigned)row0 + 16u * (unsigned)rt + lc <= (unsigned)last_row) {
            if (!(fabsf((o[0] + o[1]) + (o[2] + o[3])) <= 3.0e38f)) atomicOr(a.status, 1u);
            if constexpr (SAVE)   // layer 3's output before the tanh (the backward pass's `pre`)
              *reinterpret_cast<f32x4*>((a.pre_save + (long)((unsigned)row0 + 16u * (unsigned)rt) * kCtrl2) +
                                        (lc * (unsigned)kCtrl2 + (unsigned)(16 * j) + 4u * lg)) = o;
          }
          xn[rt][j] = v;
          continue;
        }
        if constexpr (!RNG) {
          const bool upd = j < 2 || own2;
          const unsigned trow0 = (unsigned)row0 + 16u * (unsigned)rt;
          f32x4 z = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          if (!MU && sb != 0.0f && upd && trow0 + lc <= (unsigned)last_row)   // (MU: mu only, no noise term)
            z = *reinterpret_cast<const f32x4*>((a.noise + ((long)(a.steps - 1 - i) * a.N + trow0) * kCtrl2) +
                                                (lc * (unsigned)kCtrl2 + (unsigned)(16 * j) + 4u * lg));
          const float la = upd ? ca : 1.0f, lsb = upd ? sb : 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = __builtin_fmaf(la, q[e], lsb * z[e]);
        }
        const float nkl = j < 2 ? nk : nk2;
#pragma unroll
        for (int e = 0; e < 4; ++e) xn[rt][j][e] = __builtin_fmaf(nkl, acc3[j][rt][e], q[e]);
        if constexpr (!MU) xq[(rt * 3 + j) * 64] = xn[rt][j];
      }
      if constexpr (!MU) make_x_pieces(rt, xn[rt][0], xn[rt][1], xn[rt][2]);
    }
    auto store_tile = [&](int rt) {
      {
        const unsigned trow0 = (unsigned)row0 + 16u * (unsigned)rt;     // uniform
        const bool in = trow0 + lc <= (unsigned)last_row;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const unsigned loff = lc * (unsigned)kCtrl2 + (unsigned)(16 * j) + 4u * lg;
          if ((j < 2 || own2) && in) {
            const f32x4 v0 = xn[rt][j];
            if (last) {
              *reinterpret_cast<f32x4*>(((REF ? a.out : a.x_inout) + (long)trow0 * kCtrl2) + loff) = v0;
              if (!REF && !(fabsf((v0[0] + v0[1]) + (v0[2] + v0[3])) <= 3.0e38f)) atomicOr(a.status, 1u);
            }
            if (emit) {
              f32x4 v = v0 * sc;
              if (a.clip) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = clip_keep_nan(v[e], sc[e]);
              }
              *reinterpret_cast<f32x4*>((a.emit_out + ((long)(a.n_emit - i) * a.N + trow0) * kCtrl2) + loff) = v;
            }
          }
        }
      }
    };
    if constexpr (MU) {
      // (the last tile leaves the loop HERE, so that the pieces are redefined on every path to the back edge: under an
      // `if (more)` the old ones stay live through the whole tile-step as far as the register allocator can tell -- 64 registers)
      {
        // the next tile's pieces from its image, row tile by row tile, each followed by the stores of this tile's result (after
        // the wait: stores in flight would be waited for too; interleaved: the results leave the registers as the pieces fill them)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          // (the last tile of the workgroup makes pieces of whatever its image holds -- nobody uses them: an `if (more)` around
          // this would leave the old pieces live through the whole tile-step as far as the register allocator can tell)
          f32x4 q0 = xq[(rt * 3 + 0) * 64], q1 = xq[(rt * 3 + 1) * 64];
          f32x4 q2 = xq[(rt * 3 + 2) * 64];
          const f32x4 rc = f32x4{rcw[(rt * 4 + 0) * 64], rcw[(rt * 4 + 1) * 64], rcw[(rt * 4 + 2) * 64], rcw[(rt * 4 + 3) * 64]};
          if (!own2) q2 = f32x4{rc[0], rc[1], rc[2], g == 2 ? rc[3] : 0.0f};
          FENCE();
          store_tile(rt);
          FENCE();
          if constexpr (REF) {
            long r = nwg_row0 + (long)w * (16 * RT) + 16 * rt + col;
            add_pooled(r > last_row ? last_row : r, q0, q1, q2);
          }
          make_x_pieces(rt, q0, q1, q2);
          FENCE();
        }
      }
    } else {
      if (last || emit) {   // (uniform: the last step of the launch, and the steps whose state is a candidate)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) store_tile(rt);
      }
    }
    }
    if (!more) break;
    if constexpr (MU) {
      wg_row0 = nwg_row0, row0 = nwg_row0 + (long)w * (16 * RT), scene_first = nscene_first;
      set_scn();
    }
    C2_STAMP(7)
    // the first chunk's constant part for the next step (its rows were written above, two or more barriers ago)
#pragma unroll
    for (int u = 0; u < kLead; ++u) accA[u / RT][u % RT] = rd_cst(0, u / RT, u % RT);
  }
#ifdef PSTL_C2_STAMP
  {
    unsigned long long t1_, r1_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_), "=s"(r1_)::"memory");
    if (blockIdx.x == 7 && lane == 0) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.emit_out) + w * 12;
      for (int k = 0; k < 8; ++k) dbg[k] = st_sum[k];
      dbg[8] = t1_ - st_t0, dbg[9] = r1_ - st_rt0, dbg[10] = (unsigned long long)n_iter;
    }
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last step's look-ahead DMA
  if (pieces_overflowed(ovf) || !(ovfx < 65520.0f)) atomicOr(a.status, 1u);   // a layer input left |x| < 4094 somewhere in this launch
}

}  // namespace

// The unit is compiled in three parts (build.py: -DPSTL_C2_PART=0 | 1 | 2; twelve instantiations of the kernel take minutes in one
// piece): 0 = the multi-step forms and every host-side rule, 1 = the tile-walking forms of the denoiser's single steps and of
// RefineNet's inference pass, 2 = RefineNet's training forward pass.  Undefined (tools/dbg builds): everything.
#ifndef PSTL_C2_PART
#define PSTL_C2_ALL 1
#define PSTL_C2_PART -1
#else
#define PSTL_C2_ALL 0
#endif
int launch_chain2_walk(const ChainArgs& a, hipStream_t st, int rows);           // part 1
int launch_chain2_refine_infer(const ChainArgs& a, hipStream_t st, int rows);   // part 1
int launch_chain2_refine_train(const ChainArgs& a, hipStream_t st, int rows);   // part 2

#if PSTL_C2_ALL || PSTL_C2_PART == 0
bool chain2_eligible(const ChainArgs& a) {
  // multi-step segments, and the single-step launches of the guided phase (mu_only = 1) in the form whose workgroups walk
  // several tiles (one workgroup per 256 rows pays the prologue -- state, constant rows, the first two phases of the weight
  // stream -- for ONE tile-step: measured 0.61 ms against k_chain's streamed single-step layout at 0.55 ms, 786 432 rows)
  if (a.mu_only ? (a.mu_only != 1 || a.step_hi != a.step_lo) : a.step_hi <= a.step_lo) return false;
  if (a.h1_save || a.h2_save || a.pre_save || a.init) return false;  // policy_net inference only
  if (a.rows_per_scene % 16 != 0 || a.rows_per_scene < 48) return false;
  if (a.step_hi - a.step_lo + 1 > kMaxLaunchSteps) return false;
  return true;
}

#endif

// a 192-row tile-step in per cent of a 256-row one: 21.1 against 25.1 us at 786 432 rows (16 against 12 rounds: 13.17 against
// 12.55 ms per 39-step launch), 20.8 against 26.8 at 196 608 rows (profiles/r5/chain2_192_row_workgroups.txt)
constexpr int kCost192 = 82;

template <bool RNG, int MODE, int RT>
static int launch_chain2_t(const ChainArgs& a, hipStream_t st) {
  constexpr bool MU = MODE != 0;
  long n_wg = (a.N + Carve<RT>::kWgRows - 1) / Carve<RT>::kWgRows;
  static DeviceOnce allowed;     // (per instantiation and DEVICE: pstl_common.hpp)
  const int dev = current_device();
  if (dev < 0) return PSTL_ERR_LAUNCH;
  if (!allowed.done(dev)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain2<RNG, MODE, RT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            Carve<RT>::kLdsBytes) != hipSuccess)
      return PSTL_ERR_LAUNCH;
    allowed.set(dev);
  }
  const int cus = device_cus(dev);
  if (MU && n_wg > cus) n_wg = cus;   // one workgroup per CU walks the tiles
  hipLaunchKernelGGL((k_chain2<RNG, MODE, RT>), dim3((unsigned)n_wg), dim3(256), Carve<RT>::kLdsBytes, st, a);
  return launch_status();
}

#if PSTL_C2_ALL || PSTL_C2_PART == 0
// Rows per workgroup of a launch.  Multi-step: 256.  Single-step (the workgroups walk tiles): 128 where that takes less time
// per CU -- a 128-row tile-step is 0.6 of a 256-row one (16-17 against 26-29 us measured: each A operand read feeds half the
// MFMAs), so it pays exactly where 256-row tiles would leave CUs idle: 24 576 rows are 96 tiles of 256 on 256 CUs or 192 of
// 128 (29.7 -> 19.1 us per launch), 98 304 rows two of 256 per CU or three of 128 (61.6 -> 53.4 us).  Same bits either way.
int chain2_wg_rows(const ChainArgs& a) {
  const int cus = device_cus();
  const long t4 = (a.N + 255) / 256, r4 = (t4 + cus - 1) / cus;
#ifdef PSTL_C2_FORCE_ROWS
  if (!a.mu_only) return PSTL_C2_FORCE_ROWS;
#endif
  if (!a.mu_only) {   // multi-step: rounds of 192-row workgroups (three row tiles per wave) where they take less time
    const long t3 = (a.N + 191) / 192, r3 = (t3 + cus - 1) / cus;
    return r3 * kCost192 < r4 * 100 ? 192 : 256;
  }
  const long t3 = (a.N + 191) / 192, r3 = (t3 + cus - 1) / cus, t2 = (a.N + 127) / 128, r2 = (t2 + cus - 1) / cus;
  const long c4 = r4 * 100, c3 = r3 * kCost192, c2 = r2 * 60;
  return c2 < c4 && c2 <= c3 ? 128 : c3 < c4 ? 192 : 256;
}

long chain2_step_cost(const ChainArgs& a) {
  const int cus = device_cus();
  const int rows = chain2_wg_rows(a);
  const long t = (a.N + rows - 1) / rows, r = (t + cus - 1) / cus;
  return r * (rows == 256 ? 100 : rows == 192 ? kCost192 : 60);
}

int launch_chain2(const ChainArgs& a, hipStream_t st) {
  // in-kernel noise (PSTL_FLAG_RNG) rides in the MFMA shadow; a caller's noise tensor (the parity tests) or no noise at all
  // is handled in the epilogue
  if (a.mu_only) return launch_chain2_walk(a, st, chain2_wg_rows(a));
  if (chain2_wg_rows(a) == 192) return a.rng ? launch_chain2_t<true, 0, 3>(a, st) : launch_chain2_t<false, 0, 3>(a, st);
  return a.rng ? launch_chain2_t<true, 0, 4>(a, st) : launch_chain2_t<false, 0, 4>(a, st);
}

// RefineNet's inference pass (k_chain's REFINE launches without saved activations): the tile-walking form with rect_net's
// weights and the interval head.  Rows per workgroup as for the single-step denoiser launches.
bool chain2_refine_eligible(const ChainArgs& a) {
  if (!a.init || !a.out || !a.scores) return false;
  const int saves = (a.h1_save != nullptr) + (a.h2_save != nullptr) + (a.pre_save != nullptr);
  if (saves != 0 && saves != 3) return false;   // inference, or the training forward pass with all three buffers
  if (a.rows_per_scene % 16 != 0 || a.rows_per_scene < 48) return false;
  if (a.pooled && (a.S <= 0 || a.n_shards <= 0 || a.rows_per_scene != 3 * a.S || a.S % a.n_shards != 0)) return false;
  return true;
}

int launch_chain2_refine(const ChainArgs& a0, hipStream_t st) {
  ChainArgs a = a0;
  a.mu_only = 1, a.step_hi = a.step_lo = 1;   // (one evaluation per row; the row-tile choice of the single-step form)
  const int rows = chain2_wg_rows(a);
  return a.h1_save ? launch_chain2_refine_train(a, st, rows) : launch_chain2_refine_infer(a, st, rows);
}
#endif

#if PSTL_C2_ALL || PSTL_C2_PART == 1
int launch_chain2_walk(const ChainArgs& a, hipStream_t st, int rows) {
  return rows == 128 ? launch_chain2_t<false, 1, 2>(a, st) : rows == 192 ? launch_chain2_t<false, 1, 3>(a, st)
                                                                           : launch_chain2_t<false, 1, 4>(a, st);
}
int launch_chain2_refine_infer(const ChainArgs& a, hipStream_t st, int rows) {
  return rows == 128 ? launch_chain2_t<false, 2, 2>(a, st) : rows == 192 ? launch_chain2_t<false, 2, 3>(a, st)
                                                                           : launch_chain2_t<false, 2, 4>(a, st);
}
#endif

#if PSTL_C2_ALL || PSTL_C2_PART == 2
int launch_chain2_refine_train(const ChainArgs& a, hipStream_t st, int rows) {
  return rows == 128 ? launch_chain2_t<false, 3, 2>(a, st) : rows == 192 ? launch_chain2_t<false, 3, 3>(a, st)
                                                                           : launch_chain2_t<false, 3, 4>(a, st);
}
#endif

}  // namespace pstl
