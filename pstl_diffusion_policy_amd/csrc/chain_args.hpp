// chain_args.hpp -- what the two translation units of the denoiser / RefineNet MLP chain share: the packed-weight offsets of
// one network, the launch arguments, the split-f16 scale factors, and the entry point of the row-stationary kernel
// (chain2_kernels.hip) that mlp_kernels.hip dispatches large multi-step launches to.
#pragma once
#include <hip/hip_runtime.h>

namespace pstl {

constexpr int kMaxLaunchSteps = 128;  // reverse steps per launch (longer segments are split by pstl_rollout)

constexpr float kSplitW = 1024.0f;   // weights are split as pieces of 2^10 w
constexpr float kSplitX = 16.0f;     // activations as pieces of 2^4 x (|x| < 4094; absolute error floor 2^-29)


struct ChainOff {       // one of policy_net / rect_net
  long w1f;             // [224][256]  scene columns, transposed
  long b1;              // [256]
  long w1t;             // [32][256]   timestep columns, transposed (policy only)
  long w1x;             // A-operand layout [16 T][3 q][4 r][64 lanes]
  long w2;              // A-operand layout [16 T][16 q][4 r][64]
  long b2;              // [256]
  long w3;              // A-operand layout [3 j][16 T][4 r][64]
  long b3;              // [48]
  // split-bf16 A operands (v_mfma_f32_16x16x32_bf16): 8 words per (tile, k-block, lane): 4 of bf16 "hi" pairs, 4 of "lo"
  long w1xb;            // [16 T][2 kb][8][64]
  long w2b;             // [16 T][8 kb][8][64]
  long w3b;             // [3 j][8 kb][8][64]
  // split-f16 A operands (v_mfma_f32_16x16x32_f16), same shapes: pieces of kSplitW * w (see k_pack_a_split)
  long w1xh, w2h, w3h;
};

struct ChainArgs {
  long N;
  long plan_N;           // cfg->plan_rows: the row count the k_chain / k_chain2 choice of chain_waves = 0 is made for (0: N)
  int rows_per_scene;
  int steps;
  int step_hi, step_lo;
  int mu_only;
  int n_emit;
  int clip;
  float w_max, a_max;
  ChainOff off;
  const float* packed;
  const float* base;     // (bs,256)
  const float* tbias;    // (steps,256) or null (refine)
  const float* stlp;     // (N,6)
  const float* hl;       // (N,)
  const float* beta;
  const float* alpha;
  const float* alpha_hat;
  const float* noise;    // (steps-1,N,40) or null
  int tiles_per_group;   // 16-row tiles owned by one workgroup (4..kG)
  int rng;               // draw the noise in the kernel (seed, row_offset)
  unsigned long long seed;
  const unsigned long long* seed_dev;   // cfg->dyn: the seed is read from device memory instead (HIP-graph replay)
  long row_offset;
  float* x_inout;        // (N,40)
  float* emit_out;       // (n_emit,N,40)
  // refine
  const float* init;     // (N,40)
  const float* pooled;   // (bs,3,n_shards,40) or null
  const float* scores;   // (N,)
  float* out;            // (N,40)
  int S, n_shards;
  // training (N1): activations of rect_net kept for the backward pass; null = inference
  float* h1_save;        // (N,256) relu(layer 1)
  float* h2_save;        // (N,256) relu(layer 2)
  float* pre_save;       // (N,40)  layer-3 output before tanh
  unsigned* status;      // word 2 of the packed buffer's status block: set when a split-f16 launch leaves a non-finite value
};


// k_chain2 (chain2_kernels.hip): the multi-step denoiser launch with the ROWS stationary in registers (a wave owns 64 rows
// and all 256 hidden features) and the split-f16 weights streamed L2 -> LDS; same arithmetic as k_chain's default form, other
// summation order.  chain2_eligible: the launches it takes (see there).  Returns a PSTL_* status.
bool chain2_eligible(const ChainArgs& a);
int chain2_wg_rows(const ChainArgs& a);   // rows per workgroup the launch would use: 256; 192 for some multi-step, 128 for some single-step launches
long chain2_step_cost(const ChainArgs& a);  // a reverse step of the launch, in per cent of one 256-row tile-step (rounds x the tile's cost)
int launch_chain2(const ChainArgs& a, hipStream_t st);
bool chain2_refine_eligible(const ChainArgs& a);   // RefineNet's inference pass (init / pooled / scores / out set, nothing saved)
int launch_chain2_refine(const ChainArgs& a, hipStream_t st);

}  // namespace pstl
