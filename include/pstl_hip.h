/* pstl_hip.h -- C ABI of libpstl_hip.so: the MI355X (gfx950) implementation of the DDPM sampling +
 * STL-guidance hot path of mengyuest/pSTL-diffusion-policy.
 *
 * The reference has no FFI/plugin layer: its boundary for this path is the Python surface
 * (nusc_train.py: diffusion_rollout :557, compute_stl_dense :318, generate_trajs :39, get_diffusion_coeffs :528;
 * nusc_model.py: Net.encode_feat :55, Net.forward :97, Net.rect_forward :182).  A maintainer binds the entry
 * points below with ctypes; each one names the reference lines it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to float32 unless stated otherwise; tensors are dense, row-major;
 *  - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns immediately;
 *    return value: 0 = ok, negative = error (pstl_error_string);  nothing throws, nothing allocates;
 *  - rows: N = bs * rows_per_scene, row r belongs to scene r / rows_per_scene.  The reference's row order is
 *    r = (b*S + s)*3 + mode (nusc_train.py:20,724-754) => rows_per_scene = 3*S.  Passing rows_per_scene = 1 and
 *    bs = N gives the reference's "dense" (row-replicated) layout for the scene tensors.
 *  - row buffers ((N,40) controls, (N,T,4) states and the like) are accessed 16 bytes at a time and must be 16-byte
 *    aligned (PSTL_ERR_ARG otherwise); every torch allocation and every row-wise slice of one is;
 *  - thread-safe for distinct streams; no global mutable state.
 */
#ifndef PSTL_HIP_H
#define PSTL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSTL_ABI_VERSION 6   /* 2: pstl_encode_scene takes a work buffer; --joint and --refinement entry points
                                3: status block in the packed weight buffer (pstl_packed_status_offset)
                                4: pstl_cfg.dyn -- run-time parameters in device memory (HIP-graph replay)
                                5: pstl_rollout_layout; chain_waves = 2 (the row-stationary denoiser kernel);
                                   pstl_train_create / pstl_train_destroy and the (empty) context argument of
                                   pstl_refine_backward are gone
                                6: pstl_cfg.plan_rows (every shard of a job on the same denoiser kernel); pstl_adam_step;
                                   the prepared lane table's fourth float is the reciprocal segment length */

/* compile-time shape of the path (reference defaults: nt=20, n_segs=15, hiddens=[256,256], feat 7*32) */
#define PSTL_T 20
#define PSTL_NSEG 15
#define PSTL_CTRL 40     /* nt*2 */
#define PSTL_HID 256
#define PSTL_FEAT 224
#define PSTL_NEI_PREP 12 /* floats per (scene, neighbour, t) in the prepared neighbour table */

enum {
  PSTL_OK = 0,
  PSTL_ERR_ARG = -1,        /* null pointer / bad size */
  PSTL_ERR_SHAPE = -2,      /* cfg asks for a shape this build is not specialised for */
  PSTL_ERR_LAUNCH = -3      /* hipGetLastError() after a launch */
};

enum {
  PSTL_FLAG_CLIP = 1,       /* clip normalised controls to +-max  (--diffusion_clip, nusc_train.py:651-653) */
  PSTL_FLAG_MAXIMIZE = 2,   /* guidance loss relu(100 - score)     (nusc_train.py:616-617)                 */
  PSTL_FLAG_CLIP_RECT = 4,  /* --clip_rect (nusc_model.py:230-233)                                         */
  PSTL_FLAG_NO_MERGE = 8,   /* rect_forward without merge_net pooling (not diverse_loss / --no_arch)       */
  PSTL_FLAG_RNG = 16,       /* kernels draw the diffusion noise themselves (cfg.seed, cfg.row_offset); the  */
                            /* `noise` / `z` pointer arguments are then ignored                              */
  PSTL_FLAG_KEEP_DH1 = 64,  /* pstl_refine_backward leaves dH1 (N,256) in its work buffer: pstl_encoder_backward follows      */
                            /* (--joint).  Without it the one-pass layer kernels never write dH1 to memory.                 */
  PSTL_FLAG_NORM_STL = 32   /* --norm_stl (nusc_train.py:88-91,97-113): the speed, lane-distance and clearance       */
                            /* predicates divided by v_factor = clip(vmax - vmin, 0.3), d_factor = clip((dmax - dmin)*5, */
                            /* 0.3), safe_factor = clip(dsafe, 0.3) in pstl_stl_forward / _backward / pstl_guidance_step; */
                            /* pstl_trajopt and pstl_refinement answer PSTL_ERR_SHAPE                                   */
};

/* Run-time parameters a caller may keep in DEVICE memory (pstl_cfg.dyn) instead of passing them by value: a sequence of
 * launches captured once in a HIP graph can then be replayed with another noise seed / guidance-loss scale by writing
 * these 16 bytes (the closed-loop caller does: one graph replay per simulation step, nusc_sim.py). */
typedef struct pstl_dyn {
  uint64_t seed;           /* replaces cfg.seed                                                            */
  float grad_scale;        /* replaces the grad_scale argument of pstl_guidance_step                        */
  float reserved;
} pstl_dyn;

typedef struct pstl_cfg {
  int32_t bs;              /* scenes in this shard                                        */
  int32_t rows_per_scene;  /* 3*S (scene-indexed) or 1 (dense)                            */
  int32_t S;               /* samples per (scene, mode): n_randoms == sampling_size       */
  int32_t K;               /* neighbours                                                  */
  int32_t steps;           /* --diffusion_steps                                           */
  int32_t n_shards;        /* --n_shards (merge_net max-pool groups)                      */
  int32_t flags;           /* PSTL_FLAG_*                                                 */
  int32_t chain_waves;     /* arithmetic of the MLP chains (policy_net, rect_net): 0 or 16 = default: every fp32
                              operand as two IEEE-half pieces (2^-23 per operand), three
                              v_mfma_f32_16x16x32_f16 products per fp32 product, fp32 accumulate -- as close to
                              the reference as an fp32 fmaf chain in another summation order (0: batches with
                              fewer than five 16-row tiles per CU -- the closed-loop caller's 192 rows -- run
                              their denoiser launches in a latency layout, 1..4 tiles per workgroup; 16: always
                              the throughput layout; the results are bit-identical; 0 also hands the multi-step
                              denoiser launches of batches that (nearly) fill rounds of 256- or 192-row workgroups --
                              from 45 072 rows (2 817 sixteen-row tiles) on 256 CUs -- and the single-step (mu_only = 1) launches of every batch
                              above the latency layout's sizes -- and RefineNet's inference pass of such batches -- to
                              the row-stationary kernel k_chain2: same arithmetic and domain,
                              another summation order; 2: k_chain2 for every launch it can take whatever the batch
                              size, the other launches as 16); 8 or 4 = fp32
                              MFMA (v_mfma_f32_16x16x4_f32) with 8 / 4 waves per workgroup; 32 = policy_net
                              on bfloat16 pieces (2^-17 per operand), rect_net on fp32 MFMA.
                              DOMAIN of 0 / 16 / 2: the pieces are halves of 2^10 w and 2^4 x, so the chain weights must
                              satisfy |w| < PSTL_SPLIT_F16_WMAX and every layer input |x| < 4094; outside it the
                              results are undefined (an overflowed piece is an infinity: NaNs where it reaches the
                              output, but max(NaN, 0) = 0 inside a ReLU) and the status block of the packed buffer
                              says so (pstl_packed_status_offset): every conversion to half is checked where it
                              happens.  8 / 4 / 32 have fp32's range.
                              Any other value: PSTL_ERR_SHAPE.                                                */
  float tau;               /* --smoothing_factor                                          */
  float thres;             /* --stl_nn_thres                                              */
  float w_max, a_max;      /* --mul_w_max, --mul_a_max                                    */
  float dt;                /* --dt                                                        */
  float ego_L, ego_W;      /* --ego_L, --ego_W                                            */
  float reserved_f;
  uint64_t seed;           /* PSTL_FLAG_RNG: noise stream                                 */
  int64_t row_offset;      /* PSTL_FLAG_RNG: global index of this shard's first row, so that the noise of a row */
                           /* does not depend on how the batch is split over GPUs                                */
  const pstl_dyn* dyn;     /* device pointer or NULL.  Non-null: pstl_fill_normal, pstl_rollout and pstl_guidance_step */
                           /* read seed (and grad_scale) from it at kernel start and ignore the by-value ones           */
  int64_t plan_rows;       /* chain_waves = 0 chooses between the two denoiser kernels (k_chain / k_chain2: the same     */
                           /* arithmetic in two summation orders, last bits differ) by batch size.  0: by THIS call's    */
                           /* rows -- a shard evaluated alone may then get other last bits than the same rows inside a   */
                           /* larger batch.  > 0: by this row count instead, whatever the call's own size: a job passes  */
                           /* the same number (its nominal rows per GPU) on every shard and every call, so that all of   */
                           /* its rows go through the same kernel and a row's bits do not depend on the shard that       */
                           /* holds it (the work-group layouts WITHIN a kernel never change results).                     */
} pstl_cfg;

/* state_dict blobs of the reference Net (nusc_model.py:20-46; keys "<net>.{0,2,4}.{weight,bias}").
 * weight = (out,in) row-major exactly as nn.Linear stores it.  merge/rect may be null (no --rect_head). */
typedef struct pstl_mlp3 {
  const float *w0, *b0, *w1, *b1, *w2, *b2;
} pstl_mlp3;

typedef struct pstl_weight_ptrs {
  pstl_mlp3 ego_encoder;       /* 6   -> 256 -> 256 -> 32  */
  pstl_mlp3 neighbor_encoder;  /* 7   -> 256 -> 256 -> 32  */
  pstl_mlp3 lane_encoder;      /* 45  -> 256 -> 256 -> 32  */
  pstl_mlp3 policy_net;        /* 303 -> 256 -> 256 -> 40  */
  pstl_mlp3 merge_net;         /* 40  -> 32  -> 32  -> 40  */
  pstl_mlp3 rect_net;          /* 271 -> 256 -> 256 -> 40  */
} pstl_weight_ptrs;

int pstl_version(void);
const char* pstl_error_string(int code);

/* ---- weights ----------------------------------------------------------------------------------------------- */
/* Number of floats of the packed (kernel-layout) weight buffer. */
size_t pstl_packed_weight_floats(void);
/* Float offset, inside the packed buffer, of its 16-word STATUS BLOCK -- the only part of the buffer that is written after
 * pstl_pack_weights, and the way the asynchronous entry points report the domain of the default (split-f16) arithmetic:
 *   word 0 (float): max |w| over the weights of policy_net that the MLP-chain kernel carries as half pieces (layer-1
 *                   columns of x / highlevel / stlp, layers 2 and 3), written by pstl_pack_weights; NaN if one is NaN;
 *   word 1 (float): the same for rect_net (0 without --rect_head weights);
 *   word 2 (uint32): 0 after pstl_pack_weights; set to 1 by pstl_rollout / pstl_refine / pstl_refine_train_forward when a
 *                   launch on the split-f16 arithmetic (chain_waves 0 / 16) ran outside the domain above: word 0 / 1 not
 *                   below PSTL_SPLIT_F16_WMAX, a layer input (state, hidden activation) whose half piece overflowed, or a
 *                   non-finite state / RefineNet output.  Sticky: the caller reads it when it synchronises anyway (after
 *                   the timed region), re-runs the batch with chain_waves = 8 and may clear the word.
 * A caller reads words 0-1 once after packing (one synchronisation) and selects chain_waves = 8 when either is not
 * below PSTL_SPLIT_F16_WMAX (engine.PackedWeights does). */
#define PSTL_SPLIT_F16_WMAX 63.9f
size_t pstl_packed_status_offset(void);
/* Re-lays the state_dict out for the kernels (MFMA operand order, transposed encoder matrices).
 * Replaces: Net.load_state_dict + the implicit layout of nn.Linear (nusc_train.py:1213-1215). */
int pstl_pack_weights(const pstl_weight_ptrs* w, float* packed, void* stream);
/* The same for the networks an optimiser step changed: packs again, IN PLACE, the networks whose six pointers are all non-null
 * in `w`, leaves the others as they are (RefineNet training without --joint changes rect_net only: 15 launches instead of 75
 * per step).  The max |w| status word of a re-packed chain is recomputed; the sticky domain word 2 is left alone. */
int pstl_repack_weights(const pstl_weight_ptrs* w, float* packed, void* stream);
/* tbias[t][h] = sum_k W1[h][264+k] * pe(t)[k] for t in [0,steps): the timestep embedding folded into a layer-1
 * bias.  Replaces Net.pos_encoding (nusc_model.py:48-53) + its 32 columns of policy_net layer 1. */
int pstl_time_bias(const float* packed, int steps, float* tbias /* (steps,256) */, void* stream);

/* ---- per-batch scene preparation ---------------------------------------------------------------------------- */
/* neighbors_traj (bs,K,T,7) = valid,x,y,th,v,L,W ; lanes (bs,15,3) x,y,th.
 * nei_prep (bs,K,T,12): valid, r, cx[4], cy[4], 0, 0  (circle row of each neighbour, utils.py:465-497);
 * lane_prep (bs,3,15,4): x, y, th, 1 / clamp(length of the segment to the next waypoint, 1e-7) -- 0 for a segment of length 0
 *   and for the last waypoint (ABI 6; until ABI 5: the length itself) -- for curr,left,right;
 *   opaque to the caller: the STL entry points below take exactly what this call wrote. */
int pstl_prepare_scene(const pstl_cfg* cfg, const float* neighbors_traj, const float* currlane, const float* leftlane,
                       const float* rightlane, float* nei_prep, float* lane_prep, void* stream);

/* Scene encoder.  Replaces Net.encode_feat (nusc_model.py:55-95) and the scene-constant 224 columns of layer 1 of
 * policy_net / rect_net: base_x[b][h] = bias1[h] + sum_k W1[h][k] * feature[b][k].
 * ego0 (bs,6) = ego_traj[:,0]; neighbors (bs,K,7); lanes (bs,15,3); ids (bs,) each. base_rect may be null.
 * work: pstl_encode_scene_work_floats(cfg) floats (token inputs and the activations between the three GEMM layers). */
size_t pstl_encode_scene_work_floats(const pstl_cfg* cfg);
int pstl_encode_scene(const pstl_cfg* cfg, const float* packed, const float* ego0, const float* neighbors,
                      const float* currlane, const float* leftlane, const float* rightlane, const float* curr_id,
                      const float* left_id, const float* right_id, float* work, float* feature /* (bs,224) */,
                      float* base_policy /* (bs,256) */, float* base_rect /* (bs,256) or null */, void* stream);

/* ---- reverse diffusion --------------------------------------------------------------------------------------- */
/* Runs reverse steps i = step_hi ... step_lo (step_hi >= step_lo >= 1) of diffusion_rollout (nusc_train.py:568-630)
 * for all N rows in ONE launch:  eps = policy_net(...) + x ; mu = (x - (1-a_i)/sqrt(1-ah_i) eps)/sqrt(a_i) ;
 * x <- mu + sqrt(b_i) z.   noise (steps-1, N, 40): noise[k] is used at step i = steps-1-k and ignored at i == 1
 * (the reference draws zeros there).  mu_only (requires step_hi == step_lo): 1 = x_inout receives mu and no noise
 * is added (the guidance kernel finishes the step); 2 = x_inout receives the predicted noise eps itself, i.e. one
 * Net.forward evaluation (nusc_model.py:159-162).
 * emit: for every step with i <= n_emit the new state, normalised (x * (w_max,a_max), clipped iff PSTL_FLAG_CLIP -- with
 * torch.clip's semantics: a NaN stays a NaN), is
 * written to emit_out[n_emit - i] (N,40) -- i.e. the last n_emit entries of the reference's diff_full list
 * (nusc_train.py:633-634); n_emit may be 0. */
int pstl_rollout(const pstl_cfg* cfg, float* packed /* status block written */, const float* base_policy, const float* tbias,
                 const float* stlp /* (N,6) */, const float* hl /* (N,) */, const float* beta, const float* alpha,
                 const float* alpha_hat, const float* noise, int step_hi, int step_lo, int mu_only,
                 float* x_inout /* (N,40) */, float* emit_out, int n_emit, void* stream);

/* Which kernel and layout pstl_rollout picks for a launch of cfg's batch with step_hi > step_lo (multi_step != 0) or
 * step_hi == step_lo with mu_only = 1 (0: the guided phase's launches): a query for benchmarks and tools, so that they
 * need not re-derive the rules of csrc/.  Writes
 * kernel: 0 = k_chain, latency layout (tiles_per_group 1..5 sixteen-row tiles per workgroup, empty pipeline slots skipped);
 *         1 = k_chain, throughput layout (tiles_per_group 5..12);  2 = k_chain2 (one workgroup per CU at a time, of
 *         tiles_per_group = 16 sixteen-row tiles = 256 rows, or 12 = 192 rows where that fills the rounds better; a
 *         single-step launch starts one workgroup per CU, which walks `rounds` tiles of 16 -- or 8, where 256-row tiles
 *         would leave CUs idle);
 *         3 = an exact-fp32 / bfloat16-piece variant (chain_waves 8, 4, 32: throughput layout);
 * rounds: how many waves of workgroups the launch takes on this device's CUs.  Touches no GPU memory. */
int pstl_rollout_layout(const pstl_cfg* cfg, int multi_step, int* kernel, int* tiles_per_group, int* rounds);


/* out (N,40) = the N(0,1) values the kernels draw under PSTL_FLAG_RNG for reverse step `step` (step == cfg->steps
 * is the stream of the initial state x_T, reference nusc_train.py:563).  Philox4x32-7 (rng.hpp: kPhiloxRounds) + Box-Muller, a pure function of
 * (cfg->seed, cfg->row_offset + row, column, step). */
int pstl_fill_normal(const pstl_cfg* cfg, int step, float* out, void* stream);

/* ---- dynamics + STL robustness ------------------------------------------------------------------------------- */
/* generate_trajs (nusc_train.py:29-49): trajs (N,21,4) from s0 (bs,4) and controls (N,40) in physical units. */
int pstl_generate_trajs(const pstl_cfg* cfg, const float* s0, const float* controls, float* trajs, void* stream);

/* For rep in [0,reps), row r: traj = unicycle(s0[scene], controls[rep][r]) (generate_trajs, nusc_train.py:29-49),
 * score = compute_stl_dense(...) (nusc_train.py:318-345; formulas :95-140; stl_d_lib.py).  controls (reps,N,40) are
 * in physical units (already normalised).  If states != null ((reps,N,20,4), what compute_stl_dense receives as
 * "ego_traj") they are scored as given and s0/controls may be null.
 * scores (reps,N); scores3 (3,reps,N) or null (the three formulas before mode select).
 * If sel_controls != null the best rep per row (max score, lowest rep on ties; nusc_train.py:1006-1007) is written to
 * sel_controls (N,40), sel_scores (N,), sel_idx (N,) int32. */
int pstl_stl_forward(const pstl_cfg* cfg, const float* s0 /* (bs,4) */, const float* controls, const float* states,
                     int reps, const float* nei_prep, const float* lane_prep, const float* stlp, const float* hl,
                     float* scores, float* scores3, float* sel_controls, float* sel_scores, int32_t* sel_idx,
                     void* stream);

/* prep_stl_cache (nusc_train.py:74-93): the seven signals the formulas read, signals (7,N,T) = x2curr_d, x2curr_th,
 * x2left_d, x2left_th, x2right_d, x2right_th (compute_t2l_dist, nusc_api.py:685-739) and min_nei_d
 * (compute_shortest_dist_refined, nusc_train.py:142-148), for given states (N,T,4) or for the rollout of controls (N,40)
 * from s0 (bs,4) (states == null).  The fused scoring kernels never materialise these; this entry point serves callers of
 * prep_stl_cache and the generic formula evaluator (--norm_stl). */
int pstl_stl_signals(const pstl_cfg* cfg, const float* s0, const float* controls, const float* states,
                     const float* nei_prep, const float* lane_prep, float* signals, void* stream);

/* dcontrols[r] = dscore[r] * d score[r] / d controls[r]   (what autograd gives through compute_stl_dense +
 * generate_trajs).  dscore null = all ones. */
int pstl_stl_backward(const pstl_cfg* cfg, const float* s0, const float* controls, const float* nei_prep,
                      const float* lane_prep, const float* stlp, const float* hl, const float* dscore,
                      float* dcontrols /* (N,40) */, float* scores /* (N,) or null */, void* stream);

/* Guidance block of one reverse step (nusc_train.py:599-628): niters Adam iterations on mu (un-normalised, (N,40))
 * of loss = mask_mean(relu(thres - score), valid), then x = mu + sqrt(beta_i) z.  Observable reference behaviour
 * (see oracle/pstl_oracle.py:guidance_update): iteration 0 is a plain Adam step; later iterations are
 * anchor + clip(|mu - anchor|, -beta_i, beta_i).  valid (N,) 0/1; grad_scale = (1/clip(mean(valid),1e-2))/N_global.
 * adam_neg_step / adam_bc2_sqrt: host arrays [niters] = float(-lr/(1-0.9^j)), float(sqrt(1-0.999^j)), j=1..niters.
 * work (3,N,40) scratch (m, v, anchor), only touched when niters > 1.  z (N,40) or null (= zeros, the i == 1 step);
 * under PSTL_FLAG_RNG z is ignored and the noise of reverse step `step` is drawn in the kernel (none at step 1).
 * emit_out (N,40) or null: normalised new state, as in pstl_rollout. */
int pstl_guidance_step(const pstl_cfg* cfg, const float* s0, const float* nei_prep, const float* lane_prep,
                       const float* stlp, const float* hl, const float* valid, float grad_scale, int niters,
                       const float* adam_neg_step, const float* adam_bc2_sqrt, float beta_i, int step, const float* z,
                       float* mu_x_inout /* (N,40): mu in, x out */, float* work, float* emit_out, void* stream);

/* ---- RefineNet ------------------------------------------------------------------------------------------------ */
/* Net.rect_forward (nusc_model.py:182-235) with --interval, diverse_fuse_type "add":
 * pooled = max over each shard of S/n_shards samples of merge_net(init) per (scene, mode); fused = init + pooled;
 * raw = tanh(rect_net([feature|hl|stlp|fused])); out = init + interval(raw, init) * [score < 0].
 * pooled_work (bs,3,n_shards,40) scratch.  Requires rows_per_scene == 3*S. */
int pstl_refine(const pstl_cfg* cfg, float* packed /* status block written */, const float* base_rect, const float* stlp, const float* hl,
                const float* init_controls /* (N,40) */, const float* scores /* (N,) */, float* pooled_work,
                float* out_controls /* (N,40) */, void* stream);

/* ---- RefineNet training step (config 5, nusc_train.py:1400-1427,1522-1525) ----------------------- */
/* pstl_refine with the activations kept for the backward pass: h1_save, h2_save (N,256) = relu of layers 1, 2;
 * pre_save (N,40) = layer-3 output before tanh. */
int pstl_refine_train_forward(const pstl_cfg* cfg, float* packed, const float* base_rect, const float* stlp,
                              const float* hl, const float* init_controls, const float* scores, float* pooled_work,
                              float* out_controls, float* h1_save, float* h2_save, float* pre_save, void* stream);
/* Loss mask_mean(relu(thres - score), valid) (compute_policy_loss :411): dscore[r] = -grad_scale*valid[r]*[thres-score>0]
 * (grad_scale as in pstl_guidance_step); loss_parts[256] partial sums of relu(thres - score)*valid (host adds them and
 * divides by N * clip(mean(valid), 1e-2)). */
int pstl_loss_grad(const pstl_cfg* cfg, const float* scores, const float* valid, float grad_scale, float* dscore,
                   float* loss_parts, void* stream);
/* The optimiser of the training loop (reference nusc_train.py:1233: torch.optim.Adam over rect_net.parameters(), or over
 * net.parameters() with --joint; :1522-1525 zero_grad / backward / step) on the device path: ONE step of Adam with torch's
 * defaults (no weight decay, no amsgrad) over up to PSTL_ADAM_MAX_TENSORS tensors, operation for operation what torch's
 * float32 update computes (csrc/adam_core.hpp; bit for bit against torch.optim.Adam in the tests), in one launch.
 *   params / grads : HOST arrays of n_tensors DEVICE pointers (the live parameter tensors, updated in place, and their
 *                    gradients), numel: HOST array of their element counts;
 *   exp_avg, exp_avg_sq : the moments of all tensors back to back (sum of numel floats each), zeroed by the caller before the
 *                    first step, otherwise owned by this function;
 *   sched (sched_steps,2) : DEVICE table of the per-step scalars -lr / (1 - beta1^t), sqrt(1 - beta2^t), t = 1 ... sched_steps,
 *                    which the caller computes in double precision as torch does and rounds to float32;
 *   step           : DEVICE counter of the steps done (0 before the first); the launch reads its scalars at sched[*step] and
 *                    a second, one-thread launch increments it -- nothing about a step is passed by value, so a captured
 *                    training step replays with the right bias corrections.  *step >= sched_steps reads the table's LAST
 *                    entry: a table long enough for float32(1 - beta^t) to have become 1.0 for both betas (17 323 steps for
 *                    (0.9, 0.999)) ends at the scalars' limits and is exact for every later step.
 *   one_minus_beta1, beta2, one_minus_beta2, eps : torch's Python scalars as float32 -- the differences 1 - beta formed in
 *                    DOUBLE precision first (float32(1 - 0.999) is not 1 - float32(0.999)).
 * The caller re-packs the networks whose tensors moved (pstl_repack_weights) before the kernels read them again. */
#define PSTL_ADAM_MAX_TENSORS 32
int pstl_adam_step(int n_tensors, float* const* params, const float* const* grads, const int64_t* numel, float* exp_avg,
                   float* exp_avg_sq, const float* sched, int sched_steps, int32_t* step, float one_minus_beta1, float beta2,
                   float one_minus_beta2, float eps, void* stream);
size_t pstl_train_work_floats(const pstl_cfg* cfg);
/* d loss / d rect_net parameters given dcontrols = d loss / d out_controls (from pstl_stl_backward).  w2, w3: the
 * reference-layout weights rect_net.2.weight (256,256), rect_net.4.weight (40,256).  Gradients in the reference layout:
 * dw1 (256,271), db1 (256), dw2 (256,256), db2 (256), dw3 (40,256), db3 (40).  Gradients w.r.t. the scene encoders and
 * merge_net (needed only with --joint: the reference's optimiser holds rect_net.parameters() otherwise,
 * nusc_train.py:1230-1233) come from pstl_encoder_backward / pstl_merge_backward below, which continue from what this
 * call leaves in `work`.  work: pstl_train_work_floats(cfg) floats.
 * Arithmetic: the activation-gradient products (dH2, dH1) are split-bf16 MFMA products (2^-17 per operand, fp32 range) in
 * every mode; the weight gradients dw1[:, 224:], dw2, dw3 likewise (each layer's two contractions share one pass over the
 * saved activations) unless cfg->chain_waves is 8 or 4 (the exact-fp32 request) or rows_per_scene is not a multiple of 32,
 * where they are fp32-MFMA contractions in launches of their own; dw1[:, :224] and the bias gradients are always fp32.
 * Against the reference's autograd: rtol 5e-3 (tested).  pstl_encoder_backward needs PSTL_FLAG_KEEP_DH1 in cfg->flags. */
int pstl_refine_backward(const pstl_cfg* cfg, const float* w2, const float* w3, const float* feature,
                         const float* stlp, const float* hl, const float* init_controls,
                         const float* pooled /* (bs,3,n_shards,40) from the forward call; null with PSTL_FLAG_NO_MERGE */,
                         const float* prev_scores, const float* h1, const float* h2, const float* pre,
                         const float* dcontrols, float* work, float* dw1, float* db1, float* dw2, float* db2, float* dw3,
                         float* db3, void* stream);
/* ---- --joint (reference nusc_train.py:1230-1231: Adam over net.parameters()) ------------------------------------
 * With --rect_head the loss reaches the scene encoders through `feature` (Net.forward(get_feature=True), nusc_model.py:
 * 55-95 -> Net.rect_forward :182-209) and merge_net through the pooled columns (:186-200); policy_net receives no
 * gradient (the rollout runs under no_grad and loss_diffusion is not in the --rect_head losses, :455-467).
 *
 * pstl_encode_scene_saved: pstl_encode_scene with its work buffer laid out by the caller -- per token, what the encoders'
 * backward needs.  Tokens are
 * ordered [bs ego | bs*K neighbours (scene-major) | 3*bs lanes (scene-major)], T = bs*(K+4):
 * tok_in (T,48) inputs (6 / 7 / 45 valid columns), tok_h1, tok_h2 (T,256) the two hidden layers after ReLU, tok_out (T,32). */
int pstl_encode_scene_saved(const pstl_cfg* cfg, const float* packed, const float* ego0, const float* neighbors,
                            const float* currlane, const float* leftlane, const float* rightlane, const float* curr_id,
                            const float* left_id, const float* right_id, float* feature, float* base_policy,
                            float* base_rect, float* tok_in, float* tok_h1, float* tok_h2, float* tok_out, void* stream);
/* Gradients of {ego,neighbor,lane}_encoder (arrays of 3 device pointers, in that order; reference layouts:
 * d_w0 (256,6|7|45), d_b0 (256), d_w1 (256,256), d_b1 (256), d_w2 (32,256), d_b2 (32)).  Call right after
 * pstl_refine_backward on the same stream with the SAME cfg and its work buffer `refine_work` untouched (d loss / d
 * hidden-1 and its per-scene sums are read from it; its scratch regions are reused).  rect_w1 = rect_net.0.weight
 * (256,271); enc_w1[e] = <encoder>.2.weight (256,256), enc_w2[e] = <encoder>.4.weight (32,256), live reference-layout
 * tensors.  work: pstl_encoder_backward_work_floats(cfg).  dfused (N,40) or null: WRITTEN with d loss / d (rect_net
 * input columns 231..270), the input of pstl_merge_backward. */
size_t pstl_encoder_backward_work_floats(const pstl_cfg* cfg);
int pstl_encoder_backward(const pstl_cfg* cfg, float* refine_work, const float* rect_w1, const float* const* enc_w1,
                          const float* const* enc_w2, const float* tok_in, const float* tok_h1, const float* tok_h2,
                          const float* tok_out, float* work, float* const* d_w0, float* const* d_b0, float* const* d_w1,
                          float* const* d_b1, float* const* d_w2, float* const* d_b2, float* dfused, void* stream);
/* Gradients of merge_net (40 -> 32 -> 32 -> 40; reference layouts dw0 (32,40), db0 (32), dw1 (32,32), db1 (32),
 * dw2 (40,32), db2 (40)) through the shard max-pool: the gradient of a pooled column goes to the first sample of the
 * shard holding the maximum.  packed: the weights the forward ran with.  dfused (N,ldf), first 40 columns used.
 * Requires rows_per_scene == 3*S, S <= 64, n_shards*40 <= 256 (as the forward and the DPP loss do).
 * work: pstl_merge_backward_work_floats(cfg). */
size_t pstl_merge_backward_work_floats(const pstl_cfg* cfg);
int pstl_merge_backward(const pstl_cfg* cfg, const float* packed, const float* init_controls, const float* dfused, int ldf,
                        float* work, float* dw0, float* db0, float* dw1, float* db1, float* dw2, float* db2, void* stream);

/* e7 training objective (--diverse_loss, reference nusc_train.py:442-467).  For every (scene, mode, shard) group of
 * S/n_shards samples the DPP diversity tr(I - (L+I)^-1), L = diag(q) exp(-diversity_scale |x_i - x_j|) diag(q),
 * x = rect_controls/(w_max,a_max), q = exp(score)[score>0] (detach != 0: q = [score>0], no gradient into the scores).
 * group_div (bs*3*n_shards): the diversities (loss_diversity = -mean(group_div) * diversity_weight);
 * dcontrols (N,40), dscore (N): WRITTEN with d(loss_diversity)/d rect_controls and /d scores.
 * Optional regulariser loss_reg = mask_mean(square(rect - init), [score >= 0]) (:466): reg_out[0] = loss_reg (reg_out,
 * 2 floats, and reg_work, 512 doubles, may be null when rect_reg_weight == 0); dcontrols += rect_reg_weight * d loss_reg.
 * Requires rows_per_scene == 3*S and S/n_shards <= 64. */
int pstl_diversity_loss(const pstl_cfg* cfg, const float* rect_controls, const float* init_controls, const float* scores,
                        float diversity_scale, float diversity_weight, int detach, float rect_reg_weight, float* group_div,
                        float* reg_out, double* reg_work, float* dcontrols, float* dscore, void* stream);

/* ---- metrics --------------------------------------------------------------------------------------------------- */
/* counts[0] = #rows with score>0 and valid, counts[1] = #valid rows, counts[2] = #rows,
 * counts[3] = #(scene,mode) with any sample score>0 and valid, counts[4] = #valid (scene,mode), counts[5] = 3*bs.
 * (acc / scene_acc numerators and denominators, nusc_train.py:332,339-343.)  counts: 8 x uint64, zeroed by the call.
 * sat_mask (N,) uint8 or null. */
int pstl_reduce_metrics(const pstl_cfg* cfg, const float* scores, const float* valid /* (N,) */, uint64_t* counts,
                        uint8_t* sat_mask, void* stream);

/* The closed-loop caller's choice (reference nusc_sim.py:677-683: the scores of modes 1 and 2 set to -10000, torch.argmax over
 * the (S,3) scores of the ONE scene of the batch, that row's controls): out4[0..1] = the first (w, a) of the best lane-keeping
 * sample, out4[2] = its score, out4[3] = the bit pattern of the chain-domain status word (packed_status = packed +
 * pstl_packed_status_offset() + 2, or NULL), so that control and flag reach the host in one 16-byte copy.  cfg->bs must be 1. */
int pstl_select_plan(const pstl_cfg* cfg, const float* scores /* (3*S,) */, const float* controls /* (3*S,40) */,
                     const float* packed_status, float* out4, void* stream);

/* ---- post-sampling diversity metrics (the numbers run_sampling_test prints after its timer) -------- */
/* measure_diversity (nusc_api.py:817-875), measure_extra_diversity (nusc_api.py:894-936, compute_entropy
 * utils.py:388-417, compute_area nusc_api.py:878-891) and compute_ade_fde (nusc_train.py:877-887) for the final
 * controls (N,40) (physical units; the trajectories are rolled out in the kernel), their scores (N,) and valid (N,).
 * gt_traj (bs,T,gt_stride): ground-truth ego states, components 0..3 = x,y,th,v (the dataset's ego_traj has stride 6).
 * alphas: DEVICE array of 11 floats = torch.linspace(0,1,11) (the entropy bin fractions, utils.py:406).
 * per_mode (bs,3,8) float64: std, vol, ent_s, sum_t ent_w, sum_t ent_a, area, #satisfied samples, lane valid (0/1);
 * per_scene (bs,2) float32: min-over-rows ADE and FDE (squared-error form, as the reference defines them);
 * totals (12) float64 or null: [0] sum std*valid [1] sum vol*valid [2] #valid (scene,mode) [3] sum ent_s [4] sum ent_w
 * [5] sum ent_a [6] sum area [7] #(scene,mode) [8] sum ADE [9] sum FDE [10] #scenes [11] 0 -- the reference's printed
 * values are the ratios (std = [0]/[2], vol = [1]/[2], ent_s = [3]/[7], ent_w = [4]/([7]*T), area = [6]/[7],
 * ade = [8]/[10]); sums of shards add, so a sharded run all-gathers these 12 numbers.  Requires S <= 64. */
int pstl_diversity(const pstl_cfg* cfg, const float* s0, const float* gt_traj, int gt_stride, const float* controls,
                   const float* scores, const float* valid, const float* alphas, double* per_mode, float* per_scene,
                   double* totals, void* stream);

/* ---- generic STL formulas (the operator library stl_d_lib.py; SURVEY 8b "formulas are callable") ------------------ */
/* A formula tree is flattened by the host into postfix order (children before parents; the last node is the root).
 * Node types: SIGNAL a = index of an input signal (an AP's expression, stl_d_lib.py:70-84);  NOT a (:125);
 * AND a,b / OR a,b = soft min / soft max of two nodes (:87,:113);  LISTAND = soft min over lists[list_off .. +n_list)
 * (:97);  ALWAYS / EVENTUALLY a with window [t+ts, t+te) clipped to [0,T) (:144-169; Once :171 is EVENTUALLY with
 * ts < 0).  PSTL_STL_FLAG_SOFT: the node ignores `hard` (the cumulative operators of UntimedUntil, :183-193). */
enum {
  PSTL_STL_SIGNAL = 0, PSTL_STL_NOT = 1, PSTL_STL_AND = 2, PSTL_STL_OR = 3, PSTL_STL_LISTAND = 4, PSTL_STL_ALWAYS = 5,
  PSTL_STL_EVENTUALLY = 6
};
#define PSTL_STL_FLAG_SOFT 1
#define PSTL_STL_MAX_T 1024
typedef struct pstl_stl_node {
  int32_t op, a, b, ts, te, n_list, list_off, flags;
} pstl_stl_node;

/* Robustness of the whole tree for n rows: signals (n_sig,n,T) -> out (n,T) (the value of the root at every t, as
 * formula(x, tau) returns it in the reference).  nodes / lists are DEVICE arrays.  vals (n_nodes,T,n): workspace that
 * keeps every node's value for the adjoint.  hard != 0: max/min instead of logsumexp (d["hard"], stl_d_lib.py:10-11). */
int pstl_stl_program_forward(const pstl_stl_node* nodes, int n_nodes, const int32_t* lists, int64_t n, int T,
                             const float* signals, float tau, int hard, float* vals, float* out, void* stream);
/* dsignals (n_sig,n,T) += d(sum(out * dout)) / d signals; dsignals must be zeroed by the caller; adj (n_nodes,T,n)
 * scratch; vals from the forward call. */
int pstl_stl_program_backward(const pstl_stl_node* nodes, int n_nodes, const int32_t* lists, int64_t n, int T,
                              const float* vals, float tau, int hard, const float* dout, float* adj, float* dsignals,
                              void* stream);

/* ---- trajectory optimisation (the data-augmentation loop, nusc_train.py:1302-1325) ------------------- */
/* Runs `iters` iterations of  torch.optim.Adam([params], lr)  on
 *   loss = mean(relu(thres - score) * valid) / clip(mean(valid), 1e-3) + reg_loss * (mean(relu(w^2 - w_max^2)) +
 *          mean(relu(a^2 - a_max^2)))                                  (compute_trajopt_loss_lite, nusc_train.py:287-300)
 * for all N rows in ONE launch; params (N,40) are controls in physical units (the dataset's `params`, (bs,M,3,nt,2)).
 * grad_scale = (1/clip(mean(valid),1e-3))/N_global; reg_scale = reg_loss/(N_global*nt);
 * adam_neg_step / adam_bc2_sqrt: DEVICE arrays [iters] = float(-lr/(1-0.9^k)), float(sqrt(1-0.999^k)), k = first
 * iteration number (1-based) ... ; work: 3*N*40 floats of scratch the kernel owns for the run (the iterate and Adam's m and
 * v, element-major); m and v are read back when resume != 0, so a run can be split over several calls.  scores (N,) or null: robustness of the iterate the last update started from (what the
 * reference saves as scores_*.npy).  thres = --stl_trajopt_thres. */
int pstl_trajopt(const pstl_cfg* cfg, const float* s0, const float* nei_prep, const float* lane_prep, const float* stlp,
                 const float* hl, const float* valid, float thres, float grad_scale, float reg_scale, int iters,
                 const float* adam_neg_step, const float* adam_bc2_sqrt, int resume, float* params_inout, float* work,
                 float* scores, void* stream);

/* ---- --refinement (nusc_train.py:1034-1071, inside the timed region of run_sampling_test) ---------------------------- */
/* For every row whose `controls` score <= 0 on a valid lane: `iters` (50) iterations of torch.optim.Adam([lambda], lr 0.3),
 * lambda (8 per row, start 1), on mask_mean(relu(thres - score(optim)), valid) with
 *   optim = softmax(lambda)_0 * controls + sum_i softmax(lambda)_i * list[list_idx[i-1]],  i = 1..7
 * (list = the rollout's normalised list (n_list,N,40), list_idx = the reference's {0,50,80,85,90,95,98}: a HOST array of 7);
 * out_controls = the optim of the last forward pass, or `controls` for the other rows.  A first launch scores `controls`,
 * copies the other rows through and packs each scene's rows to mix; all iterations of a row then run in ONE launch.  grad_scale = (1/clip(mean(valid),1e-2))/N_global; adam_* as for pstl_trajopt (DEVICE arrays [iters]);
 * work: pstl_refinement_work_floats(cfg) floats; grad_trace (iters,N,8) or null: d loss / d lambda per iteration (tests). */
size_t pstl_refinement_work_floats(const pstl_cfg* cfg);
int pstl_refinement(const pstl_cfg* cfg, const float* s0, const float* nei_prep, const float* lane_prep, const float* stlp,
                    const float* hl, const float* valid, float thres, float grad_scale, int iters,
                    const float* adam_neg_step, const float* adam_bc2_sqrt, const float* controls, const float* list,
                    int n_list, const int32_t* list_idx, float* work, float* out_controls, float* grad_trace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSTL_HIP_H */
