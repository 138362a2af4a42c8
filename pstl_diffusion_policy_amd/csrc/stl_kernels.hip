// stl_kernels.hip -- STL robustness kernels for gfx950 (compile with -ffp-contract=off, see stl_core.hpp).
//
// One sampled trajectory per lane, one wavefront (64 rows) per workgroup.  A lane's 20 states (and, for the
// adjoint, 20 stored suffix log-sum-exps) live in LDS, lane-interleaved (element i of lane l at lds[i*64 + l], so
// every LDS access of a wave is 64 consecutive dwords: conflict-free).  Scene data (prepared neighbour circles,
// lane waypoints) is read straight from global memory: all rows of a scene are contiguous (192 rows = 3 waves at
// S = 64), so a wave's lanes read the same addresses and the loads are served from L1/L2 as broadcasts.
// The work is VALU/transcendental bound (~10^3 exp/log/sqrt and ~2*10^4 flops per row evaluation against 160 B of
// per-row input), nowhere near the HBM roofline; see DESIGN.md.
#include <stdlib.h>

#include "pstl_common.hpp"
#include "rng.hpp"
#include "stl_core.hpp"

namespace pstl {
namespace {

constexpr int kWave = 64;

struct StlArgs {
  long N;               // rows per rep
  int rows_per_scene;
  int K;
  int reps;
  StlEnv env;
  const float* s0;        // (bs,4)
  const float* controls;  // (reps,N,40)
  const float* states;    // (reps,N,T,4) or null: score given trajectories instead of rolling out controls
  const float* nei_prep;  // (bs,K,T,12)
  const float* lane_prep; // (bs,3,15,4)
  const float* stlp;      // (N,6)
  const float* hl;        // (N,)
  float* scores;          // (reps,N)
  float* scores3;         // (3,reps,N) or null
  float* sel_controls;    // (N,40) or null
  float* sel_scores;
  int32_t* sel_idx;
  int rep_split;          // small batches: blockIdx.y is the rep (one candidate per wavefront, k_stl_select picks afterwards)
  int by_mode;            // a wavefront takes 64 samples of ONE (scene, mode) (map_row): the formula branches are wave-uniform
};

template <bool NORM = false>
__device__ __forceinline__ StlRow load_row(const float* stlp, const float* hl, long row) {
  StlRow r;
  const float* p = stlp + row * 6;
  r.vmin = p[0];
  r.vmax = p[1];
  r.dmin = p[2];
  r.dmax = p[3];
  r.dsafe = p[4];
  r.thmax = p[5];
  const float h = hl[row];
  r.mode = (h == 0.0f) ? 0 : (h == 1.0f) ? 1 : (h == 2.0f) ? 2 : 3;  // anything else scores the outlier constant
  r.vf = r.df = r.sf = 1.0f;
  if (NORM) norm_factors(r);   // --norm_stl (PSTL_FLAG_NORM_STL)
  return r;
}

// Scene tables of the row.  STAGED (host picks it when rows_per_scene % 64 == 0, so that the 64 rows of a workgroup
// share one scene): the wave copies the scene's lane waypoints and prepared neighbour circles into LDS once, and every
// later read is an LDS broadcast instead of a latency-bound global load.  Otherwise they are read from global memory.
// Dynamic LDS layout: [scratch: n_scratch x 64 floats][lanes: 3*15 float4][neighbours: K*20*12 floats].
// The workgroup this block stands for.  With by_mode (a wavefront = 64 samples of one (scene, mode)) the gps = rows_per_scene / 64
// blocks of a scene read interleaved rows -- row = (b S + s) 3 + mode: the three modes of a sample are 160 bytes apart, so the
// same 128-byte lines serve three wavefronts -- and the dispatcher deals consecutive blocks round-robin over the 8 XCDs, each
// with an L2 of its own: every line was fetched by up to three L2s (943 B per row-evaluation of k_guidance_iter against 425 B
// algorithmic, profiles/r5/pmc_summary.json).  So within every run of 8 gps blocks, block 8 j + x (XCD x, for speed only: the
// placement is observed, not promised) stands for block x gps + j: a scene's blocks share an XCD and arrive there back to back.
__device__ __forceinline__ long virt_block(int by_mode, int rows_per_scene) {
  const long x = blockIdx.x;
#ifdef PSTL_NO_XCD_MAP      // (A/B builds: tools/dbg/build_full_variant.sh noxcd -DPSTL_NO_XCD_MAP)
  return x;
#endif
  if (!by_mode) return x;
  const long gps = rows_per_scene / kWave, G = 8 * gps;
  if (x >= ((long)gridDim.x / G) * G) return x;       // the last, partial run keeps its order
  const long r = x % G;
  return x - r + (r % 8) * gps + r / 8;
}

template <bool STAGED>
__device__ __forceinline__ void scene_tables(float* lds, int n_scratch, const float* lane_prep, const float* nei_prep,
                                             int K, int rows_per_scene, long row, const f4*& lanes, const float*& nei,
                                             int by_mode = 0) {
  if (STAGED) {
    const long b = (virt_block(by_mode, rows_per_scene) * kWave) / rows_per_scene;  // uniform over the workgroup
    f4* sl = reinterpret_cast<f4*>(lds + n_scratch * kWave);
    f4* sn = sl + 3 * kNseg + 3;  // keep 16-byte alignment and a little padding
    const f4* gl = reinterpret_cast<const f4*>(lane_prep) + b * 3 * kNseg;
    const f4* gn = reinterpret_cast<const f4*>(nei_prep + b * (long)K * kT * kNeiPrep);
    for (int i = threadIdx.x; i < 3 * kNseg; i += blockDim.x) sl[i] = gl[i];
    for (int i = threadIdx.x; i < K * kT * 3; i += blockDim.x) sn[i] = gn[i];
    __syncthreads();
    lanes = sl;
    nei = reinterpret_cast<const float*>(sn);
  } else {
    const long b = row / rows_per_scene;
    lanes = reinterpret_cast<const f4*>(lane_prep) + b * 3 * kNseg;
    nei = nei_prep + b * (long)K * kT * kNeiPrep;
  }
}

__host__ __device__ inline int stl_table_floats(int K) { return (3 * kNseg + 3) * 4 + K * kT * kNeiPrep; }   // staged lanes + neighbours
// (timing builds only, -DPSTL_DBG_LDS_ENV: extra dynamic LDS from the environment = fewer resident wavefronts, for occupancy
// sweeps without rebuilding -- tools/dbg/occupancy_sweep.sh; every other build: nothing)
static inline size_t dbg_lds_pad(const char* name) {
#ifdef PSTL_DBG_LDS_ENV
  const char* v = getenv(name);
  return v ? (size_t)atol(v) : 0;
#else
  (void)name;
  return 0;
#endif
}
inline size_t stl_lds_bytes(int n_scratch, int K, bool staged) {
  return ((size_t)n_scratch * kWave + (staged ? (size_t)(3 * kNseg + 3) * 4 + (size_t)K * kT * kNeiPrep : 0)) * sizeof(float);
}

// Row handled by this lane.  by_mode (scene-indexed rows r = (b*S + s)*3 + mode with S a multiple of 64): a wavefront
// takes 64 samples of ONE (scene, mode) instead of 64 consecutive rows.  Rows of a (scene, mode) share their lane, their
// validity and mostly their fate (satisfied or not), so whole wavefronts take the cheap exits of stl_eval_grad -- an
// invalid lane skips both sweeps, a satisfied row the adjoint -- instead of idling beside the lanes that cannot.
// The workgroup -> scene map is that of the block this one stands for (virt_block(...) * 64 / rows_per_scene): scene_tables and
// the per-scene lists of --refinement (k_mix_select, k_mixopt) take the scene from there, never from blockIdx.x itself.
__device__ __forceinline__ long map_row(int by_mode, int rows_per_scene, int lane = -1) {
  const long blk = virt_block(by_mode, rows_per_scene);
  if (lane < 0) lane = threadIdx.x;
  if (!by_mode) return blk * kWave + lane;
  const int gps = rows_per_scene / kWave;   // workgroups per scene = 3 * (S / 64)
  const long b = blk / gps;
  const int g = (int)(blk % gps);
  const int mode = g % 3, chunk = g / 3, S = rows_per_scene / 3;
  return (b * S + chunk * kWave + lane) * 3 + mode;
}
static bool rows_by_mode(const pstl_cfg* cfg, bool staged) {
  return staged && cfg->rows_per_scene == 3 * cfg->S && cfg->S % kWave == 0;
}

// SPLIT: the latency layout of k_guidance_iter for scoring (small batches, the selected formula, controls as input): ten
// wavefronts compute the geometry of two time steps each into LDS, waves 0-3 walk one group of the forward sweep's running
// log-sum-exps each (stl_pre_chain), wave 0 combines them into the score.  Bit-identical scores.
constexpr int kSplitWaves = 10;             // two time steps per wave

// (five wavefronts per SIMD is what the selected-formula kernel's 7.8 KB of LDS per wavefront allow at K = 2: its registers are
// held to that -- 96 --, or the wider packed clearance loop costs it a wavefront)
template <bool ALL3, bool STAGED, bool GIVEN, bool NORM = false, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? kSplitWaves * kWave : kWave) __attribute__((amdgpu_waves_per_eu(ALL3 ? 4 : 5)))
void k_stl_forward(StlArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(!SPLIT || (STAGED && !ALL3 && !GIVEN), "latency layout: staged tables, selected formula, controls");
  if (!SPLIT) PSTL_ST_BEGIN();
  constexpr int NS = ALL3 ? kScratchFwd3 : kScratchFwd;
  const int lane = SPLIT ? (int)(threadIdx.x & (kWave - 1)) : (int)threadIdx.x;
  const int wq = SPLIT ? (int)(threadIdx.x / kWave) : 0;
  long row = SPLIT ? (long)blockIdx.x * kWave + lane : map_row(a.by_mode, a.rows_per_scene, lane);
  const f4* lanes;
  const float* nei;
  scene_tables<STAGED>(lds, NS, a.lane_prep, a.nei_prep, a.K, a.rows_per_scene, row < a.N ? row : a.N - 1, lanes, nei,
                       SPLIT ? 0 : a.by_mode);   // (the latency layout takes 64 consecutive rows: its blocks keep their order)
  const bool live = row < a.N;
  if (!SPLIT && !live) return;
  if (!live) row = a.N - 1;
  const Scratch st = {lds + lane, kWave};
  const long b = row / a.rows_per_scene;
  const StlRow r = load_row<NORM>(a.stlp, a.hl, row);
  if (!SPLIT) PSTL_ST(ST_PROLOGUE);
  float best = -INFINITY;
  int best_rep = 0;
  const int rep_lo = a.rep_split ? (int)blockIdx.y : 0, rep_hi = a.rep_split ? rep_lo + 1 : a.reps;
  for (int rep = rep_lo; rep < rep_hi; ++rep) {
    float o3[3];
    float score;
    if constexpr (SPLIT) {
      float* geo = lds + NS * kWave + stl_table_floats(a.K);
      const DynSrc src(a.s0 + b * 4, a.controls + ((long)rep * a.N + row) * (2 * kT), 1.0f, 1.0f, a.env.dt);
      if (rep > rep_lo) __syncthreads();   // the chains have read the previous candidate's geometry, wave 0 their values
      if (live && r.mode < 3)
        stl_geometry<false>(a.env, lanes + r.mode * kNseg, nei, a.K, src, (kT / kSplitWaves) * wq, (kT / kSplitWaves) * (wq + 1),
                            geo + lane, kWave);
      __syncthreads();
      // the forward sweep's chains on waves 0-3, one group each, then wave 0 combines them (stl_core.hpp, stl_pre_chain: as in
      // the guidance kernel's latency layout; their values travel through slots 4-9 of step 0, which value-only geometry leaves alone)
      float* g0 = geo + lane;
      if (wq < 4 && live && r.mode < 3) {
        const ChainOut co = stl_pre_chain<NORM, false>(wq, a.env, r, GeoPre{g0, kWave}, st);
        if (wq == 0) g0[4 * kWave] = co.o0, g0[5 * kWave] = co.o1;
        else if (wq == 1) g0[6 * kWave] = co.o0;
        else if (wq == 2) g0[7 * kWave] = co.o0, g0[8 * kWave] = co.o1;
        else g0[9 * kWave] = co.o0;
      }
      __syncthreads();
      if (wq != 0 || !live) continue;
      o3[0] = o3[1] = o3[2] = 0.0f;
      score = 1.0f;   // (an outlier mode scores the constant 1: nusc_train.py:150-151)
      if (r.mode < 3) {
        AdjCtx C;
        C.Lv1 = g0[4 * kWave], C.Lv2 = g0[5 * kWave], C.Ls = g0[6 * kWave], C.L1 = g0[7 * kWave], C.L2 = g0[8 * kWave], C.L3 = g0[9 * kWave];
        float unused;
        score = adj_pre_weights(a.env, r.mode, C, [](float) { return 0.0f; }, unused);
      }
    } else
    if (GIVEN) {
      const GivenSrc src = {reinterpret_cast<const f4*>(a.states) + ((long)rep * a.N + row) * kT};
      score = stl_eval<ALL3, -1, NORM>(a.env, r, lanes, nei, a.K, src, st, 0, o3, nullptr);
    } else {
      const DynSrc src(a.s0 + b * 4, a.controls + ((long)rep * a.N + row) * (2 * kT), 1.0f, 1.0f, a.env.dt);
      score = stl_eval<ALL3, -1, NORM>(a.env, r, lanes, nei, a.K, src, st, 0, o3, nullptr);
    }
    a.scores[(long)rep * a.N + row] = score;
    if (ALL3 && a.scores3) {
      const long stride = (long)a.reps * a.N;
      a.scores3[(long)rep * a.N + row] = o3[0];
      a.scores3[stride + (long)rep * a.N + row] = o3[1];
      a.scores3[2 * stride + (long)rep * a.N + row] = o3[2];
    }
    if (score > best || rep == rep_lo) {  // first maximum wins, like torch.max(dim=0)
      best = score;
      best_rep = rep;
    }
  }
  if (SPLIT && (wq != 0 || !live)) return;
  if (a.sel_controls && a.controls) {
    const f4* src = reinterpret_cast<const f4*>(a.controls + ((long)best_rep * a.N + row) * (2 * kT));
    f4* dst = reinterpret_cast<f4*>(a.sel_controls + row * (2 * kT));
#pragma unroll
    for (int i = 0; i < 10; ++i) dst[i] = src[i];
    a.sel_scores[row] = best;
    a.sel_idx[row] = best_rep;
  }
  if (!SPLIT) {
    PSTL_ST(ST_SELECT);
    PSTL_ST_END();
  }
}

// Candidate selection as a pass of its own (small batches, where k_stl_forward spreads the candidates over workgroups to cut
// the latency of the launch): first maximum over the reps, like torch.max(dim=0); the chosen controls are gathered.
__global__ void k_stl_select(long N, int reps, const float* scores, const float* controls, float* sel_controls,
                             float* sel_scores, int32_t* sel_idx) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  float best = scores[row];
  int best_rep = 0;
  for (int rep = 1; rep < reps; ++rep) {
    const float sc = scores[(long)rep * N + row];
    if (sc > best) best = sc, best_rep = rep;
  }
  const f4* src = reinterpret_cast<const f4*>(controls + ((long)best_rep * N + row) * (2 * kT));
  f4* dst = reinterpret_cast<f4*>(sel_controls + row * (2 * kT));
#pragma unroll
  for (int i = 0; i < 10; ++i) dst[i] = src[i];
  sel_scores[row] = best;
  sel_idx[row] = best_rep;
}

// The closed-loop caller's choice of the control to apply (reference nusc_sim.py:677-683: scores of modes 1, 2 set to -10000,
// torch.argmax over the flattened (S,3) scores of ONE scene -- first maximum wins --, then that row's control sequence): the best
// lane-keeping sample.  out[0..1] = its first (w, a), out[2] = its score, out[3] = the bit pattern of *status (the packed
// weight buffer's chain-domain word, so that the flag reaches the host in the same 16-byte copy).  One wavefront.
__global__ __launch_bounds__(kWave) void k_select_plan(int S, const float* scores /* (S,3) */, const float* controls /* (S*3,40) */,
                                                       const unsigned* status, float* out) {
  const int lane = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int s = lane; s < S; s += kWave) {
    const float v = scores[3 * s];
    if (bi == 0x7fffffff || v > best) best = v, bi = s;   // (ascending s within a lane: the first maximum stays)
  }
  // all -10000 columns lose to any finite mode-0 score; a NaN score never wins a ">" (torch.argmax would return it: the
  // domain flag in out[3] is what reports such a batch)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ob = __shfl_xor(best, m, kWave);
    const int oi = __shfl_xor(bi, m, kWave);
    if (ob > best || (ob == best && oi < bi)) best = ob, bi = oi;
  }
  if (lane == 0) {
    const float* c = controls + (long)(3 * bi) * (2 * kT);
    out[0] = c[0];
    out[1] = c[1];
    out[2] = best;
    out[3] = status ? __builtin_bit_cast(float, *status) : 0.0f;
  }
}


struct GradArgs {
  long N;
  int rows_per_scene;
  int K;
  int by_mode;
  StlEnv env;
  float wscale, ascale;
  const float* s0;
  const float* u;  // (N,40)
  const float* nei_prep;
  const float* lane_prep;
  const float* stlp;
  const float* hl;
  const float* dscore;  // (N,) or null
  float* dcontrols;     // (N,40)
  float* scores;        // (N,) or null
};

template <bool STAGED, bool NORM = false>
__global__ __launch_bounds__(kWave) void k_stl_backward(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const long row = map_row(a.by_mode, a.rows_per_scene);
  const f4* lanes;
  const float* nei;
  scene_tables<STAGED>(lds, kScratchGrad, a.lane_prep, a.nei_prep, a.K, a.rows_per_scene, row < a.N ? row : a.N - 1, lanes,
                       nei, a.by_mode);
  if (row >= a.N) return;
  const Scratch st = {lds + threadIdx.x, kWave};
  const long b = row / a.rows_per_scene;
  const StlRow r = load_row<NORM>(a.stlp, a.hl, row);
  const float ds = a.dscore ? a.dscore[row] : 1.0f;
  float* out = a.dcontrols + row * (2 * kT);
  const float score = stl_eval_grad<NORM>(
      a.env, r, lanes, nei, a.K, a.s0 + b * 4, a.u + row * (2 * kT), st, a.wscale, a.ascale, [=](float) { return ds; },
      [=](int t, float gw, float ga, float, float) {
        store_pair(out + 2 * t, gw, ga);   // one 8-byte gather store per time step
      },
      1, ds == 0.0f && !a.scores);
  if (a.scores) a.scores[row] = score;
}

// ---- guidance: one Adam iteration on mu, optionally finishing the reverse step (x = mu + sqrt(beta) z) ------------
struct GuideArgs {
  long N;
  int rows_per_scene;
  int K;
  int by_mode;
  StlEnv env;
  float wscale, ascale;    // mul_w_max, mul_a_max
  float thres;             // stl_nn_thres, or 100 with PSTL_FLAG_MAXIMIZE
  float grad_scale;        // (1/clip(mean(valid),1e-2))/N
  float neg_step, bc2_sqrt;  // Adam scalars of this iteration
  float beta_i, sqrt_beta;
  int iter, niters;
  int clip;
  const float* s0;
  const float* nei_prep;
  const float* lane_prep;
  const float* stlp;
  const float* hl;
  const float* valid;
  const float* z;   // (N,40) or null
  int rng, step;    // PSTL_FLAG_RNG: draw z for reverse step `step` in the kernel
  unsigned long long seed;
  const pstl_dyn* dyn;   // cfg->dyn: seed and grad_scale are read from device memory instead (HIP-graph replay)
  long row_offset;
  float* mu;        // (N,40) in/out
  float* work;      // (3,N,40): m, v, anchor (niters > 1 only)
  float* emit_out;  // (N,40) or null
};

// SPLIT (latency layout, small batches -- the closed-loop caller's 192 rows are three wavefronts on a 256-CU chip): the
// workgroup is kSplitWaves = 10 wavefronts over the same 64 rows.  Wave q first computes the geometry of time steps [2q, 2q + 2) of its
// lane's row -- clearance, lane distance, heading term, the winners -- into LDS (stl_geometry: the forward sweep's own calls
// on the same states; the dynamics steps before 2q are regenerated).  The sweeps then read where the one-wave kernel computes
// and are themselves shared out (parts 0-4 below: the forward sweep's chains on four waves, the adjoint's direct partials and
// the update two steps per wave, only the score and the costate recursion on wave 0).  Same operations on the same operands,
// every chain in its own order: bit-identical results.  Per launch at K = 8: 96 us (one wave) -> 38 (geometry shared, round 3)
// -> 25 (sweeps shared, round 4).
constexpr int kGeoFloats = kGeoSlots * kT;   // per lane

// (SPLIT: two ten-wave workgroups per CU -- 74 KB of LDS each at K = 2 -- need five wavefronts per SIMD: <= 96 registers)
// -DPSTL_G_ABL=1 / 2: timing-only builds that bound what another layout of the state could gain (tools/dbg/guidance_layout_bound.sh)
#ifndef PSTL_G_ABL
#define PSTL_G_ABL 0
#endif
template <bool MULTI, bool STAGED, bool NORM = false, bool SPLIT = false>
// (registers: the staged one-wave forms are held to the 168 of three wavefronts per SIMD -- what their 12.9 KB of LDS allow anyway --,
// the ten-wave form to the 96 of its two workgroups per CU)
__global__ __launch_bounds__(SPLIT ? kSplitWaves * kWave : kWave)
__attribute__((amdgpu_waves_per_eu(SPLIT ? 5 : (STAGED ? 3 : 2)))) void k_guidance_iter(GuideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(!SPLIT || STAGED, "the latency layout stages its scene tables");
  if (!SPLIT) PSTL_ST_BEGIN();
  const unsigned long long seed = a.dyn ? uniform_u64(&a.dyn->seed) : a.seed;
  const float grad_scale = a.dyn ? uniform_f32(&a.dyn->grad_scale) : a.grad_scale;
  const int lane = SPLIT ? (int)(threadIdx.x & (kWave - 1)) : (int)threadIdx.x;
  const int wq = SPLIT ? (int)(threadIdx.x / kWave) : 0;
  long row = map_row(a.by_mode, a.rows_per_scene, lane);
  const f4* lanes;
  const float* nei;
  constexpr int NS = SPLIT ? kScratchGradPre : kScratchGrad;
  scene_tables<STAGED>(lds, NS, a.lane_prep, a.nei_prep, a.K, a.rows_per_scene, row < a.N ? row : a.N - 1, lanes, nei, a.by_mode);
  GeoPre pre = {nullptr, 0};
  if (SPLIT) {
    const bool live0 = row < a.N;
    if (!live0) row = a.N - 1;
    float* geo = lds + NS * kWave + stl_table_floats(a.K);
    const StlRow rq = load_row<NORM>(a.stlp, a.hl, row);
#ifndef PSTL_DBG_GEXIT
#define PSTL_DBG_GEXIT 0   // (timing builds: leave before / after the geometry / part 0 / 1 / 2 / 3)
#endif
    if (PSTL_DBG_GEXIT == 9) return;
    if (live0 && rq.mode < 3 && grad_scale * a.valid[row] != 0.0f)
      stl_geometry(a.env, lanes + rq.mode * kNseg, nei, a.K,
                   DynSrc(a.s0 + (row / a.rows_per_scene) * 4, a.mu + row * (2 * kT), a.wscale, a.ascale, a.env.dt),
                   (kT / kSplitWaves) * wq, (kT / kSplitWaves) * (wq + 1), geo + lane, kWave);
    __syncthreads();
    pre = GeoPre{geo + lane, kWave};
  } else if (row >= a.N) {
    return;
  }
  const bool live = !SPLIT || map_row(a.by_mode, a.rows_per_scene, lane) < a.N;   // (SPLIT: dead rows stay for the barriers)
  const Scratch st = {lds + lane, kWave};
  const long b = row / a.rows_per_scene;
  const StlRow r = load_row<NORM>(a.stlp, a.hl, row);
  float* mu = a.mu + row * (2 * kT);
  const float vr = a.valid[row];
  const float gs = grad_scale * vr;
  const float thres = a.thres;
  const bool last = MULTI ? (a.iter == a.niters - 1) : true;   // (one iteration per guided step -- the !MULTI build -- is always the last)
  const long plane = a.N * (2 * kT);
  float* wm = MULTI ? a.work + row * (2 * kT) : nullptr;
  const float* zr = a.z ? a.z + row * (2 * kT) : nullptr;
  float* er = a.emit_out ? a.emit_out + row * (2 * kT) : nullptr;
  const float inv_bc2 = 1.0f / a.bc2_sqrt;
  if (!SPLIT) PSTL_ST(ST_PROLOGUE);
  // one element of the Adam update; returns the value that goes back to mu (and, through *emit_v, to emit_out)
  auto update = [=](int e, float p0, float g, float nscale, float zdrawn, float* emit_v) -> float {
    // torch.optim.Adam, single-tensor path, betas (0.9, 0.999), eps 1e-8 (see oracle guidance_update)
    float m = 0.0f, v = 0.0f;
    if (MULTI && a.iter > 0) {
      m = wm[e];
      v = wm[plane + e];
    }
    float p;
    if (!MULTI && __builtin_constant_p(g) && g == 0.0f) {
      // A row without gradient (stl_grad_zero hands in the literal 0): with fresh moments m = v = 0 and the step is
      // p0 + (neg_step * 0) / 1e-8 = p0 + (-0) = p0, bit for bit -- no arithmetic at all for the rows whose hinge is inactive
      // (84 % of them by the last guided step of the bench workload) instead of ~22 instructions per element.
      p = p0;
    } else {
    m = m + 0.1f * (g - m);
    v = v * 0.999f + (0.001f * g) * g;
    // sqrt and both divisions in their hardware forms (1 ulp each: the step lr m / (sqrt(v) + eps) is at most lr, so the state
    // moves by < 1e-8 against the IEEE forms; v_sqrt_f32 takes a denormal v as 0, where eps = 1e-8 decides anyway)
    const float denom = PSTL_SQRT_ADJ(v) * inv_bc2 + 1e-8f;
    p = p0 + (a.neg_step * m) * PSTL_RCP_ADJ(denom);
    }
    if (MULTI) {
      wm[e] = m;
      wm[plane + e] = v;
      if (a.iter == 0) {
        wm[2 * plane + e] = p;  // anchor = the value after the first Adam step
      } else {
        const float an = wm[2 * plane + e];
        p = an + fminf(fabsf(p - an), a.beta_i);
      }
    }
    if (!last) return p;
    // (zdrawn: this element of the step's noise quad -- drawn in the kernel, read from the caller's tensor, or zero: noise_quad)
    const float x = p + a.sqrt_beta * zdrawn;
    float c = x * nscale;
    if (a.clip) c = c < -nscale ? -nscale : (c > nscale ? nscale : c);   // torch.clip: a NaN stays a NaN
    *emit_v = c;
    return x;
  };
  // The noise of time steps 2 q, 2 q + 1 (elements 4 q .. 4 q + 3 of the row): drawn here (PSTL_FLAG_RNG: one Philox block), one
  // 16-byte read of the caller's tensor (the parity tests), or zeros (the last reverse step).  Round 6: until now every ELEMENT
  // chose its source by itself -- a scalar branch or two and, in tensor mode, a 4-byte gather with a full wait, forty times a row.
  auto noise_quad = [=](int q) -> f4 {
    f4 z4 = f4{0.0f, 0.0f, 0.0f, 0.0f};
    if (!last) return z4;
    if (a.rng) {
      if (a.step > 1) {
        float zz[4];
        normal4(seed, a.row_offset + row, q, a.step, zz);
        z4 = f4{zz[0], zz[1], zz[2], zz[3]};
      }
    } else if (zr) {
      z4 = *reinterpret_cast<const f4*>(zr + 4 * q);
    }
    return z4;
  };
  // the update of time step t: Adam on (w, a), and on the last iteration the noise (quad z4 of steps t | 1 and t & ~1) and emission
  // (Both callers walk t = T-1 ... 0, an odd step right before its even neighbour: the odd step's results wait in `held` and
  // leave with the even step's as ONE 16-byte store per tensor -- half the store instructions of the 8-byte pairs this kernel
  // used to issue, each of which touches 64 lines of the row-major state: a wavefront's rows are 480 bytes apart.  Nothing
  // reads mu[2t .. 2t+3] between the two steps: the adjoint only looks at earlier time steps.)
  auto apply = [=](int t, float gw, float ga, float w0, float a0, const f4& z4, f4& held_mu, f4& held_em) {
    const int o = (t & 1) * 2;
    float ew = 0.0f, ea = 0.0f;
    const float nw = update(2 * t, w0, gw, a.wscale, o ? z4.z : z4.x, &ew);
    const float na = update(2 * t + 1, a0, ga, a.ascale, o ? z4.w : z4.y, &ea);
    if (t & 1) {
      held_mu.x = nw, held_mu.y = na, held_em.x = ew, held_em.y = ea;
    } else {
#if PSTL_G_ABL == 1      // timing only: no state / candidate stores at all (what ANY store layout could save at most)
      if (nw == 12345.678f) mu[0] = ew + ea + held_em.x + held_mu.x;
#elif PSTL_G_ABL == 2    // timing only: element-major addresses (a wavefront's lanes 12 bytes apart instead of 480)
      float* me = a.mu + row + (long)(2 * t) * a.N;
      me[0] = nw, me[a.N] = na, me[2 * a.N] = held_mu.x, me[3 * a.N] = held_mu.y;
      if (last && er) {
        float* ee = a.emit_out + row + (long)(2 * t) * a.N;
        ee[0] = ew, ee[a.N] = ea, ee[2 * a.N] = held_em.x, ee[3 * a.N] = held_em.y;
      }
#else
      store_quad(mu + 2 * t, nw, na, held_mu.x, held_mu.y);
      if (last && er) store_quad(er + 2 * t, ew, ea, held_em.x, held_em.y);
#endif
    }
  };
  if constexpr (SPLIT) {
    // Wave 0 leaves every step's gradient and stored controls in LDS; after a barrier each wave updates its own two steps
    // (= one noise quad): the elements are independent, so the ten waves share the Adam / Philox / emission work as well.
    static_assert(kT == 2 * kSplitWaves, "a wave's two time steps are one noise quad");
    // After the geometry both sweeps run in parts (stl_core.hpp, stl_pre_chain / adj_pre_*), a barrier between them:
    //   (0) waves 0-3 each walk one group of the forward sweep's running log-sum-exps over the 20 steps (speed | clearance |
    //       lane distance or band | heading), the lane changes' suffix tables included;
    //   (1) wave 0 combines them into the score, the hinge and the weights of the formula's terms;
    //   (2) every wave computes the direct partials of its own two time steps (~110 instructions per step that used to sit on
    //       wave 0's serial path);
    //   (3) wave 0 runs the costate recursion (~20 per step) and leaves every step's gradient in LDS;
    //   (4) every wave updates its own two steps (= one noise quad; the stored controls it reads itself).
    // Measured per launch at 192 rows, K = 2, before the split: geometry 14 us, forward sweep 11, adjoint 10, update 2.
    // No buffer of its own for any of it (the workgroup stays under 80 KB and two fit a CU) -- all in slots nobody reads any more:
    //   the chains' values (Lv1, Lv2, Ls, L1 | Lfb, L2, L3 | Lft): slots 4-9 of step 0 (adjoint slots; step 0 has no adjoint);
    //   the six weights: slots 0-3, 10, 11 of step 0 (the chains have read step 0's forward values); slot 12: 1 = the row
    //   has no gradient (satisfied, invalid or an outlier mode);
    //   direct partials (gx, gy, gth, gv) of step t: slots 4-7 of step t, over the clearance partials they were made from;
    //   gradient (gw, ga) of step t: slots 0, 1 of step t (forward values part 2 has consumed).
    float* geo_l = lds + NS * kWave + stl_table_floats(a.K) + lane;
    auto slot = [=](int t, int c) -> float* { return geo_l + (kGeoSlots * t + c) * kWave; };
    if (PSTL_DBG_GEXIT == 1) return;
    const bool act = live && r.mode < 3 && gs != 0.0f;   // (an invalid lane has zero loss weight: neither sweep is needed)
    if (wq < 4 && act) {
      const ChainOut co = stl_pre_chain<NORM>(wq, a.env, r, pre, st);
      if (wq == 0) *slot(0, 4) = co.o0, *slot(0, 5) = co.o1;
      else if (wq == 1) *slot(0, 6) = co.o0;
      else if (wq == 2) *slot(0, 7) = co.o0, *slot(0, 8) = co.o1;
      else *slot(0, 9) = co.o0;
    }
    __syncthreads();
    if (PSTL_DBG_GEXIT == 2) return;
    auto load_ctx = [=](AdjCtx& C) {
      C.Lv1 = *slot(0, 4), C.Lv2 = *slot(0, 5), C.Ls = *slot(0, 6), C.L1 = *slot(0, 7), C.L2 = *slot(0, 8), C.L3 = *slot(0, 9);
    };
    bool dead = true;
    if (wq == 0) {
      if (act) {
        AdjCtx C;
        load_ctx(C);
        float dsc;
        adj_pre_weights(a.env, r.mode, C, [=](float score) { return (thres - score > 0.0f) ? -gs : 0.0f; }, dsc);
        if (dsc != 0.0f) {
          *slot(0, 0) = C.om[0], *slot(0, 1) = C.om[1], *slot(0, 2) = C.om[2], *slot(0, 3) = C.om[3], *slot(0, 10) = C.om[4],
          *slot(0, 11) = C.om[5];
          dead = false;
        }
      }
      *slot(0, 12) = dead ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (PSTL_DBG_GEXIT == 3) return;
    const bool has_grad = live && *slot(0, 12) == 0.0f;
    if (has_grad) {
      AdjCtx C;
      load_ctx(C);
      C.om[0] = *slot(0, 0), C.om[1] = *slot(0, 1), C.om[2] = *slot(0, 2), C.om[3] = *slot(0, 3), C.om[4] = *slot(0, 10),
      C.om[5] = *slot(0, 11);
      PSTL_NOUNROLL
      for (int t = 2 * wq + 1; t >= 2 * wq && t >= 1; --t) {
        float gx, gy, gth, gv;
        adj_pre_direct<NORM>(a.env, r, C, pre, st, t, gx, gy, gth, gv);
        *slot(t, 4) = gx, *slot(t, 5) = gy, *slot(t, 6) = gth, *slot(t, 7) = gv;
      }
    }
    __syncthreads();
    if (PSTL_DBG_GEXIT == 4) return;
    if (wq == 0 && has_grad)
      adj_pre_costate(
          a.env, pre, a.wscale, a.ascale,
          [=](int t, float& gx, float& gy, float& gth, float& gv) { gx = *slot(t, 4), gy = *slot(t, 5), gth = *slot(t, 6), gv = *slot(t, 7); },
          [=](int t, float gw, float ga) { *slot(t, 0) = gw, *slot(t, 1) = ga; });
    __syncthreads();
    if (PSTL_DBG_GEXIT == 5) return;
    if (!live) return;
    const f4 z4 = noise_quad(wq);
    f4 held_mu = f4{0.0f, 0.0f, 0.0f, 0.0f}, held_em = held_mu;
    for (int t = 2 * wq + 1; t >= 2 * wq; --t) {
      float w0, a0;
      ctrl_pair(mu, 1, t, w0, a0);   // (the stored controls of this wave's own steps; nobody else reads or rewrites them)
      apply(t, has_grad ? *slot(t, 0) : 0.0f, has_grad ? *slot(t, 1) : 0.0f, w0, a0, z4, held_mu, held_em);
    }
  } else {
    // mu[2t], mu[2t+1] are rewritten by emit(t) while the adjoint walks t = T-1 ... 0; the adjoint has already taken every
    // value it still needs from earlier time steps only (and hands the current one to emit, so mu is not read here)
    stl_eval_grad<NORM>(
        a.env, r, lanes, nei, a.K, a.s0 + b * 4, PSTL_G_ABL == 2 ? a.mu + row : mu, st, a.wscale, a.ascale,
        [=](float score) { return (thres - score > 0.0f) ? -gs : 0.0f; },
        [=, z4 = f4{0.0f, 0.0f, 0.0f, 0.0f}, held_mu = f4{0.0f, 0.0f, 0.0f, 0.0f}, held_em = f4{0.0f, 0.0f, 0.0f, 0.0f}](
            int t, float gw, float ga, float w0, float a0) mutable {
          // a noise quad covers two time steps (elements 4q .. 4q+3); emit() comes in the order t = T-1 ... 0, so the quad is
          // drawn at the odd step and kept for the even one: one Philox draw per two time steps
          if (t & 1) z4 = noise_quad(t >> 1);
          apply(t, gw, ga, w0, a0, z4, held_mu, held_em);
        },
        PSTL_G_ABL == 2 ? a.N : 1, gs == 0.0f);   // an invalid lane has zero loss weight: Adam sees exact zeros, only the noise is added
    PSTL_ST(ST_OTHER);
    PSTL_ST_END();
  }
}

// ---- trajectory optimisation (nusc_train.py:1302-1325 with compute_trajopt_loss_lite :287-316) --------
// params (N,40): controls in physical units.  Every row is independent given the batch-wide constants, so ONE launch
// runs all `iters` Adam iterations of a row back to back (forward sweep + adjoint + update), the scene tables staying
// in LDS: the reference launches ~10^4 kernels per iteration, 2000 iterations per batch.
//   loss = mean(relu(thres - score) * valid) / clip(mean(valid), 1e-3)
//        + reg * (mean(relu(w^2 - w_max^2)) + mean(relu(a^2 - a_max^2)))
struct TrajoptArgs {
  long N;
  int rows_per_scene;
  int K;
  int by_mode;
  StlEnv env;
  float thres, grad_scale;   // (1/clip(mean(valid),1e-3))/N
  float reg_scale;           // reg_loss / (N * nt): d reg / d relu-term
  float w_max2, a_max2;
  int iters;
  const float* neg_step;     // [iters] device: -lr / (1 - 0.9^k)
  const float* bc2_sqrt;     // [iters] device: sqrt(1 - 0.999^k)
  const float* s0;
  const float* nei_prep;
  const float* lane_prep;
  const float* stlp;
  const float* hl;
  const float* valid;
  float* params;             // (N,40) in/out
  float* work;               // (3,40,N) element-major: the iterate, Adam m, v (m, v read when resume != 0)
  float* scores;             // (N,) score of the iterate the LAST update started from (what the reference reports)
  int resume;
};

// amdgpu_waves_per_eu(3, 4): the register allocation is held to the 3 wavefronts per SIMD that the 10 KB of LDS per
// wavefront allows (the kernel is latency-bound; at 169 registers = 2 per SIMD it ran 10 % slower)
template <bool STAGED>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_trajopt(TrajoptArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const long row = map_row(a.by_mode, a.rows_per_scene);
  const long slot = (long)blockIdx.x * kWave + threadIdx.x;   // position in the element-major work buffer (lane-contiguous)
  const f4* lanes;
  const float* nei;
  scene_tables<STAGED>(lds, kScratchGrad, a.lane_prep, a.nei_prep, a.K, a.rows_per_scene, row < a.N ? row : a.N - 1, lanes,
                       nei, a.by_mode);
  if (row >= a.N) return;
  const Scratch st = {lds + threadIdx.x, kWave};
  const long b = row / a.rows_per_scene;
  const StlRow r = load_row(a.stlp, a.hl, row);
  // The iterate and Adam's moments live element-major in `work` ((3, 40, N): element e of row r at [e*N + r]) for the
  // whole run, so that the 64 lanes of a wavefront always touch 256 contiguous bytes: in the row-major layout every one
  // of the ~300 accesses per row and iteration was a 64-line gather that thrashed the 32 KB L1 (2.9 ms per iteration
  // over 786 432 rows at 12 resident wavefronts per CU; 1.7 ms at 7).
  const long N = a.N;
  const long plane = N * (2 * kT);
  float* u = a.work + slot;            // plane 0: the iterate
  float* wm = a.work + plane + slot;   // plane 1: m, plane 2: v
  {
    const float* src = a.params + row * (2 * kT);
    PSTL_NOUNROLL
    for (int e = 0; e < 2 * kT; ++e) u[e * N] = src[e];
  }
  const float gs = a.grad_scale * a.valid[row];
  const float thres = a.thres;
  float score = 0.0f;
  for (int it = 0; it < a.iters; ++it) {
    const float neg_step = a.neg_step[it], bc2 = a.bc2_sqrt[it];
    const bool fresh = (it == 0 && !a.resume);
    auto update = [=](int e, float p0, float g, float lim2) {
      // d reg / d p: relu'(p^2 - lim^2) * 2p * reg_scale   (pow backward: grad * (2 * p))
      if (p0 * p0 - lim2 > 0.0f) g = g + a.reg_scale * (2.0f * p0);
      float m = fresh ? 0.0f : wm[e * N], v = fresh ? 0.0f : wm[plane + e * N];
      m = m + 0.1f * (g - m);
      v = v * 0.999f + (0.001f * g) * g;
      const float denom = sqrtf(v) / bc2 + 1e-8f;
      u[e * N] = p0 + (neg_step * m) / denom;
      wm[e * N] = m;
      wm[plane + e * N] = v;
    };
    score = stl_eval_grad<false, true>(   // (WAVE_ZERO: this loop's emit() stores lane-contiguously)
        a.env, r, lanes, nei, a.K, a.s0 + b * 4, u, st, 1.0f, 1.0f,
        [=](float sc) { return (thres - sc > 0.0f) ? -gs : 0.0f; },
        [=](int t, float gw, float ga, float w0, float a0) {
          update(2 * t, w0, gw, a.w_max2);
          update(2 * t + 1, a0, ga, a.a_max2);
        },
        N);
  }
  {
    float* dst = a.params + row * (2 * kT);
    PSTL_NOUNROLL
    for (int e = 0; e < 2 * kT; ++e) dst[e] = u[e * N];
  }
  if (a.scores) a.scores[row] = score;
}

// ---- --refinement (nusc_train.py:1034-1071): 50 Adam iterations over per-row mixing weights, in ONE launch -----------
// For every row whose current controls score <= 0 (and whose lane is valid) the reference optimises lambda (8 values,
// start 1, Adam lr 0.3) so that   optim = sum_k softmax(lambda)_k * base_k,   base_0 = the current controls, base_1..7 =
// entries 0, 50, 80, 85, 90, 95, 98 of the rollout's list, minimises mask_mean(relu(5e-4 - score(optim)), valid); the
// result is the optim of the LAST forward pass.  Rows are independent, so one launch runs all iterations of a row back
// to back, exactly like k_trajopt: the bases, the iterate and its gradient live element-major in `work` (plane p of the
// lane's slot at work[p * n_slots + slot]: 64 lanes touch 256 contiguous bytes), the scene tables stay in LDS.
// Planes: [0,40) iterate | [40,80) d loss / d iterate | [80,400) the 8 bases | [400,424) lambda, Adam m, v.
constexpr int kMixK = 8, kMixPlanes = 2 * (2 * kT) + kMixK * (2 * kT) + 3 * kMixK;

struct MixArgs {
  long N;
  int rows_per_scene;
  int K;
  int by_mode;
  StlEnv env;
  float thres, grad_scale;
  int iters;
  const float* neg_step;     // [iters] device: -lr / (1 - 0.9^k)
  const float* bc2_sqrt;     // [iters] device: sqrt(1 - 0.999^k)
  const float* s0;
  const float* nei_prep;
  const float* lane_prep;
  const float* stlp;
  const float* hl;
  const float* valid;
  const float* base[kMixK];  // (N,40) each, physical units
  float* work;               // kMixPlanes * n_slots floats
  float* out;                // (N,40)
  float* grad_trace;         // (iters,N,8) d loss / d lambda of every iteration, or null (tests)
  int* mix_count;            // COMPACT: [scenes] rows to mix, and
  int* mix_list;             //          [scenes][rows_per_scene] their row indices (k_mix_select)
};

// Which rows --refinement mixes (score of the current controls <= 0 on a valid lane, nusc_train.py:1045-1046), packed per
// scene: the other rows copy their controls and are done.  Typically ~30 % of the rows are mixed; run where they sit,
// a wavefront of k_mixopt would carry ~19 live lanes through all 50 iterations -- packed, a scene's rows fill one dense
// wavefront instead of three sparse ones.  The order inside a scene's list depends on which wavefront reaches the counter
// first; results do not (a row's arithmetic knows nothing of its lane, outputs go to out[row]).
template <bool STAGED>
__global__ __launch_bounds__(kWave) void k_mix_select(MixArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const long row = map_row(a.by_mode, a.rows_per_scene);
  const f4* lanes;
  const float* nei;
  scene_tables<STAGED>(lds, kScratchFwd, a.lane_prep, a.nei_prep, a.K, a.rows_per_scene, row < a.N ? row : a.N - 1, lanes, nei, a.by_mode);
  constexpr int E = 2 * kT;
  bool mix = false;
  if (row < a.N) {
    const Scratch st = {lds + threadIdx.x, kWave};
    const long b = row / a.rows_per_scene;
    const StlRow r = load_row(a.stlp, a.hl, row);
    const float score0 = stl_eval<false, -1>(a.env, r, lanes, nei, a.K, DynSrc(a.s0 + b * 4, a.base[0] + row * E, 1.0f, 1.0f, a.env.dt),
                                             st, 0, nullptr, nullptr);
    mix = score0 <= 0.0f && a.valid[row] > 0.0f;
    if (!mix) {
      const f4* src = reinterpret_cast<const f4*>(a.base[0] + row * E);
      f4* dst = reinterpret_cast<f4*>(a.out + row * E);
      PSTL_UNROLL
      for (int q = 0; q < E / 4; ++q) dst[q] = src[q];
      if (a.grad_trace) {
        PSTL_NOUNROLL
        for (int it = 0; it < a.iters; ++it)
          PSTL_NOUNROLL
          for (int k = 0; k < kMixK; ++k) a.grad_trace[((long)it * a.N + row) * kMixK + k] = 0.0f;
      }
    }
  }
  const unsigned long long m = __ballot(mix);
  if (m == 0ull) return;
  // the scene of the block this one stands for (virt_block: the one its rows and its staged tables belong to), uniform over
  // the wavefront (rows_per_scene % 64 == 0)
  const long scene = (virt_block(a.by_mode, a.rows_per_scene) * kWave) / a.rows_per_scene;
  int base = 0;
  if (threadIdx.x == 0) base = atomicAdd(a.mix_count + scene, __popcll(m));
  base = __shfl(base, 0);
  if (mix) a.mix_list[scene * a.rows_per_scene + base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = (int)(row - scene * a.rows_per_scene);
}

template <bool STAGED, bool COMPACT = false>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_mixopt(MixArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  long row = map_row(a.by_mode, a.rows_per_scene);
  if (COMPACT) {   // wavefront j of a scene takes entries [64 j, 64 j + 64) of the scene's list of rows to mix
    const long vb = virt_block(a.by_mode, a.rows_per_scene);   // (scene_tables stages the tables of THIS block's scene)
    const long scene = (vb * kWave) / a.rows_per_scene;
    const int j = (int)(vb % (a.rows_per_scene / kWave));
    const int n = a.mix_count[scene];
    if (j * kWave >= n) return;                               // whole wavefront: nothing left
    const int idx = j * kWave + threadIdx.x;
    row = idx < n ? scene * a.rows_per_scene + a.mix_list[scene * a.rows_per_scene + idx] : a.N;   // (a.N: idle lane)
  }
  const long slot = (long)blockIdx.x * kWave + threadIdx.x;
  const long P = (long)gridDim.x * kWave;      // slots per plane
  const f4* lanes;
  const float* nei;
  scene_tables<STAGED>(lds, kScratchGrad, a.lane_prep, a.nei_prep, a.K, a.rows_per_scene, row < a.N ? row : a.N - 1, lanes,
                       nei, a.by_mode);
  if (row >= a.N) return;
  const Scratch st = {lds + threadIdx.x, kWave};
  const long b = row / a.rows_per_scene;
  const StlRow r = load_row(a.stlp, a.hl, row);
  constexpr int E = 2 * kT;
  float* U = a.work + slot;
  float* G = a.work + (long)E * P + slot;
  float* B = a.work + 2L * E * P + slot;              // base k, element e at B[(k*E + e) * P]
  float* LM = a.work + (2L + kMixK) * E * P + slot;   // lambda k at LM[k*P], m at LM[(8+k)*P], v at LM[(16+k)*P]
  PSTL_NOUNROLL
  for (int k = 0; k < kMixK; ++k) {
    const f4* src = reinterpret_cast<const f4*>(a.base[k] + row * E);
    PSTL_NOUNROLL
    for (int q = 0; q < E / 4; ++q) {
      const f4 v = src[q];
      float* d = B + ((long)k * E + 4 * q) * P;
      d[0] = v.x, d[P] = v.y, d[2 * P] = v.z, d[3 * P] = v.w;
    }
  }
  const float vr = a.valid[row];
  // the rows to mix: score of the current controls <= 0 on a valid lane (nusc_train.py:1045-1046); COMPACT: k_mix_select
  // has sorted that out already
  const float score0 = COMPACT ? 0.0f
                               : stl_eval<false, -1>(a.env, r, lanes, nei, a.K, DynSrc(a.s0 + b * 4, B, 1.0f, 1.0f, a.env.dt, P),
                                                     st, 0, nullptr, nullptr);
  float* out = a.out + row * E;
  if (!COMPACT && !(score0 <= 0.0f && vr > 0.0f)) {
    PSTL_NOUNROLL
    for (int e = 0; e < E; ++e) out[e] = B[(long)e * P];
    if (a.grad_trace) {
      PSTL_NOUNROLL
      for (int it = 0; it < a.iters; ++it)
        PSTL_NOUNROLL
        for (int k = 0; k < kMixK; ++k) a.grad_trace[((long)it * a.N + row) * kMixK + k] = 0.0f;
    }
    return;
  }
  PSTL_NOUNROLL
  for (int k = 0; k < kMixK; ++k) LM[(long)k * P] = 1.0f, LM[(long)(kMixK + k) * P] = 0.0f, LM[(long)(2 * kMixK + k) * P] = 0.0f;
  const float gs = a.grad_scale * vr;
  const float thres = a.thres;
  auto ratios = [&](float (&rt)[kMixK]) {          // torch.softmax over the 8 weights
    float mx = -INFINITY;
    PSTL_UNROLL
    for (int k = 0; k < kMixK; ++k) {
      rt[k] = LM[(long)k * P];
      mx = fmaxf(mx, rt[k]);
    }
    float sum = 0.0f;
    PSTL_UNROLL
    for (int k = 0; k < kMixK; ++k) {
      rt[k] = expf(rt[k] - mx);
      sum += rt[k];
    }
    PSTL_UNROLL
    for (int k = 0; k < kMixK; ++k) rt[k] = rt[k] / sum;
  };
  for (int it = 0; it < a.iters; ++it) {
    {
      float rt[kMixK];
      ratios(rt);
      PSTL_NOUNROLL
      for (int e = 0; e < E; ++e) {     // base_0 * r_0 + sum_{k >= 1} base_k * r_k   (:1056-1057)
        float acc = B[((long)E + e) * P] * rt[1];
        PSTL_UNROLL
        for (int k = 2; k < kMixK; ++k) acc += B[((long)k * E + e) * P] * rt[k];
        U[(long)e * P] = B[(long)e * P] * rt[0] + acc;
      }
    }
    stl_eval_grad<false, true>(   // (WAVE_ZERO, as in k_trajopt)
        a.env, r, lanes, nei, a.K, a.s0 + b * 4, U, st, 1.0f, 1.0f,
        [=](float sc) { return (thres - sc > 0.0f) ? -gs : 0.0f; },
        [=](int t, float gw, float ga, float, float) {
          G[(long)(2 * t) * P] = gw;
          G[(long)(2 * t + 1) * P] = ga;
        },
        P);
    // d loss / d ratio_k = <G, base_k>; softmax backward; one Adam step on lambda
    float rt[kMixK], gr[kMixK];
    ratios(rt);
    PSTL_UNROLL
    for (int k = 0; k < kMixK; ++k) gr[k] = 0.0f;
    PSTL_NOUNROLL
    for (int e = 0; e < E; ++e) {
      const float g = G[(long)e * P];
      PSTL_UNROLL
      for (int k = 0; k < kMixK; ++k) gr[k] += g * B[((long)k * E + e) * P];
    }
    float dot = 0.0f;
    PSTL_UNROLL
    for (int k = 0; k < kMixK; ++k) dot += rt[k] * gr[k];
    const float neg_step = a.neg_step[it], bc2 = a.bc2_sqrt[it];
    PSTL_UNROLL
    for (int k = 0; k < kMixK; ++k) {
      const float g = rt[k] * (gr[k] - dot);
      if (a.grad_trace) a.grad_trace[((long)it * a.N + row) * kMixK + k] = g;
      float m = LM[(long)(kMixK + k) * P], v = LM[(long)(2 * kMixK + k) * P];
      m = m + 0.1f * (g - m);
      v = v * 0.999f + (0.001f * g) * g;
      const float denom = sqrtf(v) / bc2 + 1e-8f;
      LM[(long)k * P] = LM[(long)k * P] + (neg_step * m) / denom;
      LM[(long)(kMixK + k) * P] = m;
      LM[(long)(2 * kMixK + k) * P] = v;
    }
  }
  PSTL_NOUNROLL
  for (int e = 0; e < E; ++e) out[e] = U[(long)e * P];   // the iterate of the last forward pass (:1071)
}

// prep_stl_cache (nusc_train.py:74-93): the seven signals the formulas read, per row and time step --
// signed lateral distance and heading error to the current / left / right lane (compute_t2l_dist, nusc_api.py:685-739)
// and the clearance to the closest neighbour (compute_shortest_dist_refined, nusc_train.py:142-148).
// out (7, N, T): x2curr_d, x2curr_th, x2left_d, x2left_th, x2right_d, x2right_th, min_nei_d.
__global__ __launch_bounds__(kWave) void k_stl_signals(long N, int rows_per_scene, int K, StlEnv env, const float* s0,
                                                       const float* controls, const float* states, const float* nei_prep,
                                                       const float* lane_prep, float* out) {
  const long row = (long)blockIdx.x * kWave + threadIdx.x;
  if (row >= N) return;
  const long b = row / rows_per_scene;
  const f4* lanes = reinterpret_cast<const f4*>(lane_prep) + b * 3 * kNseg;
  const float* nei = nei_prep + b * (long)K * kT * kNeiPrep;
  DynSrc dyn(s0 ? s0 + b * 4 : lane_prep, controls ? controls + row * (2 * kT) : lane_prep, 1.0f, 1.0f, env.dt);
  GivenSrc giv{states ? reinterpret_cast<const f4*>(states) + row * kT : nullptr};
  const long plane = N * kT;
  for (int t = 0; t < kT; ++t) {
    float x, y, th, v, c, s;
    if (states) giv.get(t, x, y, th, v, c, s); else dyn.get(t, x, y, th, v, c, s);
    float* o = out + row * kT + t;
    for (int m = 0; m < 3; ++m) {
      LaneHit h;
      lane_eval<false>(lanes + m * kNseg, x, y, th, h);
      o[(2 * m) * plane] = h.d;
      o[(2 * m + 1) * plane] = h.th;
    }
    ClearHit ch;
    clearance_eval<false>(env, nei, K, t, x, y, c, s, ch);
    o[6 * plane] = ch.dn;
  }
}

// generate_trajs (nusc_train.py:39-49): T+1 states per row
__global__ void k_generate_trajs(long R, int rows_per_scene, const float* s0, const float* controls, float dt,
                                 float* trajs) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= R) return;
  const float* s = s0 + (row / rows_per_scene) * 4;
  const float* u = controls + row * (2 * kT);
  float x = s[0], y = s[1], th = s[2], v = s[3];
  f4* out = reinterpret_cast<f4*>(trajs) + row * (kT + 1);
  for (int t = 0; t <= kT; ++t) {
    out[t] = f4{x, y, th, v};
    if (t == kT) break;
    const float dx = v * cosf(th), dy = v * sinf(th);
    x = x + dx * dt;
    y = y + dy * dt;
    th = th + u[2 * t] * dt;
    v = v + u[2 * t + 1] * dt;
  }
}

__global__ void k_prepare(long n_nei, long n_lane_pts, const float* nei, const float* l0, const float* l1,
                          const float* l2, float* nei_prep, float* lane_prep) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_nei) {
    float in[7], out[kNeiPrep];
#pragma unroll
    for (int j = 0; j < 7; ++j) in[j] = nei[i * 7 + j];
    prep_neighbor(in, out);
#pragma unroll
    for (int j = 0; j < kNeiPrep; ++j) nei_prep[i * kNeiPrep + j] = out[j];
  }
  if (i < n_lane_pts) {  // i = (b*3 + m)*15 + j
    const long j = i % kNseg, bm = i / kNseg, m = bm % 3, b = bm / 3;
    const float* src = (m == 0 ? l0 : m == 1 ? l1 : l2) + (b * kNseg + j) * 3;
    prep_lane_point(src, j + 1 < kNseg ? src + 3 : nullptr, lane_prep + i * 4);
  }
}

// (a kernel, not hipMemsetAsync: under stream capture -- the closed loop's and bench.py's HIP graphs -- the memset node of this
// ROCm did not zero its 64 bytes on replay, and the counters accumulated on top of the previous replay's)
__global__ void k_zero_words(int n, unsigned* p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// grid-stride row counts; one pair of 64-bit atomics per workgroup (a few hundred in total instead of one per wave on
// two addresses).  Integer counts are order-independent, so the result is reproducible bit for bit.
__global__ __launch_bounds__(256) void k_metrics_rows(long N, const float* scores, const float* valid,
                                                      unsigned long long* counts, uint8_t* sat_mask) {
  __shared__ unsigned int s_sat[4], s_val[4];
  unsigned int sat = 0, val = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const bool s = scores[i] > 0.0f;
    const bool v = valid[i] > 0.0f;
    sat += (s && v) ? 1u : 0u;
    val += v ? 1u : 0u;
    if (sat_mask) sat_mask[i] = s ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sat += __shfl_down(sat, o);
    val += __shfl_down(val, o);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    s_sat[w] = sat;
    s_val[w] = val;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&counts[0], (unsigned long long)(s_sat[0] + s_sat[1] + s_sat[2] + s_sat[3]));
    atomicAdd(&counts[1], (unsigned long long)(s_val[0] + s_val[1] + s_val[2] + s_val[3]));
  }
}

__global__ void k_metrics_scenes(int bs, int S, const float* scores, const float* valid, unsigned long long* counts) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (scene, mode)
  unsigned sat = 0, val = 0;
  if (i == 0) {
    counts[2] = (unsigned long long)bs * S * 3;
    counts[5] = (unsigned long long)bs * 3;
  }
  if (i < (long)bs * 3) {
    const long b = i / 3, m = i % 3;
    bool any = false;
    for (int s = 0; s < S; ++s) any = any || (scores[(b * S + s) * 3 + m] > 0.0f);
    const bool v = valid[(b * S) * 3 + m] > 0.0f;
    sat = (any && v) ? 1u : 0u;
    val = v ? 1u : 0u;
  }
  const unsigned long long bsat = __ballot(sat), bv = __ballot(val);
  if ((threadIdx.x & 63) == 0) {
    if (bsat) atomicAdd(&counts[3], (unsigned long long)__popcll(bsat));
    if (bv) atomicAdd(&counts[4], (unsigned long long)__popcll(bv));
  }
}

}  // namespace
}  // namespace pstl

using namespace pstl;

// the 64 rows of every workgroup share one scene (and the staged tables fit comfortably in LDS)
static bool scene_staged(const pstl_cfg* cfg) { return cfg->rows_per_scene % kWave == 0 && cfg->K <= 16; }
// up to one 64-row group per CU, the guidance kernel runs its latency layout (k_guidance_iter<.., SPLIT>)
static long guidance_split_max_groups() { return device_cus(); }   // (of the current device: pstl_common.hpp)

static int allow_lds(const void* fn, size_t bytes) {
  if (bytes <= 48 * 1024) return PSTL_OK;
  // (remembered per kernel and device: the largest size already allowed -- the attribute call costs host time per launch)
  struct Seen {
    const void* fn;
    int dev;
    size_t bytes;
  };
  static Seen seen[64];
  static int n_seen = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return PSTL_ERR_LAUNCH;
  for (int i = 0; i < n_seen; ++i)
    if (seen[i].fn == fn && seen[i].dev == dev) {
      if (seen[i].bytes >= bytes) return PSTL_OK;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return PSTL_ERR_LAUNCH;
      seen[i].bytes = bytes;
      return PSTL_OK;
    }
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return PSTL_ERR_LAUNCH;
  if (n_seen < 64) seen[n_seen++] = Seen{fn, dev, bytes};
  return PSTL_OK;
}

#if defined(PSTL_STL_STAMP)
// diagnostic builds only: the section table of the stamped kernels (32 x uint64 cycles; slot 31 = wavefronts), then zeroed
extern "C" int pstl_debug_stl_stamps(unsigned long long* out32) {
  if (hipDeviceSynchronize() != hipSuccess) return PSTL_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(pstl::pstl_st_global), 32 * sizeof(unsigned long long)) != hipSuccess) return PSTL_ERR_LAUNCH;
  static const unsigned long long zero[32] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(pstl::pstl_st_global), zero, sizeof(zero)) != hipSuccess) return PSTL_ERR_LAUNCH;
  return PSTL_OK;
}
#endif

extern "C" int pstl_prepare_scene(const pstl_cfg* cfg, const float* neighbors_traj, const float* currlane,
                                  const float* leftlane, const float* rightlane, float* nei_prep, float* lane_prep,
                                  void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!currlane || !leftlane || !rightlane || !lane_prep || (cfg->K > 0 && (!neighbors_traj || !nei_prep)))
    return PSTL_ERR_ARG;
  const long n_nei = (long)cfg->bs * cfg->K * kT, n_lane = (long)cfg->bs * 3 * kNseg;
  const long n = n_nei > n_lane ? n_nei : n_lane;
  hipLaunchKernelGGL(k_prepare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), n_nei, n_lane,
                     neighbors_traj, currlane, leftlane, rightlane, nei_prep, lane_prep);
  return launch_status();
}

extern "C" int pstl_generate_trajs(const pstl_cfg* cfg, const float* s0, const float* controls, float* trajs,
                                   void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!s0 || !controls || !trajs) return PSTL_ERR_ARG;
  const long R = n_rows(cfg);
  hipLaunchKernelGGL(k_generate_trajs, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, as_stream(stream), R,
                     cfg->rows_per_scene, s0, controls, cfg->dt, trajs);
  return launch_status();
}

extern "C" int pstl_stl_forward(const pstl_cfg* cfg, const float* s0, const float* controls, const float* states,
                                int reps, const float* nei_prep, const float* lane_prep, const float* stlp,
                                const float* hl, float* scores, float* scores3, float* sel_controls,
                                float* sel_scores, int32_t* sel_idx, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if ((!states && (!s0 || !controls)) || !lane_prep || !stlp || !hl || !scores || reps < 1) return PSTL_ERR_ARG;
  if (!aligned16(controls) || !aligned16(states) || !aligned16(sel_controls)) return PSTL_ERR_ARG;
  if (cfg->K > 0 && !nei_prep) return PSTL_ERR_ARG;
  if (sel_controls && (!sel_scores || !sel_idx)) return PSTL_ERR_ARG;
  StlArgs a;
  a.N = n_rows(cfg);
  a.rows_per_scene = cfg->rows_per_scene;
  a.K = cfg->K;
  a.reps = reps;
  a.env = make_env(cfg->tau, cfg->dt, cfg->ego_L, cfg->ego_W);
  a.s0 = s0;
  a.controls = controls;
  a.states = states;
  a.nei_prep = nei_prep;
  a.lane_prep = lane_prep;
  a.stlp = stlp;
  a.hl = hl;
  a.scores = scores;
  a.scores3 = scores3;
  a.sel_controls = sel_controls;
  a.sel_scores = sel_scores;
  a.sel_idx = sel_idx;
  dim3 grid((unsigned)((a.N + kWave - 1) / kWave));
  // fewer wavefronts than two per SIMD: one candidate per wavefront instead of `reps` in sequence (closed-loop caller:
  // 192 rows x 5 candidates = 15 wavefronts of one candidate each instead of 3 of five)
  a.rep_split = (reps > 1 && (long)grid.x * reps <= 2048) ? 1 : 0;
  const bool select_after = a.rep_split && sel_controls && controls;
  if (a.rep_split) {
    grid.y = (unsigned)reps;
    if (select_after) a.sel_controls = nullptr;
  }
  const bool staged = scene_staged(cfg);
  // (the selected-formula kernel gains from wavefronts of one (scene, mode): its formula branches become wave-uniform; the
  // all-three kernel evaluates everything for every row anyway)
  a.by_mode = (!scores3 && rows_by_mode(cfg, staged)) ? 1 : 0;
  const size_t lds = stl_lds_bytes(scores3 ? kScratchFwd3 : kScratchFwd, cfg->K, staged) + dbg_lds_pad("PSTL_DBG_LDS_PAD_FORWARD");
  void (*fn)(StlArgs);
  if (cfg->flags & PSTL_FLAG_NORM_STL) {
    if (states)
      fn = scores3 ? (staged ? k_stl_forward<true, true, true, true> : k_stl_forward<true, false, true, true>)
                   : (staged ? k_stl_forward<false, true, true, true> : k_stl_forward<false, false, true, true>);
    else
      fn = scores3 ? (staged ? k_stl_forward<true, true, false, true> : k_stl_forward<true, false, false, true>)
                   : (staged ? k_stl_forward<false, true, false, true> : k_stl_forward<false, false, false, true>);
  } else if (states)
    fn = scores3 ? (staged ? k_stl_forward<true, true, true> : k_stl_forward<true, false, true>)
                 : (staged ? k_stl_forward<false, true, true> : k_stl_forward<false, false, true>);
  else
    fn = scores3 ? (staged ? k_stl_forward<true, true, false> : k_stl_forward<true, false, false>)
                 : (staged ? k_stl_forward<false, true, false> : k_stl_forward<false, false, false>);
  // few wavefronts (a small batch): the latency layout, where it is instantiated (staged tables, the selected formula, controls)
  const bool split = staged && !scores3 && controls && !states && (long)grid.x * grid.y <= guidance_split_max_groups();
  size_t lds_total = lds;
  if (split) {
    lds_total += (size_t)kGeoSlots * kT * kWave * sizeof(float);
    fn = (cfg->flags & PSTL_FLAG_NORM_STL) ? k_stl_forward<false, true, false, true, true> : k_stl_forward<false, true, false, false, true>;
  }
  if (int e = allow_lds(reinterpret_cast<const void*>(fn), lds_total)) return e;
  hipLaunchKernelGGL(fn, grid, dim3(split ? kSplitWaves * kWave : kWave), lds_total, as_stream(stream), a);
  if (select_after)
    hipLaunchKernelGGL(k_stl_select, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, as_stream(stream), a.N, reps,
                       (const float*)scores, controls, sel_controls, sel_scores, sel_idx);
  return launch_status();
}

extern "C" int pstl_stl_backward(const pstl_cfg* cfg, const float* s0, const float* controls, const float* nei_prep,
                                 const float* lane_prep, const float* stlp, const float* hl, const float* dscore,
                                 float* dcontrols, float* scores, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!s0 || !controls || !lane_prep || !stlp || !hl || !dcontrols) return PSTL_ERR_ARG;
  if (!aligned16(controls) || !aligned16(dcontrols)) return PSTL_ERR_ARG;
  if (cfg->K > 0 && !nei_prep) return PSTL_ERR_ARG;
  GradArgs a;
  a.N = n_rows(cfg);
  a.rows_per_scene = cfg->rows_per_scene;
  a.K = cfg->K;
  a.env = make_env(cfg->tau, cfg->dt, cfg->ego_L, cfg->ego_W);
  a.wscale = 1.0f;
  a.ascale = 1.0f;
  a.s0 = s0;
  a.u = controls;
  a.nei_prep = nei_prep;
  a.lane_prep = lane_prep;
  a.stlp = stlp;
  a.hl = hl;
  a.dscore = dscore;
  a.dcontrols = dcontrols;
  a.scores = scores;
  const bool staged = scene_staged(cfg);
  a.by_mode = rows_by_mode(cfg, staged) ? 1 : 0;
  const size_t lds = stl_lds_bytes(kScratchGrad, cfg->K, staged);
  void (*fn)(GradArgs) = (cfg->flags & PSTL_FLAG_NORM_STL) ? (staged ? k_stl_backward<true, true> : k_stl_backward<false, true>)
                                                           : (staged ? k_stl_backward<true> : k_stl_backward<false>);
  if (int e = allow_lds(reinterpret_cast<const void*>(fn), lds)) return e;
  hipLaunchKernelGGL(fn, dim3((unsigned)((a.N + kWave - 1) / kWave)), dim3(kWave), lds, as_stream(stream), a);
  return launch_status();
}

extern "C" int pstl_guidance_step(const pstl_cfg* cfg, const float* s0, const float* nei_prep, const float* lane_prep,
                                  const float* stlp, const float* hl, const float* valid, float grad_scale, int niters,
                                  const float* adam_neg_step, const float* adam_bc2_sqrt, float beta_i, int step,
                                  const float* z, float* mu_x_inout, float* work, float* emit_out, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!s0 || !lane_prep || !stlp || !hl || !valid || !mu_x_inout || !adam_neg_step || !adam_bc2_sqrt || niters < 1)
    return PSTL_ERR_ARG;
  if (cfg->K > 0 && !nei_prep) return PSTL_ERR_ARG;
  if (niters > 1 && !work) return PSTL_ERR_ARG;
  if (!aligned16(mu_x_inout) || !aligned16(emit_out) || !aligned16(z)) return PSTL_ERR_ARG;
  GuideArgs a;
  a.N = n_rows(cfg);
  a.rows_per_scene = cfg->rows_per_scene;
  a.K = cfg->K;
  a.env = make_env(cfg->tau, cfg->dt, cfg->ego_L, cfg->ego_W);
  a.wscale = cfg->w_max;
  a.ascale = cfg->a_max;
  a.thres = (cfg->flags & PSTL_FLAG_MAXIMIZE) ? 100.0f : cfg->thres;
  a.grad_scale = grad_scale;
  a.beta_i = beta_i;
  a.sqrt_beta = sqrtf(beta_i);
  a.niters = niters;
  a.clip = (cfg->flags & PSTL_FLAG_CLIP) ? 1 : 0;
  a.s0 = s0;
  a.nei_prep = nei_prep;
  a.lane_prep = lane_prep;
  a.stlp = stlp;
  a.hl = hl;
  a.valid = valid;
  a.z = z;
  a.rng = (cfg->flags & PSTL_FLAG_RNG) ? 1 : 0;
  a.step = step;
  a.seed = cfg->seed;
  a.dyn = cfg->dyn;
  a.row_offset = (long)cfg->row_offset;
  a.mu = mu_x_inout;
  a.work = work;
  a.emit_out = emit_out;
  const dim3 grid((unsigned)((a.N + kWave - 1) / kWave));
  const bool staged = scene_staged(cfg);
  a.by_mode = rows_by_mode(cfg, staged) ? 1 : 0;
  const size_t lds = stl_lds_bytes(kScratchGrad, cfg->K, staged) + dbg_lds_pad("PSTL_DBG_LDS_PAD_GUIDANCE");
  void (*fn)(GuideArgs) = niters > 1 ? (staged ? k_guidance_iter<true, true> : k_guidance_iter<true, false>)
                                     : (staged ? k_guidance_iter<false, true> : k_guidance_iter<false, false>);
  if (cfg->flags & PSTL_FLAG_NORM_STL)
    fn = niters > 1 ? (staged ? k_guidance_iter<true, true, true> : k_guidance_iter<true, false, true>)
                    : (staged ? k_guidance_iter<false, true, true> : k_guidance_iter<false, false, true>);
  // few wavefronts: the latency layout (kSplitWaves waves per 64 rows, the geometry split over them by time step)
  // (up to two groups per CU: the ten-wave workgroups then run in two rounds of ~40 us, against ~100 us for one round of lone
  // wavefronts; beyond that the one-wave kernel's wavefronts start to share SIMDs and win)
  const bool split = staged && (long)grid.x <= 2 * guidance_split_max_groups();
  size_t lds_total = lds;
  if (split) {
    lds_total = stl_lds_bytes(kScratchGradPre, cfg->K, true) + (size_t)kGeoFloats * kWave * sizeof(float);   // + the geometry
    if (cfg->flags & PSTL_FLAG_NORM_STL) fn = niters > 1 ? k_guidance_iter<true, true, true, true> : k_guidance_iter<false, true, true, true>;
    else fn = niters > 1 ? k_guidance_iter<true, true, false, true> : k_guidance_iter<false, true, false, true>;
  }
  if (int e = allow_lds(reinterpret_cast<const void*>(fn), lds_total)) return e;
  for (int j = 0; j < niters; ++j) {
    a.iter = j;
    a.neg_step = adam_neg_step[j];
    a.bc2_sqrt = adam_bc2_sqrt[j];
    hipLaunchKernelGGL(fn, grid, dim3(split ? kSplitWaves * kWave : kWave), lds_total, as_stream(stream), a);
    if (int e = launch_status()) return e;
  }
  return PSTL_OK;
}

extern "C" int pstl_trajopt(const pstl_cfg* cfg, const float* s0, const float* nei_prep, const float* lane_prep,
                            const float* stlp, const float* hl, const float* valid, float thres, float grad_scale,
                            float reg_scale, int iters, const float* adam_neg_step, const float* adam_bc2_sqrt, int resume,
                            float* params_inout, float* work, float* scores, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (cfg->flags & PSTL_FLAG_NORM_STL) return PSTL_ERR_SHAPE;   // the traj-opt loop is built for the default formulas only
  if (!s0 || !lane_prep || !stlp || !hl || !valid || !params_inout || !work || !adam_neg_step || !adam_bc2_sqrt ||
      iters < 1)
    return PSTL_ERR_ARG;
  if (cfg->K > 0 && !nei_prep) return PSTL_ERR_ARG;
  TrajoptArgs a;
  a.N = n_rows(cfg);
  a.rows_per_scene = cfg->rows_per_scene;
  a.K = cfg->K;
  a.env = make_env(cfg->tau, cfg->dt, cfg->ego_L, cfg->ego_W);
  a.thres = thres;
  a.grad_scale = grad_scale;
  a.reg_scale = reg_scale;
  a.w_max2 = cfg->w_max * cfg->w_max;
  a.a_max2 = cfg->a_max * cfg->a_max;
  a.iters = iters;
  a.neg_step = adam_neg_step;
  a.bc2_sqrt = adam_bc2_sqrt;
  a.s0 = s0;
  a.nei_prep = nei_prep;
  a.lane_prep = lane_prep;
  a.stlp = stlp;
  a.hl = hl;
  a.valid = valid;
  a.params = params_inout;
  a.work = work;
  a.scores = scores;
  a.resume = resume;
  const bool staged = scene_staged(cfg);
  a.by_mode = rows_by_mode(cfg, staged) ? 1 : 0;
  const size_t lds = stl_lds_bytes(kScratchGrad, cfg->K, staged);
  void (*fn)(TrajoptArgs) = staged ? k_trajopt<true> : k_trajopt<false>;
  if (int e = allow_lds(reinterpret_cast<const void*>(fn), lds)) return e;
  hipLaunchKernelGGL(fn, dim3((unsigned)((a.N + kWave - 1) / kWave)), dim3(kWave), lds, as_stream(stream), a);
  return launch_status();
}

extern "C" size_t pstl_refinement_work_floats(const pstl_cfg* cfg) {
  if (check_cfg(cfg)) return 0;
  const long blocks = (n_rows(cfg) + kWave - 1) / kWave;
  // element-major planes of the mixing loop + the per-scene lists of rows to mix (a count per scene, an index per row)
  return (size_t)kMixPlanes * (size_t)blocks * kWave + (size_t)((cfg->bs + 63) / 64) * 64 + (size_t)n_rows(cfg);
}

extern "C" int pstl_refinement(const pstl_cfg* cfg, const float* s0, const float* nei_prep, const float* lane_prep,
                               const float* stlp, const float* hl, const float* valid, float thres, float grad_scale,
                               int iters, const float* adam_neg_step, const float* adam_bc2_sqrt, const float* controls,
                               const float* list, int n_list, const int32_t* list_idx, float* work, float* out_controls,
                               float* grad_trace, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (cfg->flags & PSTL_FLAG_NORM_STL) return PSTL_ERR_SHAPE;   // --refinement: default formulas only
  if (!s0 || !lane_prep || !stlp || !hl || !valid || !controls || !list || !list_idx || !work || !out_controls ||
      !adam_neg_step || !adam_bc2_sqrt || iters < 1)
    return PSTL_ERR_ARG;
  if (cfg->K > 0 && !nei_prep) return PSTL_ERR_ARG;
  MixArgs a;
  a.N = n_rows(cfg);
  a.rows_per_scene = cfg->rows_per_scene;
  a.K = cfg->K;
  a.env = make_env(cfg->tau, cfg->dt, cfg->ego_L, cfg->ego_W);
  a.thres = thres;
  a.grad_scale = grad_scale;
  a.iters = iters;
  a.neg_step = adam_neg_step;
  a.bc2_sqrt = adam_bc2_sqrt;
  a.s0 = s0;
  a.nei_prep = nei_prep;
  a.lane_prep = lane_prep;
  a.stlp = stlp;
  a.hl = hl;
  a.valid = valid;
  a.base[0] = controls;
  for (int k = 1; k < kMixK; ++k) {
    if (list_idx[k - 1] < 0 || list_idx[k - 1] >= n_list) return PSTL_ERR_ARG;   // the reference: IndexError
    a.base[k] = list + (long)list_idx[k - 1] * a.N * (2 * kT);
  }
  a.work = work;
  a.out = out_controls;
  a.grad_trace = grad_trace;
  const bool staged = scene_staged(cfg);
  a.by_mode = rows_by_mode(cfg, staged) ? 1 : 0;
  const size_t lds = stl_lds_bytes(kScratchGrad, cfg->K, staged);
  const unsigned blocks = (unsigned)((a.N + kWave - 1) / kWave);
  a.mix_count = nullptr;
  a.mix_list = nullptr;
#ifdef PSTL_MIX_NO_COMPACT   // timing/bit-comparison builds only (tools/dbg/refinement_time.py): the uncompacted walk
  if (staged) {
    void (*fl)(MixArgs) = k_mixopt<true>;
    if (int e = allow_lds(reinterpret_cast<const void*>(fl), lds)) return e;
    hipLaunchKernelGGL(fl, dim3(blocks), dim3(kWave), lds, as_stream(stream), a);
    return launch_status();
  }
#endif
  if (staged) {   // (rows_per_scene % 64 == 0: a wavefront's rows share a scene) pack the rows to mix per scene first
    hipStream_t st = as_stream(stream);
    int* ints = reinterpret_cast<int*>(work + (size_t)kMixPlanes * blocks * kWave);
    a.mix_count = ints;
    a.mix_list = ints + ((cfg->bs + 63) / 64) * 64;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((cfg->bs + 255) / 256)), dim3(256), 0, st, (int)cfg->bs,
                       reinterpret_cast<unsigned*>(a.mix_count));
    const size_t lds_sel = stl_lds_bytes(kScratchFwd, cfg->K, true);
    if (int e = allow_lds(reinterpret_cast<const void*>(k_mix_select<true>), lds_sel)) return e;
    hipLaunchKernelGGL(k_mix_select<true>, dim3(blocks), dim3(kWave), lds_sel, st, a);
    void (*fc)(MixArgs) = k_mixopt<true, true>;
    if (int e = allow_lds(reinterpret_cast<const void*>(fc), lds)) return e;
    hipLaunchKernelGGL(fc, dim3(blocks), dim3(kWave), lds, st, a);
    return launch_status();
  }
  void (*fn)(MixArgs) = k_mixopt<false>;
  if (int e = allow_lds(reinterpret_cast<const void*>(fn), lds)) return e;
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(kWave), lds, as_stream(stream), a);
  return launch_status();
}

extern "C" int pstl_stl_signals(const pstl_cfg* cfg, const float* s0, const float* controls, const float* states,
                                const float* nei_prep, const float* lane_prep, float* signals, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!lane_prep || !signals || (!states && (!s0 || !controls)) || (cfg->K > 0 && !nei_prep)) return PSTL_ERR_ARG;
  if (!aligned16(controls) || !aligned16(states)) return PSTL_ERR_ARG;
  const long N = n_rows(cfg);
  hipLaunchKernelGGL(k_stl_signals, dim3((unsigned)((N + kWave - 1) / kWave)), dim3(kWave), 0, as_stream(stream), N,
                     cfg->rows_per_scene, cfg->K, make_env(cfg->tau, cfg->dt, cfg->ego_L, cfg->ego_W), s0, controls, states,
                     nei_prep, lane_prep, signals);
  return launch_status();
}

extern "C" int pstl_select_plan(const pstl_cfg* cfg, const float* scores, const float* controls, const float* packed_status,
                                float* out4, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!scores || !controls || !out4) return PSTL_ERR_ARG;
  if (cfg->bs != 1 || cfg->rows_per_scene != 3 * cfg->S) return PSTL_ERR_SHAPE;   // the closed loop plans one scene at a time
  hipLaunchKernelGGL(k_select_plan, dim3(1), dim3(kWave), 0, as_stream(stream), cfg->S, scores, controls,
                     reinterpret_cast<const unsigned*>(packed_status), out4);
  return launch_status();
}

extern "C" int pstl_reduce_metrics(const pstl_cfg* cfg, const float* scores, const float* valid, uint64_t* counts,
                                   uint8_t* sat_mask, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!scores || !valid || !counts) return PSTL_ERR_ARG;
  if (cfg->rows_per_scene != 3 * cfg->S) return PSTL_ERR_SHAPE;
  const long N = n_rows(cfg);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, st, 16, reinterpret_cast<unsigned*>(counts));
  auto* c = reinterpret_cast<unsigned long long*>(counts);
  const long nb = (N + 255) / 256;
  hipLaunchKernelGGL(k_metrics_rows, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, st, N, scores, valid, c,
                     sat_mask);
  hipLaunchKernelGGL(k_metrics_scenes, dim3((unsigned)(((long)cfg->bs * 3 + 255) / 256)), dim3(256), 0, st, cfg->bs,
                     cfg->S, scores, valid, c);
  return launch_status();
}
