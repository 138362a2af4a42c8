// diversity_kernels.hip -- the metrics the reference's sampling harness computes on the CPU after its timer stops
// (nusc_train.py:1107-1140), as one gfx950 kernel so that the final reduction of a sharded run is a handful of numbers:
//   measure_diversity ......... nusc_api.py:817-875   masked std over the satisfied samples, per-step convex-hull area
//   measure_extra_diversity ... nusc_api.py:894-936   entropies of scores / controls (utils.py:388-417), occupancy area
//                                                     of a 100x100 histogram (compute_area, nusc_api.py:878-891)
//   compute_ade_fde ........... nusc_train.py:877-887
//
// One wavefront per (scene, mode); lane = sample (S <= 64).  The trajectory of a sample is rolled out in registers
// (same arithmetic as k_generate_trajs), everything else is wavefront reductions (ballot/popcount for the histograms,
// xor-shuffles for sums and extrema).  The convex hulls are Andrew's monotone chain: a bitonic sort across the 64 lanes
// puts the satisfied samples' points of every time step into LDS in lexicographic order, then 40 lanes (time step x
// {lower, upper}) walk their chain (top two stack entries in registers, the rest as byte indices in LDS) and accumulate
// the shoelace sum in float64 (MI355X runs fp64 VALU at full rate).  The 40 control histograms (fixed, monotone edges) are
// counted with one LDS atomic per value and their p*log2(p) terms spread over the lanes; the score histogram, whose edges
// can degenerate, tests every bin on its own with ballots as the reference does.  Measured at 4096 scenes x 64 x 3
// (tools/dbg/div_ablation.sh): 0.78 ms = hull sort 0.31 + hull scan 0.25 + entropies 0.08 + std 0.07 + occupancy 0.05 +
// rollout/ADE 0.05 (the ballot form of the control histograms cost 0.28).  Compile with -ffp-contract=off: histogram and
// entropy bin edges are float32 expressions that must round like the reference's separate torch ops.
#include "pstl_common.hpp"

#include <type_traits>

namespace pstl {
namespace {

constexpr int kT = 20;
constexpr int kWave = 64;
constexpr int kEntBins = 10;
constexpr int kHist = 100;                                   // bins per axis of compute_area
constexpr int kHistWords = (kHist * kHist + 31) / 32;

struct DivArgs {
  int bs, S, gt_stride;
  float dt, w_max, a_max;
  const float* s0;        // (bs,4)
  const float* gt;        // (bs,T,gt_stride): ground-truth ego states, first 4 components x,y,th,v
  const float* controls;  // (N,40) physical units
  const float* scores;    // (N,)
  const float* valid;     // (N,)
  const float* alphas;    // (11,) torch.linspace(0,1,11) as the host computes it
  double* per_mode;       // (bs,3,8)
  float* per_scene;       // (bs,2), pre-set to +inf
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Bin counts of one 10-bin histogram of compute_entropy (utils.py:388-417): every bin is tested on its own, exactly as
// the reference's `spotted` tensor does (edges are float32 blends xmin*(1-a) + xmax*a and need not be monotone when
// xmax - xmin is a few ulps).  The counts are wave-uniform; they are parked in lane (slot*10 + k) of `cnt`/`tot` so that
// the p*log2(p) terms of six histograms are evaluated by one pass over the wavefront (see entropy_terms).
__device__ __forceinline__ void hist10(float x, bool in, float xmin, float xmax, const float* al, int slot, int lane,
                                       float& cnt, float& tot) {
  float c[kEntBins];
  float total = 0.0f;
  float lo = xmin * (1.0f - al[0]) + xmax * al[0];
#pragma unroll
  for (int k = 0; k < kEntBins; ++k) {
    const float hi = xmin * (1.0f - al[k + 1]) + xmax * al[k + 1];
    c[k] = (float)__popcll(__ballot(in && x >= lo && x < hi));
    total += c[k];
    lo = hi;
  }
#pragma unroll
  for (int k = 0; k < kEntBins; ++k)
    if (lane == slot * kEntBins + k) {
      cnt = c[k];
      tot = total;
    }
}
// -p log2(clip(p)) of the bin this lane holds (lanes >= 60 hold nothing)
__device__ __forceinline__ float entropy_term(float cnt, float tot, int lane) {
  const float p = cnt / fmaxf(tot, 1e-5f);
  return lane < 6 * kEntBins ? (-p) * log2f(fmaxf(p, 1e-5f)) : 0.0f;
}

// bin of x among the 101 float32 edges torch.linspace(lo, hi, 101) produces (scalar formula), found the way
// torch.histogramdd does: linear guess, then a local search in [guess-1, guess+2); rightmost edge inclusive.
__device__ __forceinline__ int hist_bin(float x, float lo, float hi, float step, float scale) {
  const int pos = (int)((x - lo) * (float)kHist / scale);
  int res = pos - 2;
#pragma unroll
  for (int d = -1; d <= 1; ++d) {   // edges pos-1, pos, pos+1: the bin is (index of the first edge > x) - 1
    const int k = pos + d;
    const float e = k < (kHist + 1) / 2 ? lo + step * (float)k : hi - step * (float)(kHist - k);
    if (k >= 0 && k <= kHist && !(e > x)) res = k;
  }
  res = res > kHist - 1 ? kHist - 1 : res;
  return res < 0 ? 0 : res;
}

__device__ __forceinline__ bool lex_less(float ax, float ay, float bx, float by) {
  return ax < bx || (ax == bx && ay < by);
}

#ifndef PSTL_DIV_SKIP
#define PSTL_DIV_SKIP 0   // timing-only ablations (tools/dbg): 1 std, 2 hull, 4 | 32 hull scan (32 keeps the sort), 8 entropies, 16 occupancy
#endif
// SPLIT (small batches: a (scene, mode) is ONE wavefront, a chain of ~30 k dependent instructions -- 130 us whatever the batch):
// the workgroup is kDivWaves = 10 wavefronts over the same 64 samples; every wave rolls the trajectories out (cheap), wave q
// sorts and scans the hulls of time steps 2q, 2q + 1, the chains'
// shoelace sums meet in LDS in the lane order the single wave has them, and wave 0 finishes (std, entropies, occupancy) while
// the others have left.  Same operations, same summation orders: the same bits.
constexpr int kDivWaves = 10;
template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? kDivWaves * kWave : kWave) void k_diversity(DivArgs a) {
  __shared__ float2 s_pts[kT][kWave];          // satisfied samples' points per time step, sorted (x, then y)
  __shared__ uint8_t s_stack[2 * kT][kWave];   // chain stacks (indices into s_pts[t])
  __shared__ unsigned s_occ[kHistWords];
  __shared__ float s_al[kEntBins + 1];
  __shared__ double s_sh[2 * kT];              // SPLIT: the shoelace sum of chain (t, lower | upper)
  static_assert(!SPLIT || kT == 2 * kDivWaves, "two time steps per wave");

  const int lane = SPLIT ? (int)(threadIdx.x & (kWave - 1)) : (int)threadIdx.x;
  const int wq = SPLIT ? (int)(threadIdx.x / kWave) : 0;
  const int b = blockIdx.x / 3, mode = blockIdx.x % 3;
  const int S = a.S;
  const bool live = lane < S;
  const long row = ((long)b * S + (live ? lane : 0)) * 3 + mode;
  if (wq == 0) {
    if (lane <= kEntBins) s_al[lane] = a.alphas[lane];
    for (int i = lane; i < kHistWords; i += kWave) s_occ[i] = 0u;
  }

  const float score = a.scores[row];
  const bool rvalid = a.valid[row] > 0.0f;
  const bool sat = live && score > 0.0f;
  const bool vsat = sat && rvalid;
  const unsigned long long satmask = __ballot(sat), vsatmask = __ballot(vsat);
  const int n_sat = __popcll(satmask);
  const bool mode_valid = a.valid[((long)b * S) * 3 + mode] > 0.0f;

  // ---- rollout (generate_trajs, nusc_train.py:39-49) fused with ADE/FDE (nusc_train.py:877-887) -----------------
  const float* s = a.s0 + (long)b * 4;
  const float* u = a.controls + row * (2 * kT);
  const float* gt = a.gt + (long)b * kT * a.gt_stride;
  const float x0 = s[0], y0 = s[1];
  float x = x0, y = y0, th = s[2], v = s[3];
  float xr[kT], yr[kT], ct[kT], st[kT];
  const float m = rvalid ? 1.0f : 0.0f, pad = (1.0f - m) * 10000.0f;
  float err_sum = 0.0f, err_last = 0.0f;
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    xr[t] = x - x0;
    yr[t] = y - y0;
    const float* g = gt + t * a.gt_stride;
    const float e0 = (g[0] - x) * m + pad, e1 = (g[1] - y) * m + pad, e2 = (g[2] - th) * m + pad,
                e3 = (g[3] - v) * m + pad;
    const float et = ((e0 * e0 + e1 * e1) + e2 * e2) + e3 * e3;
    err_sum += et;
    err_last = et;
    ct[t] = cosf(th);
    st[t] = sinf(th);
    const float dx = v * ct[t], dy = v * st[t];
    x = x + dx * a.dt;
    y = y + dy * a.dt;
    th = th + u[2 * t] * a.dt;
    v = v + u[2 * t + 1] * a.dt;
  }
  if (wq == 0) {
    const float ade = wave_min(live ? err_sum / (float)kT : INFINITY);
    const float fde = wave_min(live ? err_last : INFINITY);
    if (lane == 0) {   // non-negative floats order like their bit patterns; min is order-independent => reproducible
      atomicMin(reinterpret_cast<unsigned*>(a.per_scene) + b * 2, __float_as_uint(ade));
      atomicMin(reinterpret_cast<unsigned*>(a.per_scene) + b * 2 + 1, __float_as_uint(fde));
    }
  }
  __syncthreads();

  // ---- masked std over the satisfied samples, mean over the 40 position features (nusc_api.py:824-831) ----------
  // variance of feature f lands in lane f, so that a single sqrt serves all 40 features
  double std_acc = 0.0;
  auto masked_std = [&]() {
  if (n_sat > 0 && !(PSTL_DIV_SKIP & 1)) {
    const double inv = 1.0 / (double)n_sat;
    double var = 0.0;
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      const double px = sat ? (double)xr[t] : 0.0, py = sat ? (double)yr[t] : 0.0;
      const double mx = wave_sum(px) * inv, my = wave_sum(py) * inv;
      const double vx = wave_sum(px * px) * inv - mx * mx, vy = wave_sum(py * py) * inv - my * my;
      if (lane == 2 * t) var = vx;
      if (lane == 2 * t + 1) var = vy;
    }
    std_acc = wave_sum(lane < 2 * kT ? sqrt(fmax(var, 0.0)) : 0.0) / (double)(2 * kT);
  }
  };
  if (!SPLIT) masked_std();   // (SPLIT: after the hulls, so that the other waves do not wait for wave 0 at the hull's barrier)

  // ---- per-step convex-hull area of the satisfied samples (nusc_api.py:838-865) --------------------------------
  double vol = 0.0;
  if (mode_valid && n_sat >= 3 && !(PSTL_DIV_SKIP & 2)) {
    // (Round 6: B time steps per pass through the sorting network -- the 21 x 2 cross-lane exchanges of ONE sort are a chain of
    // ~120-cycle ds_bpermute round trips, which is what the sort's 0.31 ms were; B independent sorts in lock-step put B
    // exchanges in flight per stage.  The single wave takes four steps at a time, a wave of the ten-wave layout its two.)
    auto sort_steps = [&](auto b_tag, int tfirst) {
      constexpr int B = decltype(b_tag)::value;
      float px[B], py[B];
#pragma unroll
      for (int b = 0; b < B; ++b) {   // unsatisfied samples carry +inf and end up behind the others
        float xv = 0.0f, yv = 0.0f;
#pragma unroll
        for (int t = 0; t < kT; ++t)   // (constant indices into the register arrays; tfirst is wave-uniform)
          if (t == tfirst + b) xv = xr[t], yv = yr[t];
        px[b] = sat ? xv : INFINITY, py[b] = sat ? yv : INFINITY;
      }
#pragma unroll
      for (int k = 2; k <= kWave; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const bool up = (lane & k) == 0;            // this block of k lanes sorts ascending
          const bool lower = (lane & j) == 0;         // this lane keeps the smaller element of its pair when ascending
          const bool take_min = (up == lower);
          float qx[B], qy[B];
#pragma unroll
          for (int b = 0; b < B; ++b) qx[b] = __shfl_xor(px[b], j), qy[b] = __shfl_xor(py[b], j);
#pragma unroll
          for (int b = 0; b < B; ++b) {
            const bool q_less = lex_less(qx[b], qy[b], px[b], py[b]);
            const bool p_less = lex_less(px[b], py[b], qx[b], qy[b]);
            const bool swap = take_min ? q_less : p_less;
            px[b] = swap ? qx[b] : px[b];
            py[b] = swap ? qy[b] : py[b];
          }
        }
      }
#pragma unroll
      for (int b = 0; b < B; ++b) s_pts[tfirst + b][lane] = make_float2(px[b], py[b]);
    };
    if (SPLIT) {
      sort_steps(std::integral_constant<int, 2>{}, 2 * wq);
    } else {
#pragma unroll
      for (int t0 = 0; t0 < kT; t0 += 4) sort_steps(std::integral_constant<int, 4>{}, t0);   // (unrolled: constant register indices)
    }
    __syncthreads();
    double sh = 0.0;
    if (PSTL_DIV_SKIP & 32) sh = (double)s_pts[lane % kT][lane].x;   // (keeps the sort alive when the scan is ablated)
    const int chain = SPLIT ? 4 * wq + lane : lane;          // (t, lower | upper): the single wave's lane
    if ((SPLIT ? lane < 4 : lane < 2 * kT) && !(PSTL_DIV_SKIP & (4 | 32))) {
      const int t = chain >> 1;
      const bool upper = chain & 1;
      const float2* P = s_pts[t];
      uint8_t* stk = s_stack[chain];
      const float2 org = P[0];
      // the two topmost stack entries live in registers: o (below), p (top); k = stack size
      int k = 0;
      float2 o = org, p = org;
      float2 qn = P[upper ? n_sat - 1 : 0];     // (the next point is requested one iteration ahead: its LDS latency sat in front
      for (int ii = 0; ii < n_sat; ++ii) {      //  of every iteration's first cross product)
        const int i = upper ? n_sat - 1 - ii : ii;
        const float2 q = qn;
        if (ii + 1 < n_sat) qn = P[upper ? i - 1 : i + 1];
        while (k >= 2) {
          const double cr = ((double)p.x - (double)o.x) * ((double)q.y - (double)o.y) -
                            ((double)p.y - (double)o.y) * ((double)q.x - (double)o.x);
          if (cr > 0.0) break;
          --k;                                   // pop: the new top is o, the entry below it comes back from LDS
          p = o;
          if (k >= 2) o = P[stk[k - 2]];
        }
        stk[k++] = (uint8_t)i;
        o = p;
        p = q;
      }
      float2 prev = P[stk[0]];
      for (int e = 1; e < k; ++e) {
        const float2 cur = P[stk[e]];
        sh += ((double)prev.x - (double)org.x) * ((double)cur.y - (double)org.y) -
              ((double)prev.y - (double)org.y) * ((double)cur.x - (double)org.x);
        prev = cur;
      }
    }
    if (SPLIT) {
      if (lane < 4) s_sh[chain] = sh;
      __syncthreads();
      sh = lane < 2 * kT ? s_sh[lane] : 0.0;
    }
    vol = 0.5 * wave_sum(sh);
  }
  if (SPLIT) {
    if (wq != 0) return;   // (the remaining barriers are wave 0's alone: a barrier counts the waves that have not ended)
    masked_std();
  }

  // ---- entropies (nusc_api.py:906-926): 41 histograms (scores, w_t, a_t), six at a time over the wavefront --------
  float ent_s = 0.0f, ent_w = 0.0f, ent_a = 0.0f;
  if (!(PSTL_DIV_SKIP & 8)) {
    const float smin = wave_min(vsat ? score : INFINITY) - 1e-5f;
    const float smax = wave_max(vsat ? score : -INFINITY) + 1e-5f;
    float cnt = 0.0f, tot = 0.0f;
    hist10(score, vsat, smin, smax, s_al, 0, lane, cnt, tot);   // (its edges may degenerate: every bin tested on its own)
    ent_s = (float)wave_sum((double)(lane < kEntBins ? entropy_term(cnt, tot, lane) : 0.0f));
    // The 40 control histograms share two fixed edge sets (-w_max .. w_max, -a_max .. a_max; monotone, so "every bin on
    // its own" is "the bin whose lower edge is the last one <= x"): a lane finds the bin of its value with 11 compares
    // and counts it with one LDS atomic; the p*log2(p) terms are then spread over the lanes, one (histogram, bin) each.
    // (bin counts of the w_t and a_t histograms: in the LDS of the hull points, which are no longer needed)
    unsigned (*s_hist)[kEntBins] = reinterpret_cast<unsigned (*)[kEntBins]>(&s_pts[0][0]);
    __syncthreads();
    for (int i = lane; i < 2 * kT * kEntBins; i += kWave) (&s_hist[0][0])[i] = 0u;
    __syncthreads();
    float ew[kEntBins + 1], ea[kEntBins + 1];
#pragma unroll
    for (int k = 0; k <= kEntBins; ++k) {
      ew[k] = (-a.w_max) * (1.0f - s_al[k]) + a.w_max * s_al[k];
      ea[k] = (-a.a_max) * (1.0f - s_al[k]) + a.a_max * s_al[k];
    }
    if (vsat) {
#pragma unroll
      for (int t = 0; t < kT; ++t) {
        const float wv = u[2 * t], av = u[2 * t + 1];
        int cw = 0, ca = 0;
#pragma unroll
        for (int k = 0; k <= kEntBins; ++k) {
          cw += wv >= ew[k] ? 1 : 0;
          ca += av >= ea[k] ? 1 : 0;
        }
        if (cw >= 1 && cw <= kEntBins) atomicAdd(&s_hist[t][cw - 1], 1u);               // ew[cw-1] <= w < ew[cw]
        if (ca >= 1 && ca <= kEntBins) atomicAdd(&s_hist[kT + t][ca - 1], 1u);
      }
    }
    __syncthreads();
    double acc_w = 0.0, acc_a = 0.0;
    for (int i = lane; i < 2 * kT * kEntBins; i += kWave) {
      const int h = i / kEntBins;
      unsigned total = 0u;
#pragma unroll
      for (int k = 0; k < kEntBins; ++k) total += s_hist[h][k];
      const float p = (float)s_hist[h][i % kEntBins] / fmaxf((float)total, 1e-5f);
      const float term = (-p) * log2f(fmaxf(p, 1e-5f));
      if (h < kT) acc_w += (double)term;
      else acc_a += (double)term;
    }
    ent_w = (float)wave_sum(acc_w);
    ent_a = (float)wave_sum(acc_a);
  }

  // ---- occupancy area (compute_area, nusc_api.py:878-891) ----------------------------------------------------------
  // val is valids_rev (bs*3*nt, m) RESHAPED to (bs*3, m, nt): sample j at step t is gated by sample (j*nt + t) % m
  float xlo = INFINITY, xhi = -INFINITY, ylo = INFINITY, yhi = -INFINITY;
  unsigned gate = 0u;
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    const int src = live ? (lane * kT + t) % S : 0;
    const bool g = (vsatmask >> src) & 1ull;
    gate |= g ? (1u << t) : 0u;
    const float gf = g ? 1.0f : 0.0f;
    const float hx = (xr[t] * ct[t] + yr[t] * st[t]) * gf, hy = (-xr[t] * st[t] + yr[t] * ct[t]) * gf;
    if (live) {
      xlo = fminf(xlo, hx);
      xhi = fmaxf(xhi, hx);
      ylo = fminf(ylo, hy);
      yhi = fmaxf(yhi, hy);
    }
  }
  xlo = wave_min(xlo);
  xhi = wave_max(xhi);
  ylo = wave_min(ylo);
  yhi = wave_max(yhi);
  if (xlo == xhi) { xlo -= 0.5f; xhi += 0.5f; }
  if (ylo == yhi) { ylo -= 0.5f; yhi += 0.5f; }
  const float xlen = xhi - xlo, ylen = yhi - ylo;
  const float xstep = xlen / (float)kHist, ystep = ylen / (float)kHist;
  if (live && !(PSTL_DIV_SKIP & 16)) {
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      const float gf = (gate >> t) & 1u ? 1.0f : 0.0f;
      const float hx = (xr[t] * ct[t] + yr[t] * st[t]) * gf, hy = (-xr[t] * st[t] + yr[t] * ct[t]) * gf;
      const int bin = hist_bin(hx, xlo, xhi, xstep, xlen) * kHist + hist_bin(hy, ylo, yhi, ystep, ylen);
      atomicOr(&s_occ[bin >> 5], 1u << (bin & 31));
    }
  }
  __syncthreads();
  int occ = 0;
  for (int i = lane; i < kHistWords; i += kWave) occ += __popc(s_occ[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) occ += __shfl_xor(occ, o);
  const float area = (((float)occ / (float)(kHist * kHist)) * xlen) * ylen;

  if (lane == 0) {
    double* o = a.per_mode + (long)blockIdx.x * 8;
    o[0] = std_acc;
    o[1] = vol;
    o[2] = (double)ent_s;
    o[3] = (double)ent_w;
    o[4] = (double)ent_a;
    o[5] = (double)area;
    o[6] = (double)n_sat;
    o[7] = mode_valid ? 1.0 : 0.0;
  }
}

// totals[0] sum std over valid (scene,mode), [1] sum vol over valid, [2] #valid (scene,mode), [3] sum ent_s,
// [4] sum_t ent_w, [5] sum_t ent_a, [6] sum area, [7] #(scene,mode), [8] sum ade, [9] sum fde, [10] #scenes.
// One workgroup, fixed summation order => bit-reproducible.
__global__ __launch_bounds__(256) void k_diversity_totals(int bs, const double* per_mode, const float* per_scene,
                                                          double* totals) {
  __shared__ double sm[256][11];
  double acc[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) acc[k] = 0.0;
  for (int i = threadIdx.x; i < bs * 3; i += 256) {
    const double* p = per_mode + (long)i * 8;
    const double val = p[7];
    acc[0] += p[0] * val;
    acc[1] += p[1] * val;
    acc[2] += val;
    acc[3] += p[2];
    acc[4] += p[3];
    acc[5] += p[4];
    acc[6] += p[5];
    acc[7] += 1.0;
  }
  for (int i = threadIdx.x; i < bs; i += 256) {
    acc[8] += (double)per_scene[i * 2];
    acc[9] += (double)per_scene[i * 2 + 1];
    acc[10] += 1.0;
  }
#pragma unroll
  for (int k = 0; k < 11; ++k) sm[threadIdx.x][k] = acc[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 11; ++k) sm[threadIdx.x][k] += sm[threadIdx.x + o][k];
    __syncthreads();
  }
  if (threadIdx.x < 11) totals[threadIdx.x] = sm[0][threadIdx.x];
  if (threadIdx.x == 11) totals[11] = 0.0;
}

}  // namespace
}  // namespace pstl

using namespace pstl;

static __global__ void k_fill_words(int n, unsigned v, unsigned* p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

static long div_cu_count() { return device_cus(); }   // (of the current device: pstl_common.hpp)

extern "C" int pstl_diversity(const pstl_cfg* cfg, const float* s0, const float* gt_traj, int gt_stride,
                              const float* controls, const float* scores, const float* valid, const float* alphas,
                              double* per_mode, float* per_scene, double* totals, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!s0 || !gt_traj || !controls || !scores || !valid || !alphas || !per_mode || !per_scene || gt_stride < 4)
    return PSTL_ERR_ARG;
  if (cfg->rows_per_scene != 3 * cfg->S || cfg->S > kWave) return PSTL_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  // (+inf for the atomicMin of ADE / FDE -- by a kernel, not a memset node: memset nodes did not replay under HIP-graph capture)
  hipLaunchKernelGGL(k_fill_words, dim3((unsigned)((cfg->bs * 2 + 255) / 256)), dim3(256), 0, st, cfg->bs * 2, 0x7f800000u,
                     reinterpret_cast<unsigned*>(per_scene));
  DivArgs a;
  a.bs = cfg->bs;
  a.S = cfg->S;
  a.gt_stride = gt_stride;
  a.dt = cfg->dt;
  a.w_max = cfg->w_max;
  a.a_max = cfg->a_max;
  a.s0 = s0;
  a.gt = gt_traj;
  a.controls = controls;
  a.scores = scores;
  a.valid = valid;
  a.alphas = alphas;
  a.per_mode = per_mode;
  a.per_scene = per_scene;
  // few (scene, mode) pairs: ten waves each (one workgroup per CU at ~165 registers: one round of them, 99 us against the 130 us
  // of the single-wave chain; with two rounds -- 24 576 rows -- the single waves win)
  if ((long)cfg->bs * 3 <= div_cu_count())
    hipLaunchKernelGGL(k_diversity<true>, dim3((unsigned)cfg->bs * 3), dim3(kDivWaves * kWave), 0, st, a);
  else
    hipLaunchKernelGGL(k_diversity<false>, dim3((unsigned)cfg->bs * 3), dim3(kWave), 0, st, a);
  if (totals)
    hipLaunchKernelGGL(k_diversity_totals, dim3(1), dim3(256), 0, st, cfg->bs, per_mode, per_scene, totals);
  return launch_status();
}
