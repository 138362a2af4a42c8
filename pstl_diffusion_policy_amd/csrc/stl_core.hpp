// stl_core.hpp -- per-row math of the STL robustness path: unicycle rollout, point-to-lane distance, car-to-car
// clearance, the three STL formulas (soft-min/soft-max temporal operators) and their adjoint.
//
// One row (one sampled trajectory) is evaluated by one lane.  The functions are __host__ __device__ so that the very
// same arithmetic can be compiled with g++ for the CPU-side unit tests and sanitizer builds (tests/ only -- the
// product path always runs the HIP build).  This translation unit must be compiled with -ffp-contract=off: the
// reference evaluates these expressions as separate, individually rounded torch ops, and two of them
// (the signed triangle area in lane_eval, the circle-centre distances) cancel catastrophically in world coordinates.
//
// Reference restated here (file:line in /root/reference):
//   generate_trajs / dynamics ........ nusc_train.py:29-49
//   compute_t2l_dist ................. nusc_api.py:685-739 ("efficient", inline=False, clip=False, with_angle=True)
//   get_anchor_point / dist_* ........ utils.py:465-526 ; compute_shortest_dist_refined nusc_train.py:142-148
//   formulas ......................... nusc_train.py:95-140 ; operators stl_d_lib.py:6-26,87-112,144-169
//   mode select ...................... nusc_train.py:150-151,318-323
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PSTL_HD __host__ __device__ __forceinline__
#define PSTL_UNROLL _Pragma("unroll")
#define PSTL_NOUNROLL _Pragma("nounroll")
#else
#define PSTL_HD inline
#define PSTL_UNROLL
#define PSTL_NOUNROLL
#endif

namespace pstl {

constexpr int kT = 20;        // horizon nt
constexpr int kNseg = 15;     // lane waypoints
constexpr int kFwin = 10;     // Eventually(0, nt//2)
constexpr int kNeiPrep = 12;  // floats per prepared (neighbour, t)
constexpr int kScratchFloats = 4 * kT + 2 * kFwin;  // states + the two stored suffix tables of the backward pass

struct alignas(16) f4 {
  float x, y, z, w;
};

// per-lane scratch: element i of this lane lives at p[i * stride]  (LDS: stride = blockDim.x, host: stride = 1)
struct Scratch {
  float* p;
  int stride;
  PSTL_HD float& at(int i) const { return p[i * stride]; }
};

struct StlEnv {
  float tau;
  float dt;
  float eoff[4];  // ego circle centres along the body axis
  float er;       // ego circle radius
};

struct StlRow {
  float vmin, vmax, dmin, dmax, dsafe, thmax;
  int mode;  // 0 keep lane, 1 left, 2 right, 3 outlier (score == 1)
};

// Circle row of a car (utils.py:474-486 with num_L = 4, num_W = 1): radius and the 4 centre offsets along the body axis.
PSTL_HD void circle_row(float L, float W, float* off, float& r) {
  const float rl = L / 4.0f / 2.0f;
  const float rw = W / 1.0f / 2.0f;
  r = fminf(fmaxf(rl, rw), W / 2.0f);
  const float x2 = -L / 2.0f, x1 = L / 2.0f;
  // torch.linspace(0, 1, 4) in float32
  const float a[4] = {0.0f, 0x1.555556p-2f, 0x1.555554p-1f, 1.0f};
  for (int i = 0; i < 4; ++i) off[i] = (x2 + r) * (1.0f - a[i]) + (x1 - r) * a[i];
}

// Prepared neighbour entry for one (neighbour, t): in = valid,x,y,th,v,L,W ; out = valid, r, cx[4], cy[4], 0, 0.
// (get_anchor_point, utils.py:483-494; the lateral offset -W/2 + r is exactly 0 for num_W = 1.)
PSTL_HD void prep_neighbor(const float* in, float* out) {
  float off[4], r;
  circle_row(in[5], in[6], off, r);
  const float c = cosf(in[3]), s = sinf(in[3]);
  out[0] = in[0];
  out[1] = r;
  for (int i = 0; i < 4; ++i) {
    out[2 + i] = off[i] * c + in[1];
    out[6 + i] = off[i] * s + in[2];
  }
  out[10] = 0.0f;
  out[11] = 0.0f;
}

PSTL_HD StlEnv make_env(float tau, float dt, float ego_L, float ego_W) {
  StlEnv e;
  e.tau = tau;
  e.dt = dt;
  circle_row(ego_L, ego_W, e.eoff, e.er);
  return e;
}

// ---------------------------------------------------------------------------------------------------------------
// online log-sum-exp:  value() == logsumexp of everything add()ed so far (same maths as torch.logsumexp, which
// subtracts the maximum; here the maximum is tracked incrementally)
// ---------------------------------------------------------------------------------------------------------------
struct Lse {
  float m, s;
  PSTL_HD void init() {
    m = -INFINITY;
    s = 0.0f;
  }
  PSTL_HD void add(float a) {
    const float hi = fmaxf(a, m);
    const float e = expf(fminf(a, m) - hi);  // exp(-|a-m|); exp(-inf) = 0 on the first add
    s = (a > m) ? (s * e + 1.0f) : (s + e);
    m = hi;
  }
  PSTL_HD float value() const { return logf(s) + m; }
};

PSTL_HD float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  return logf(expf(a - m) + expf(b - m)) + m;
}

// ---------------------------------------------------------------------------------------------------------------
// A6  unicycle states 0..T-1 into scratch[4*t + {0,1,2,3}]
// ---------------------------------------------------------------------------------------------------------------
PSTL_HD void rollout_states(const float* s0, const float* u, float wscale, float ascale, float dt, Scratch st) {
  float x = s0[0], y = s0[1], th = s0[2], v = s0[3];
  PSTL_NOUNROLL
  for (int t = 0; t < kT; ++t) {
    st.at(4 * t + 0) = x;
    st.at(4 * t + 1) = y;
    st.at(4 * t + 2) = th;
    st.at(4 * t + 3) = v;
    const float w = u[2 * t] * wscale;
    const float a = u[2 * t + 1] * ascale;
    const float dx = v * cosf(th);
    const float dy = v * sinf(th);
    x = x + dx * dt;
    y = y + dy * dt;
    th = th + w * dt;
    v = v + a * dt;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// A9  signed lateral distance + heading error to the closest segment pair of a 15-waypoint lane
// ---------------------------------------------------------------------------------------------------------------
struct LaneHit {
  float d, th;
  float dd_dx, dd_dy, dth_dth;  // partials (only meaningful when requested)
};

template <bool GRAD>
PSTL_HD void lane_eval(const f4* lane, float px, float py, float pth, LaneHit& h) {
  f4 p = lane[0];
  float ex = px - p.x, ey = py - p.y;
  float prev = sqrtf(ex * ex + ey * ey);
  float best = INFINITY;
  f4 p2 = p, p3 = p;
  f4 q = p;
  PSTL_UNROLL
  for (int j = 0; j < kNseg - 1; ++j) {
    const f4 n = lane[j + 1];
    ex = px - n.x;
    ey = py - n.y;
    const float cur = sqrtf(ex * ex + ey * ey);
    const float s = prev + cur;
    if (s < best) {  // strict: lowest index wins ties, like torch.argmin
      best = s;
      p2 = q;
      p3 = n;
    }
    prev = cur;
    q = n;
  }
  const float area = px * (p2.y - p3.y) + p2.x * (p3.y - py) + p3.x * (py - p2.y);
  const float sx = p2.x - p3.x, sy = p2.y - p3.y;
  const float bl = sqrtf(sx * sx + sy * sy);
  const float qx = px - p2.x, qy = py - p2.y;
  const float q2 = qx * qx + qy * qy;
  const float l2 = sqrtf(fmaxf(q2, 1e-3f));
  const bool normal = (bl != 0.0f);
  const float cbl = fmaxf(bl, 1e-7f);
  h.d = normal ? area / cbl : l2;
  const float du = p2.z - pth;
  h.th = 1.0f - cosf(du);
  if (GRAD) {
    if (normal) {
      h.dd_dx = (p2.y - p3.y) / cbl;
      h.dd_dy = (p3.x - p2.x) / cbl;
    } else if (q2 >= 1e-3f) {
      h.dd_dx = qx / l2;
      h.dd_dy = qy / l2;
    } else {
      h.dd_dx = 0.0f;
      h.dd_dy = 0.0f;
    }
    h.dth_dth = -sinf(du);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// A10 clearance to the K neighbours at time t.  nei points at the prepared table of this scene: (K, T, 12) floats
//     = valid, r, cx[4], cy[4], pad, pad.
// ---------------------------------------------------------------------------------------------------------------
struct ClearHit {
  float dn;
  float d_dx, d_dy, d_dth;
};

template <bool GRAD>
PSTL_HD void clearance_eval(const StlEnv& env, const float* nei, int K, int t, float x, float y, float th, ClearHit& h) {
  const float c = cosf(th), s = sinf(th);
  float ex[4], ey[4];
  PSTL_UNROLL
  for (int i = 0; i < 4; ++i) {
    ex[i] = env.eoff[i] * c + x;
    ey[i] = env.eoff[i] * s + y;
  }
  float best = INFINITY;
  float gx = 0.0f, gy = 0.0f, gth = 0.0f;
  PSTL_NOUNROLL
  for (int k = 0; k < K; ++k) {
    const f4* e = reinterpret_cast<const f4*>(nei + (size_t)(k * kT + t) * kNeiPrep);
    const f4 a = e[0], b = e[1], cc = e[2];
    const float valid = a.x, r = a.y;
    const float nx[4] = {a.z, a.w, b.x, b.y};
    const float ny[4] = {b.z, b.w, cc.x, cc.y};
    float q = INFINITY;
    float bdx = 0.0f, bdy = 0.0f, boff = 0.0f;  // the closest circle pair (first one on ties)
    PSTL_UNROLL
    for (int i = 0; i < 4; ++i) {
      PSTL_UNROLL
      for (int j = 0; j < 4; ++j) {
        const float dx = ex[i] - nx[j], dy = ey[i] - ny[j];
        const float qq = dx * dx + dy * dy;
        if (qq < q) {
          q = qq;
          if (GRAD) {
            bdx = dx;
            bdy = dy;
            boff = env.eoff[i];
          }
        }
      }
    }
    const float dist = sqrtf(q);  // sqrt is monotone: min over sqrt == sqrt of min
    const float car = dist - env.er - r;
    const float clipped = fminf(fmaxf(car, -5.0f), 20.0f);
    const float val = clipped * valid + (1.0f - valid) * 100.0f;
    if (val < best) {  // lowest neighbour index wins ties (torch.min)
      best = val;
      if (GRAD) {
        const bool pass = (car >= -5.0f) && (car <= 20.0f) && (dist > 0.0f);
        const float g = pass ? valid / dist : 0.0f;
        const float ddx = bdx * g, ddy = bdy * g;
        gx = ddx;
        gy = ddy;
        gth = ddx * (-boff * s) + ddy * (boff * c);
      }
    }
  }
  h.dn = best;
  if (GRAD) {
    h.d_dx = gx;
    h.d_dy = gy;
    h.d_dth = gth;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// A8  formulas.  Time runs backwards (t = T-1 ... 0) so that every suffix soft-min G s[t] = softmin(s[t:T]) is a
//     running log-sum-exp; F10 G s = softmax over t < 10 of those suffix values is a second running log-sum-exp.
// ---------------------------------------------------------------------------------------------------------------
struct LaneAcc {
  Lse g1, g2, g3;  // keep-lane terms: G(d - dmin), G(dmax - d), G((thmax - th)/thmax)
  Lse gb;          // suffix soft-min of the band term softmin2(d - dmin, dmax - d)
  Lse fb, ft;      // F10 over G(band)[t] and over G(th-term)[t]
  PSTL_HD void init() {
    g1.init();
    g2.init();
    g3.init();
    gb.init();
    fb.init();
    ft.init();
  }
  // KEEP: accumulate the keep-lane family; REACH: the reach family (shares g3)
  template <bool KEEP, bool REACH>
  PSTL_HD void step(const StlRow& r, float tau, int t, float d, float th, float* lb_store, float* lt_store) {
    const float s1 = d - r.dmin;
    const float s2 = -d + r.dmax;
    const float s3 = (r.thmax - th) / r.thmax;
    g3.add(-s3 * tau);
    if (KEEP) {
      g1.add(-s1 * tau);
      g2.add(-s2 * tau);
    }
    if (REACH) {
      const float band = -(lse2(-s1 * tau, -s2 * tau) / tau);
      gb.add(-band * tau);
      if (t < kFwin) {
        const float lb = gb.value(), lt = g3.value();
        fb.add(-(lb / tau) * tau);
        ft.add(-(lt / tau) * tau);
        if (lb_store) {
          *lb_store = lb;
          *lt_store = lt;
        }
      }
    }
  }
};

PSTL_HD float conj6(const float* v, int n, float tau) {  // softmin over n <= 6 values, torch.logsumexp style
  float a[6];
  float m = -INFINITY;
  PSTL_UNROLL
  for (int i = 0; i < 6; ++i) {
    a[i] = i < n ? -v[i] * tau : -INFINITY;
    m = fmaxf(m, a[i]);
  }
  float s = 0.0f;
  PSTL_UNROLL
  for (int i = 0; i < 6; ++i) s += (i < n) ? expf(a[i] - m) : 0.0f;
  return -((logf(s) + m) / tau);
}

// Evaluates the formulas of one row whose states are already in scratch.
//   ALL3 = true : all three formulas -> out3[0..2]; returns the mode-selected score
//   ALL3 = false: only the formula of r.mode
template <bool ALL3>
PSTL_HD float stl_eval(const StlEnv& env, const StlRow& r, const f4* lanes, const float* nei, int K, Scratch st,
                       float* out3) {
  const float tau = env.tau;
  Lse gv1, gv2, gsafe;
  gv1.init();
  gv2.init();
  gsafe.init();
  LaneAcc L0, L1, L2;
  L0.init();
  L1.init();
  L2.init();
  const int mode = r.mode;
  const f4* sel_lane = lanes + (mode < 3 ? mode : 0) * kNseg;
  PSTL_NOUNROLL
  for (int t = kT - 1; t >= 0; --t) {
    const float x = st.at(4 * t), y = st.at(4 * t + 1), th = st.at(4 * t + 2), v = st.at(4 * t + 3);
    gv1.add(-(v - r.vmin) * tau);
    gv2.add(-(-v + r.vmax) * tau);
    ClearHit ch;
    clearance_eval<false>(env, nei, K, t, x, y, th, ch);
    gsafe.add(-(ch.dn - r.dsafe) * tau);
    LaneHit h;
    if (ALL3) {
      lane_eval<false>(lanes, x, y, th, h);
      L0.step<true, false>(r, tau, t, h.d, h.th, nullptr, nullptr);
      lane_eval<false>(lanes + kNseg, x, y, th, h);
      L1.step<false, true>(r, tau, t, h.d, h.th, nullptr, nullptr);
      lane_eval<false>(lanes + 2 * kNseg, x, y, th, h);
      L2.step<false, true>(r, tau, t, h.d, h.th, nullptr, nullptr);
    } else {
      lane_eval<false>(sel_lane, x, y, th, h);
      L0.step<true, true>(r, tau, t, h.d, h.th, nullptr, nullptr);
    }
  }
  const float Vv1 = -(gv1.value() / tau), Vv2 = -(gv2.value() / tau), Vs = -(gsafe.value() / tau);
  float sc[3] = {0.0f, 0.0f, 0.0f};
  if (ALL3 || mode == 0) {
    const float v[6] = {Vv1, Vv2, -(L0.g1.value() / tau), -(L0.g2.value() / tau), -(L0.g3.value() / tau), Vs};
    sc[0] = conj6(v, 6, tau);
  }
  if (ALL3) {
    const float v1[5] = {Vv1, Vv2, L1.fb.value() / tau, L1.ft.value() / tau, Vs};
    sc[1] = conj6(v1, 5, tau);
    const float v2[5] = {Vv1, Vv2, L2.fb.value() / tau, L2.ft.value() / tau, Vs};
    sc[2] = conj6(v2, 5, tau);
  } else if (mode == 1 || mode == 2) {
    const float v1[5] = {Vv1, Vv2, L0.fb.value() / tau, L0.ft.value() / tau, Vs};
    sc[mode] = conj6(v1, 5, tau);
  }
  if (ALL3 && out3) {
    out3[0] = sc[0];
    out3[1] = sc[1];
    out3[2] = sc[2];
  }
  // get_stl_scores (nusc_train.py:150-151): masked sum of the three formulas plus the constant 1 for outliers
  if (ALL3) {
    return sc[0] * (mode == 0 ? 1.0f : 0.0f) + sc[1] * (mode == 1 ? 1.0f : 0.0f) + sc[2] * (mode == 2 ? 1.0f : 0.0f) +
           1.0f * (mode == 3 ? 1.0f : 0.0f);
  }
  return mode == 3 ? 1.0f : sc[mode];
}

// ---------------------------------------------------------------------------------------------------------------
// Forward + adjoint of one row: returns the score and calls emit(t, gw, ga) once for every t in [0,T) with
// (gw, ga) = dscore_fn(score) * d score / d (u[2t], u[2t+1])  (u = the 40 control values that rollout_states
// multiplied by wscale/ascale).  Only the formula of r.mode carries gradient (the others are multiplied by a 0 mask
// in the reference).  Scratch must hold kScratchFloats floats per lane.
// ---------------------------------------------------------------------------------------------------------------
template <class DScoreFn, class EmitFn>
PSTL_HD float stl_eval_grad(const StlEnv& env, const StlRow& r, const f4* lanes, const float* nei, int K, Scratch st,
                            float wscale, float ascale, DScoreFn dscore_fn, EmitFn emit) {
  const float tau = env.tau;
  const int mode = r.mode;
  if (mode >= 3) {
    PSTL_NOUNROLL
    for (int t = 0; t < kT; ++t) emit(t, 0.0f, 0.0f);
    return 1.0f;
  }
  const f4* lane = lanes + mode * kNseg;
  const int LB = 4 * kT, LT = 4 * kT + kFwin;
  // ---- pass 2: values -----------------------------------------------------------------------------------------
  Lse gv1, gv2, gsafe;
  gv1.init();
  gv2.init();
  gsafe.init();
  LaneAcc A;
  A.init();
  PSTL_NOUNROLL
  for (int t = kT - 1; t >= 0; --t) {
    const float x = st.at(4 * t), y = st.at(4 * t + 1), th = st.at(4 * t + 2), v = st.at(4 * t + 3);
    gv1.add(-(v - r.vmin) * tau);
    gv2.add(-(-v + r.vmax) * tau);
    ClearHit ch;
    clearance_eval<false>(env, nei, K, t, x, y, th, ch);
    gsafe.add(-(ch.dn - r.dsafe) * tau);
    LaneHit h;
    lane_eval<false>(lane, x, y, th, h);
    float lb = 0.0f, lt = 0.0f;
    A.step<true, true>(r, tau, t, h.d, h.th, &lb, &lt);
    if (t < kFwin) {
      st.at(LB + t) = lb;
      st.at(LT + t) = lt;
    }
  }
  const float Lv1 = gv1.value(), Lv2 = gv2.value(), Ls = gsafe.value();
  const float Vv1 = -(Lv1 / tau), Vv2 = -(Lv2 / tau), Vs = -(Ls / tau);
  const float L1 = A.g1.value(), L2 = A.g2.value(), L3 = A.g3.value();
  const float Lfb = A.fb.value(), Lft = A.ft.value();
  float V[6];
  int n;
  if (mode == 0) {
    V[0] = Vv1, V[1] = Vv2, V[2] = -(L1 / tau), V[3] = -(L2 / tau), V[4] = -(L3 / tau), V[5] = Vs;
    n = 6;
  } else {
    V[0] = Vv1, V[1] = Vv2, V[2] = Lfb / tau, V[3] = Lft / tau, V[4] = Vs;
    n = 5;
  }
  const float score = conj6(V, n, tau);
  const float Lout = -score * tau;  // logsumexp of (-V_i tau)
  const float dscore_in = dscore_fn(score);
  float om[6];
  PSTL_UNROLL
  for (int i = 0; i < 6; ++i) om[i] = i < n ? expf(-V[i] * tau - Lout) * dscore_in : 0.0f;  // d score / d V_i
  const float o_v1 = om[0], o_v2 = om[1], o_s = (mode == 0) ? om[5] : om[4];
  emit(kT - 1, 0.0f, 0.0f);  // the last control never reaches a scored state
  // ---- pass 3: adjoint, backwards in time ---------------------------------------------------------------------
  float lx = 0.0f, ly = 0.0f, lth = 0.0f, lv = 0.0f;  // lambda_{t+1}
  const float dt = env.dt;
  PSTL_NOUNROLL
  for (int t = kT - 1; t >= 1; --t) {
    const float x = st.at(4 * t), y = st.at(4 * t + 1), th = st.at(4 * t + 2), v = st.at(4 * t + 3);
    // direct partials of the score w.r.t. state t
    float gx, gy, gth, gv;
    gv = o_v1 * expf(-(v - r.vmin) * tau - Lv1) - o_v2 * expf(-(-v + r.vmax) * tau - Lv2);
    ClearHit ch;
    clearance_eval<true>(env, nei, K, t, x, y, th, ch);
    const float gs = o_s * expf(-(ch.dn - r.dsafe) * tau - Ls);
    gx = gs * ch.d_dx;
    gy = gs * ch.d_dy;
    gth = gs * ch.d_dth;
    LaneHit h;
    lane_eval<true>(lane, x, y, th, h);
    const float s1 = h.d - r.dmin, s2 = -h.d + r.dmax, s3 = (r.thmax - h.th) / r.thmax;
    float gd, gsth;  // d score / d d_t , d score / d s3_t
    if (mode == 0) {
      gd = om[2] * expf(-s1 * tau - L1) - om[3] * expf(-s2 * tau - L2);
      gsth = om[4] * expf(-s3 * tau - L3);
    } else {
      const float a1 = -s1 * tau, a2 = -s2 * tau;
      const float lp = lse2(a1, a2);
      const float band = -(lp / tau);
      const float ab = -band * tau, a3 = -s3 * tau;
      float wb = 0.0f, wt = 0.0f;
      const int tmax = t < kFwin - 1 ? t : kFwin - 1;
      PSTL_NOUNROLL
      for (int k = 0; k <= tmax; ++k) {
        const float lb = st.at(LB + k), lt = st.at(LT + k);
        wb += expf(-(lb / tau) * tau - Lfb) * expf(ab - lb);
        wt += expf(-(lt / tau) * tau - Lft) * expf(a3 - lt);
      }
      gd = om[2] * wb * (expf(a1 - lp) - expf(a2 - lp));
      gsth = om[3] * wt;
    }
    gx += gd * h.dd_dx;
    gy += gd * h.dd_dy;
    gth += gsth * (-1.0f / r.thmax) * h.dth_dth;
    // lambda_t = direct_t + J_t^T lambda_{t+1}
    const float c = cosf(th), s = sinf(th);
    const float nlth = gth + lth + lx * (-(v * s) * dt) + ly * ((v * c) * dt);
    const float nlv = gv + lv + lx * (c * dt) + ly * (s * dt);
    lx = gx + lx;
    ly = gy + ly;
    lth = nlth;
    lv = nlv;
    // state_t = f(state_{t-1}, u_{t-1}):  th_t = th_{t-1} + w dt ; v_t = v_{t-1} + a dt
    emit(t - 1, lth * dt * wscale, lv * dt * ascale);
  }
  return score;
}

}  // namespace pstl
