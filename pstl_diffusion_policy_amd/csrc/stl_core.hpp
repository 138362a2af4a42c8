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

// exp/log of the soft-min/soft-max operators.  On the device these are the hardware v_exp_f32 / v_log_f32 forms
// (about 1 ulp; arguments here are <= 0 and results are divided by tau = 100 afterwards, so a robustness score moves
// by ~1e-8): they are ~5x cheaper than the correctly rounded library calls and there are ~300 of them per row.
// Trigonometric functions, divisions and square roots stay exact (positions integrate over 20 steps).
#if defined(__HIP_DEVICE_COMPILE__)
#define PSTL_EXP(x) __expf(x)
#define PSTL_LOG(x) __logf(x)
// 1-ulp hardware square root: ONLY for ranking the 14 waypoint pairs of a lane (the winner's distance itself is
// computed with exact arithmetic afterwards); a different winner needs two pair sums within 1 ulp of each other
#define PSTL_SQRT_RANK(x) __builtin_amdgcn_sqrtf(x)
#define PSTL_SINCOS(x, s, c) sincosf((x), (s), (c))
// Hardware forms (v_rcp_f32, v_sqrt_f32: 1 ulp) for quantities that only ever SCALE a gradient or an optimiser step -- the
// adjoint's partial derivatives, Adam's m / (sqrt(v) + eps) -- never a score, a satisfaction mask, a candidate choice or the
// argument of a soft-min exponential (those are multiplied by tau = 100: see the note at kGeoSlots): the forward sweeps do not
// use them.  The reference's autograd gradients are matched to rtol 5e-3 (tests); an IEEE division or square root is ~10-12
// instructions here, the hardware form one (quarter rate).
#define PSTL_RCP_ADJ(x) __builtin_amdgcn_rcpf(x)
#define PSTL_SQRT_ADJ(x) __builtin_amdgcn_sqrtf(x)
#else
#define PSTL_EXP(x) expf(x)
#define PSTL_LOG(x) logf(x)
#define PSTL_SQRT_RANK(x) sqrtf(x)
#define PSTL_SINCOS(x, s, c) (*(s) = sinf(x), *(c) = cosf(x))
#define PSTL_RCP_ADJ(x) (1.0f / (x))
#define PSTL_SQRT_ADJ(x) sqrtf(x)
#endif

namespace pstl {

// -DPSTL_STL_STAMP (diagnostic builds only; tools/dbg/stl_stamps.sh): PSTL_ST(k) books the shader cycles since the wave's previous
// stamp to section k (s_memtime; lane 0 adds to a per-workgroup table in LDS, the kernel's tail adds that to a device table which
// pstl_debug_stl_stamps hands out).  The one-wave-per-workgroup kernels only.  In every other build PSTL_ST(k) is nothing.
#if defined(PSTL_STL_STAMP) && defined(__HIPCC__)
constexpr int kStampSlots = 32;
__device__ unsigned long long pstl_st_global[kStampSlots];
#endif
#if defined(PSTL_STL_STAMP) && defined(__HIP_DEVICE_COMPILE__)
__shared__ unsigned pstl_st_acc[kStampSlots];
__shared__ unsigned pstl_st_last;
__device__ __forceinline__ void pstl_st_begin() {
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x < kStampSlots) pstl_st_acc[threadIdx.x] = 0u;
  const unsigned lo = (unsigned)__builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) pstl_st_last = lo;
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pstl_st_mark(int k) {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned lo = (unsigned)__builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    pstl_st_acc[k] += lo - pstl_st_last;
    pstl_st_last = lo;
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void pstl_st_end() {
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x < kStampSlots && pstl_st_acc[threadIdx.x]) atomicAdd(&pstl_st_global[threadIdx.x], (unsigned long long)pstl_st_acc[threadIdx.x]);
  if (threadIdx.x == 0) atomicAdd(&pstl_st_global[kStampSlots - 1], 1ull);   // waves
}
#define PSTL_ST(k) pstl_st_mark(k)
// (a value the section before the stamp produces: pinned in front of it, or the optimiser sinks the section's work to where the
// value is used -- behind the stamp -- and books it to the wrong section)
#define PSTL_ST_KEEP(x) asm volatile("" : "+v"(x))
#define PSTL_ST_BEGIN() pstl_st_begin()
#define PSTL_ST_END() pstl_st_end()
#else
#define PSTL_ST(k) ((void)0)
#define PSTL_ST_KEEP(x) ((void)0)
#define PSTL_ST_BEGIN() ((void)0)
#define PSTL_ST_END() ((void)0)
#endif
// sections (tools/dbg/stl_stamps.py prints them under these names)
enum {
  ST_PROLOGUE = 0, ST_DYN = 1, ST_CLEAR = 2, ST_LANE_RANK = 3, ST_LANE_TAIL = 4, ST_LSE = 5, ST_FWD_FINISH = 6, ST_ADJ_HEAD = 7,
  ST_ADJ_REDERIVE = 8, ST_ADJ_CLEAR = 9, ST_ADJ_LANE = 10, ST_ADJ_WEIGHTS = 11, ST_ADJ_COSTATE = 12, ST_EMIT = 13, ST_ZERO_EMIT = 14,
  ST_OTHER = 15, ST_SELECT = 16
};

constexpr int kT = 20;        // horizon nt
constexpr int kNseg = 15;     // lane waypoints
constexpr int kFwin = 10;     // Eventually(0, nt//2)
constexpr int kNeiPrep = 12;  // floats per prepared (neighbour, t)
// per-lane scratch floats: forward needs the first 10 values of the two "reach" signals per evaluated side lane;
// the adjoint additionally keeps the state (x, y, heading, speed) at every 4th step (see stl_eval_grad)
constexpr int kScratchFwd = 2 * kFwin;       // selected formula only
constexpr int kScratchFwd3 = 4 * kFwin;      // all three formulas (left and right lane)
constexpr int kCkStride = 4;                     // state checkpoints every 4 steps (the adjoint re-derives blocks of 4 states)
constexpr int kCk = kT / kCkStride;               // 5 checkpoints of (x, y, th, v)
constexpr int kScratchGrad = 4 * kCk + 2 * kFwin;
constexpr int kScratchGradPre = 2 * kFwin;        // ... with the precomputed geometry (PRE): no checkpoints

struct alignas(16) f4 {
  float x, y, z, w;
};
// a pair of floats the device compiler keeps in two adjacent registers and works on with the packed-fp32 instructions
#if defined(__HIPCC__)
typedef float v2f __attribute__((ext_vector_type(2)));
#else
struct v2f {
  float x, y;
};
inline v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
inline v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
inline v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
#endif

// per-lane scratch: element i of this lane lives at p[i * stride]  (LDS: stride = blockDim.x, host: stride = 1)
struct Scratch {
  float* p;
  int stride;
  PSTL_HD float& at(int i) const { return p[i * stride]; }
};

// Winners of the forward sweep's hard minima, one entry per time step, for the adjoint -- which would otherwise rank the 14
// segment pairs of the lane and the 16 circle pairs of every neighbour a second time at the very same state:
//   ln: 4 bits per step = index of the winning waypoint pair (compute_t2l_dist's argmin);
//   cl: 8 bits per step = neighbour (4 bits; 15 = no neighbour, the constant 100) | circle pair 4 i + j (4 bits).
// Eight registers per lane.  Neighbour indices need K <= kRecMaxK; beyond that the adjoint ranks again.
constexpr int kRecMaxK = 14;
struct Rec {
  unsigned ln[3], cl[5];
};
PSTL_HD void rec_clear(Rec& r) {
  r.ln[0] = r.ln[1] = r.ln[2] = 0u;
  r.cl[0] = r.cl[1] = r.cl[2] = r.cl[3] = r.cl[4] = 0u;
}
// The record is two shift registers: every step's entry enters at the top and everything moves down by one entry (funnel
// shifts: 3 + 5 instructions and two shift-ors per step; the "or into word t / 8" form it replaces picked its word with eight
// selects).  rec_put must be called for t = 0 ... kT-1 in that order, exactly once each (stl_eval_rec does); afterwards step
// t's clearance byte sits at bit 8 t of the 160-bit cl, its lane nibble at bit 16 + 4 t of the 96-bit ln.
PSTL_HD unsigned funnel_shr(unsigned hi, unsigned lo, int sh) { return (lo >> sh) | (hi << (32 - sh)); }
PSTL_HD void rec_put(Rec& r, int t, unsigned seg, unsigned win) {
  (void)t;
  r.ln[0] = funnel_shr(r.ln[1], r.ln[0], 4);
  r.ln[1] = funnel_shr(r.ln[2], r.ln[1], 4);
  r.ln[2] = (r.ln[2] >> 4) | (seg << 28);
  r.cl[0] = funnel_shr(r.cl[1], r.cl[0], 8);
  r.cl[1] = funnel_shr(r.cl[2], r.cl[1], 8);
  r.cl[2] = funnel_shr(r.cl[3], r.cl[2], 8);
  r.cl[3] = funnel_shr(r.cl[4], r.cl[3], 8);
  r.cl[4] = (r.cl[4] >> 8) | (win << 24);
}
PSTL_HD unsigned rec_seg(const Rec& r, int t) {
  static_assert(kT == 20, "the lane nibbles of 20 steps end at bit 16 of the 96-bit register");
  const int bit = 16 + 4 * t, wa = bit >> 5;
  const unsigned x = wa == 0 ? r.ln[0] : (wa == 1 ? r.ln[1] : r.ln[2]);
  return (x >> (bit & 31)) & 15u;
}
PSTL_HD unsigned rec_win(const Rec& r, int t) {
  const int wb = t >> 2;
  const unsigned x = wb == 0 ? r.cl[0] : (wb == 1 ? r.cl[1] : (wb == 2 ? r.cl[2] : (wb == 3 ? r.cl[3] : r.cl[4])));
  return (x >> (8 * (t & 3))) & 255u;
}

struct StlEnv {
  float tau;
  float dt;
  float eoff[4];  // ego circle centres along the body axis
  float er;       // ego circle radius
};

struct StlRow {
  float vmin, vmax, dmin, dmax, dsafe, thmax;
  int mode;  // 0 keep lane, 1 left, 2 right, 3 outlier (score == 1)
  // --norm_stl (nusc_train.py:88-91): divisors of the speed / lane-distance / clearance predicates (set by norm_factors;
  // only read by the NORM instantiations)
  float vf, df, sf;
};
// v_factor = clip(vmax - vmin, 0.3), d_factor = clip((dmax - dmin) * 5, 0.3), safe_factor = clip(dsafe, 0.3)
PSTL_HD void norm_factors(StlRow& r) {
  r.vf = fmaxf(r.vmax - r.vmin, 0.3f);
  r.df = fmaxf((r.dmax - r.dmin) * 5.0f, 0.3f);
  r.sf = fmaxf(r.dsafe, 0.3f);
}
// a predicate under --norm_stl: the reference divides (an IEEE division, kept: scores must agree with it sign for sign)
template <bool NORM>
PSTL_HD float over(float a, float f) { return NORM ? a / f : a; }

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

// Prepared lane waypoint j of a 15-point lane: (x, y, heading, 1 / clamp(|w_j - w_{j+1}|, 1e-7)) -- the reciprocal of the
// length of the segment that starts here; 0 for a segment of length 0 (an invalid lane's all-zero waypoints) and for the last
// point.  The segment is scene data: 20 x rows-per-scene evaluations share the value.  Round 6: the reciprocal instead of the
// length (round 4) -- compute_t2l_dist's `area / clamp(len, 1e-7)` is then one multiplication per row and step instead of an
// IEEE division (~12 instructions); the product differs from the quotient by at most 1 ulp of a distance of a few metres.
// `next` = waypoint j + 1 or null.
PSTL_HD void prep_lane_point(const float* pt, const float* next, float* out) {
  out[0] = pt[0];
  out[1] = pt[1];
  out[2] = pt[2];
  float inv = 0.0f;
  if (next) {
    const float sx = pt[0] - next[0], sy = pt[1] - next[1];
    const float bl = sqrtf(sx * sx + sy * sy);
    if (bl != 0.0f) inv = 1.0f / fmaxf(bl, 1e-7f);
  }
  out[3] = inv;
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
    // exp(-|a - m|) (exp(-inf) = 0 on the first add); the new maximum rescales the old sum, otherwise the term joins it.
    // (the rescaling is one fused multiply-add -- the sum is a running one in any case, and torch.logsumexp's own
    // order of summation is another)
    const float d = a - m;
    const float e = PSTL_EXP(-fabsf(d));
    s = (d > 0.0f) ? __builtin_fmaf(s, e, 1.0f) : (s + e);
    m = fmaxf(a, m);
  }
  PSTL_HD float value() const { return PSTL_LOG(s) + m; }
};

PSTL_HD float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  return PSTL_LOG(PSTL_EXP(a - m) + PSTL_EXP(b - m)) + m;
}

// ---------------------------------------------------------------------------------------------------------------
// A9  signed lateral distance + heading error to the closest segment pair of a 15-waypoint lane
// ---------------------------------------------------------------------------------------------------------------
PSTL_HD unsigned geo_bits(float f) {
  unsigned u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
PSTL_HD float geo_float(unsigned u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
PSTL_HD unsigned umin3(unsigned a, unsigned b, unsigned c) {
  const unsigned ab = a < b ? a : b;
  return ab < c ? ab : c;
}

struct LaneHit {
  float d, th;
  float dd_dx, dd_dy, dth_dth;  // partials (only meaningful when requested)
  int jb;                       // the winning waypoint pair
};

// seg >= 0: the winning pair is known (recorded by the forward sweep at this very state): no ranking
template <bool GRAD>
PSTL_HD void lane_eval(const f4* lane, float px, float py, float pth, LaneHit& h, int seg = -1) {
  int jb = 0;
  if (seg >= 0) {
    jb = seg;
  } else {
    // argmin_j (d_j + d_{j+1}), lowest index on ties (torch.argmin).  Only the INDEX is needed (the winner's distance is
    // computed exactly below), so it rides in the four low mantissa bits of its sum -- sums are non-negative floats, whose bit
    // patterns order like unsigned integers -- and one tree of unsigned minima finds the winner: 14 x (and-or) + 7 three-way
    // minima instead of 14 x (compare + two selects), the most expensive instructions of this loop (profiles/r6/valu_rate.txt).
    // Sums closer than 16 ulp (2^-20 relative: 4 um of the ~4 m between waypoints) count as ties; the distances themselves come
    // from the 1-ulp hardware square root, and the positions they are measured from carry 30 um of float32 rounding at
    // nuScenes' coordinates, so the ranking was never sharper than that.
    f4 p = lane[0];
    float ex = px - p.x, ey = py - p.y;
    float prev = PSTL_SQRT_RANK(ex * ex + ey * ey);
    static_assert((kNseg - 1) % 2 == 0, "two pairs per three-way minimum");
    unsigned best = 0xffffffffu, held = 0u;
    PSTL_UNROLL
    for (int j = 0; j < kNseg - 1; ++j) {
      const f4 n = lane[j + 1];
      ex = px - n.x;
      ey = py - n.y;
      const float cur = PSTL_SQRT_RANK(ex * ex + ey * ey);
      const unsigned key = (geo_bits(prev + cur) & ~15u) | (unsigned)j;
      if (j & 1) best = umin3(best, held, key);
      else held = key;
      prev = cur;
    }
    jb = (int)(best & 15u);
    PSTL_ST_KEEP(jb);
    PSTL_ST(ST_LANE_RANK);
  }
  h.jb = jb;
  const f4 p2 = lane[jb], p3 = lane[jb + 1];
  const float area = px * (p2.y - p3.y) + p2.x * (p3.y - py) + p3.x * (py - p2.y);
  const float ibl = p2.w;   // 1 / clamp(|p2 - p3|, 1e-7), 0 for a degenerate segment: prep_lane_point
  const float qx = px - p2.x, qy = py - p2.y;
  const bool normal = (ibl != 0.0f);
  float q2 = 0.0f, l2 = 0.0f;
  if (normal) {
    h.d = area * ibl;
  } else {   // a degenerate segment (an invalid lane's all-zero waypoints): the distance to the point itself
    q2 = qx * qx + qy * qy;
    l2 = sqrtf(fmaxf(q2, 1e-3f));
    h.d = l2;
  }
  const float du = p2.z - pth;
  float sdu = 0.0f, cdu = 1.0f;
  // The heading term 1 - cos(du).  It feeds a soft-min exponent (tau (thmax - h.th) / thmax), so it has to be good to ~1e-7 --
  // the hardware v_cos_f32 is not.  For the heading errors that occur (|du| <= pi/4) the Taylor polynomial of 1 - cos,
  // truncated below 3e-8, is CLOSER to the exact value than the reference's own float32 `1 - cos(du)` (whose cosine is rounded
  // to 6e-8 next to 1) in ~8 instructions instead of cosf's ~30; beyond that the library call.  Round 4 introduced it for the
  // adjoint's own evaluation; since round 6 the forward sweeps take the same value (so do the scores).
  if (fabsf(du) <= 0.78f) {
    const float x2 = du * du;
    h.th = x2 * (0.5f - x2 * (0x1.555556p-5f - x2 * (0x1.6c16c2p-10f - x2 * 0x1.a01a02p-16f)));
    if (GRAD) sdu = du * (1.0f - x2 * (0x1.555556p-3f - x2 * (0x1.111112p-7f - x2 * 0x1.a01a02p-13f)));
  } else if (GRAD) {
    PSTL_SINCOS(du, &sdu, &cdu);
    h.th = 1.0f - cdu;
  } else {
    cdu = cosf(du);
    h.th = 1.0f - cdu;
  }
  if (GRAD) {
    if (normal) {
      h.dd_dx = (p2.y - p3.y) * ibl;
      h.dd_dy = (p3.x - p2.x) * ibl;
    } else if (q2 >= 1e-3f) {
      const float r = PSTL_RCP_ADJ(l2);
      h.dd_dx = qx * r;
      h.dd_dy = qy * r;
    } else {
      h.dd_dx = 0.0f;
      h.dd_dy = 0.0f;
    }
    h.dth_dth = -sdu;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// A10 clearance to the K neighbours at time t.  nei points at the prepared table of this scene: (K, T, 12) floats
//     = valid, r, cx[4], cy[4], pad, pad.
// ---------------------------------------------------------------------------------------------------------------
struct ClearHit {
  float dn;
  float d_dx, d_dy, d_dth;
  unsigned win;   // REC: neighbour << 4 | circle pair of the minimum (neighbour 15: none, the constant 100)
};

template <bool GRAD, bool REC = false>
PSTL_HD void clearance_eval(const StlEnv& env, const float* nei, int K, int t, float x, float y, float c, float s,
                            ClearHit& h) {
  unsigned win = 0xF0u;
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
    // Exact shortcuts (the result is a hard minimum over neighbours and circle pairs, so a neighbour that cannot be the
    // strict minimum changes nothing): an invalid neighbour contributes exactly 100 whatever its distance, and a valid
    // one whose lower bound -- distance of the ego reference point to its nearest circle centre, minus the longest ego
    // offset and both radii, minus a margin far above the rounding of these few operations -- already reaches the
    // current minimum is skipped without looking at the 16 pairs.  (All 64 lanes of a wave are samples of one scene
    // and mode: the skip is usually wave-uniform.)
    if (valid == 0.0f) {
      if (100.0f < best) {
        best = 100.0f;
        if (GRAD) gx = gy = gth = 0.0f;
        if (REC) win = 0xF0u;
      }
      continue;
    }
    if (valid == 1.0f && best < INFINITY) {
      float m2 = INFINITY;
      PSTL_UNROLL
      for (int j = 0; j < 4; ++j) {
        const float dx = x - nx[j], dy = y - ny[j];
        m2 = fminf(m2, dx * dx + dy * dy);
      }
      const float emax = fmaxf(fmaxf(fabsf(env.eoff[0]), fabsf(env.eoff[1])), fmaxf(fabsf(env.eoff[2]), fabsf(env.eoff[3])));
      const float lb = PSTL_SQRT_RANK(m2) * 0.99999f - (emax + env.er + r) - 1e-3f;   // (1-ulp square root: inside the margin)
      if (fminf(fmaxf(lb, -5.0f), 20.0f) >= best) continue;
    }
    float q = INFINITY;
    float bdx = 0.0f, bdy = 0.0f, boff = 0.0f;  // the closest circle pair (first one on ties)
    unsigned pair = 15u;
    if (!GRAD) {
      // value only: a tree of plain minima over the 16 independent pair distances (the same number as the sequential
      // "if (qq < q)" scan for any non-NaN input, without its compare -> select -> compare chain and the wait states
      // that go with it)
      // (two neighbour circles per packed instruction: 5 packed operations per two pairs, every group of four independent of
      // the others -- the (dx, dy)-packed form the vectoriser found by itself was 3 per pair in one dependent chain, an s_nop
      // between every two of them; same operations on the same operands: same bits)
      // (the minimum is a running one: only the recording sweep keeps the sixteen values for the equality search below)
      float qs[16];
      const v2f nxa = v2f{nx[0], nx[1]}, nxb = v2f{nx[2], nx[3]}, nya = v2f{ny[0], ny[1]}, nyb = v2f{ny[2], ny[3]};
      PSTL_UNROLL
      for (int i = 0; i < 4; ++i) {
        const v2f exi = v2f{ex[i], ex[i]}, eyi = v2f{ey[i], ey[i]};
        const v2f dxa = exi - nxa, dya = eyi - nya, dxb = exi - nxb, dyb = eyi - nyb;
        const v2f qa = dxa * dxa + dya * dya, qb = dxb * dxb + dyb * dyb;
        if (REC) qs[4 * i + 0] = qa.x, qs[4 * i + 1] = qa.y, qs[4 * i + 2] = qb.x, qs[4 * i + 3] = qb.y;
        q = fminf(fminf(q, fminf(qa.x, qa.y)), fminf(qb.x, qb.y));
      }
      if (REC) {   // which pair it was: the first one equal to the minimum (what the sequential "<" scan of the adjoint keeps)
        PSTL_UNROLL
        for (int n = 14; n >= 0; --n) pair = (qs[n] == q) ? (unsigned)n : pair;
      }
    } else {
      // with the gradient the winning pair has to be tracked (finding it afterwards by equality with the tree minimum
      // was measured slower: 16 compares and 48 selects)
      PSTL_UNROLL
      for (int i = 0; i < 4; ++i) {
        PSTL_UNROLL
        for (int j = 0; j < 4; ++j) {
          const float dx = ex[i] - nx[j], dy = ey[i] - ny[j];
          const float qq = dx * dx + dy * dy;
          if (qq < q) {
            q = qq;
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
      if (REC) win = ((unsigned)k << 4) | pair;
      if (GRAD) {
        const bool pass = (car >= -5.0f) && (car <= 20.0f) && (dist > 0.0f);
        const float g = pass ? valid * PSTL_RCP_ADJ(dist) : 0.0f;
        const float ddx = bdx * g, ddy = bdy * g;
        gx = ddx;
        gy = ddy;
        gth = ddx * (-boff * s) + ddy * (boff * c);
      }
    }
  }
  h.dn = best;
  if (REC) h.win = win;
  if (GRAD) {
    h.d_dx = gx;
    h.d_dy = gy;
    h.d_dth = gth;
  }
}

// The same value and partials from the minimum's recorded position (neighbour, circle pair) -- one pair instead of 16 K:
// the operations that produced the recorded minimum, repeated on the same operands (bit-identical).
PSTL_HD void clearance_from_winner(const StlEnv& env, const float* nei, int t, float x, float y, float c, float s,
                                   unsigned win, ClearHit& h) {
  const unsigned k = win >> 4, pi = win & 15u;
  h.dn = 100.0f;
  h.d_dx = h.d_dy = h.d_dth = 0.0f;
  if (k == 15u) return;
  const f4* e = reinterpret_cast<const f4*>(nei + (size_t)(k * kT + t) * kNeiPrep);
  const f4 a = e[0], b = e[1], cc = e[2];
  const float valid = a.x, r = a.y;
  const unsigned i = pi >> 2, j = pi & 3u;
  const float off = i == 0 ? env.eoff[0] : (i == 1 ? env.eoff[1] : (i == 2 ? env.eoff[2] : env.eoff[3]));
  const float nxj = j == 0 ? a.z : (j == 1 ? a.w : (j == 2 ? b.x : b.y));
  const float nyj = j == 0 ? b.z : (j == 1 ? b.w : (j == 2 ? cc.x : cc.y));
  const float exi = off * c + x, eyi = off * s + y;
  const float dx = exi - nxj, dy = eyi - nyj;
  const float q = dx * dx + dy * dy;
  const float dist = sqrtf(q);
  const float car = dist - env.er - r;
  const float clipped = fminf(fmaxf(car, -5.0f), 20.0f);
  h.dn = clipped * valid + (1.0f - valid) * 100.0f;
  const bool pass = (car >= -5.0f) && (car <= 20.0f) && (dist > 0.0f);
  const float g = pass ? valid * PSTL_RCP_ADJ(dist) : 0.0f;
  const float ddx = dx * g, ddy = dy * g;
  h.d_dx = ddx;
  h.d_dy = ddy;
  h.d_dth = ddx * (-off * s) + ddy * (off * c);
}

// ---------------------------------------------------------------------------------------------------------------
// State sources: the unicycle driven by the row's controls (A6, nusc_train.py:29-49), or trajectories handed in by the
// caller (what compute_stl_dense receives).  get() returns state t and cos/sin of its heading (shared by the dynamics
// and by the ego circle row).
// ---------------------------------------------------------------------------------------------------------------
struct alignas(8) f2 {
  float x, y;
};
PSTL_HD void store_quad(float* p, float x, float y, float z, float w) {   // one 16-byte store (p 16-byte aligned)
#if defined(__HIPCC__)
  typedef float st4 __attribute__((ext_vector_type(4)));
  st4 v;
  v.x = x, v.y = y, v.z = z, v.w = w;
  *reinterpret_cast<st4*>(p) = v;
#else
  p[0] = x, p[1] = y, p[2] = z, p[3] = w;
#endif
}
PSTL_HD void store_pair(float* p, float x, float y) {   // one 8-byte store
#if defined(__HIPCC__)
  typedef float st2 __attribute__((ext_vector_type(2)));
  st2 v;
  v.x = x;
  v.y = y;
  *reinterpret_cast<st2*>(p) = v;
#else
  p[0] = x;
  p[1] = y;
#endif
}

// (w, a) of time step t from a row's control buffer: contiguous rows (us == 1, 16-byte aligned: 160 B per row) are read
// 16 bytes = two time steps at a time -- every per-lane access is a gather over 64 cache lines whatever its width, so
// wider accesses are proportionally fewer gathers; element-major buffers (us = N) are coalesced and read one by one.
#if defined(__HIPCC__)
typedef float ctrl4 __attribute__((ext_vector_type(4)));   // a native vector type: the 16-byte load stays in registers
typedef float ctrl2 __attribute__((ext_vector_type(2)));
#else
struct alignas(16) ctrl4 {
  float x, y, z, w;
};
struct alignas(8) ctrl2 {
  float x, y;
};
#endif

struct CtrlReader {
  const float* u;
  long us;
  float w1, a1;   // the second time step of the last 16-byte read
  PSTL_HD CtrlReader(const float* u_, long us_) : u(u_), us(us_), w1(0.0f), a1(0.0f) {}
  PSTL_HD void get(int t, float& w, float& a) {   // t must be visited in increasing order, starting at an even step
    if (us == 1) {
      if ((t & 1) == 0) {
        const ctrl4 c = *reinterpret_cast<const ctrl4*>(u + 2 * t);
        w = c.x;
        a = c.y;
        w1 = c.z;
        a1 = c.w;
      } else {
        w = w1;
        a = a1;
      }
    } else {
      w = u[(2 * t) * us];
      a = u[(2 * t + 1) * us];
    }
  }
};
PSTL_HD void ctrl_pair(const float* u, long us, int t, float& w, float& a) {   // one (w, a) pair, any order
  if (us == 1) {
    const ctrl2 p = *reinterpret_cast<const ctrl2*>(u + 2 * t);
    w = p.x;
    a = p.y;
  } else {
    w = u[(2 * t) * us];
    a = u[(2 * t + 1) * us];
  }
}

struct DynSrc {
  float x, y, th, v;
  CtrlReader rd;
  float ws, as, dt;
  // us: distance (in floats) between consecutive control values of this row: 1, or N for an element-major buffer
  PSTL_HD DynSrc(const float* s0, const float* u_, float ws_, float as_, float dt_, long us_ = 1)
      : x(s0[0]), y(s0[1]), th(s0[2]), v(s0[3]), rd(u_, us_), ws(ws_), as(as_), dt(dt_) {}
  PSTL_HD void get(int t, float& X, float& Y, float& TH, float& V, float& c, float& s) {
    X = x, Y = y, TH = th, V = v;
    PSTL_SINCOS(th, &s, &c);
    float wr, ar;
    rd.get(t, wr, ar);
    const float w = wr * ws;
    const float a = ar * as;
    const float dx = v * c;
    const float dy = v * s;
    x = x + dx * dt;
    y = y + dy * dt;
    th = th + w * dt;
    v = v + a * dt;
  }
};

struct GivenSrc {
  const f4* p;  // (T,4) states of this row
  PSTL_HD void get(int t, float& X, float& Y, float& TH, float& V, float& c, float& s) const {
    const f4 q = p[t];
    X = q.x, Y = q.y, TH = q.z, V = q.w;
    PSTL_SINCOS(q.z, &s, &c);
  }
};

// ---------------------------------------------------------------------------------------------------------------
// A8  formulas, evaluated in ONE forward sweep over time.
//   "Always" at time 0 (G s[0] = softmin over all t) is an order-independent running log-sum-exp.
//   "Eventually(0,10, Always(...))" needs the suffix soft-mins G s[k] = softmin(s[k:T]) for k < 10: the sweep keeps
//   the tail t >= 10 as one running log-sum-exp and parks the first 10 values of the signal in scratch; a 10-step
//   backward scan then yields every suffix value, which feeds the outer soft-max.  The 405 torch.logsumexp calls of
//   the reference per evaluation become ~100 exp and ~30 log per row.
// ---------------------------------------------------------------------------------------------------------------
struct ReachAcc {  // one "reach the side lane" pair: band = softmin2(d - dmin, dmax - d) and the heading term
  Lse tailb, tailt;
  PSTL_HD void init() {
    tailb.init();
    tailt.init();
  }
  PSTL_HD void step(float tau, int t, float s1, float s2, float a3, Scratch st, int base) {
    const float band = -(lse2(-s1 * tau, -s2 * tau) / tau);
    const float ab = -band * tau;
    if (t < kFwin) {
      st.at(base + t) = ab;
      st.at(base + kFwin + t) = a3;
    } else {
      tailb.add(ab);
      tailt.add(a3);
    }
  }
  // suffix scan: leaves L_k = logsumexp(a[k:T]) in scratch (the adjoint needs them) and returns the two F10 values
  PSTL_HD void finish(float tau, Scratch st, int base, float& Lfb, float& Lft) {
    Lse fb, ft;
    fb.init();
    ft.init();
    PSTL_NOUNROLL
    for (int k = kFwin - 1; k >= 0; --k) {
      tailb.add(st.at(base + k));
      tailt.add(st.at(base + kFwin + k));
      const float lb = tailb.value(), lt = tailt.value();
      st.at(base + k) = lb;
      st.at(base + kFwin + k) = lt;
      fb.add(-(lb / tau) * tau);
      ft.add(-(lt / tau) * tau);
    }
    Lfb = fb.value();
    Lft = ft.value();
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
  for (int i = 0; i < 6; ++i) s += (i < n) ? PSTL_EXP(a[i] - m) : 0.0f;
  return -((PSTL_LOG(s) + m) / tau);
}

// The geometry of the sweeps, computed ahead (latency layout of the STL kernels: the waves of a workgroup each take a few of
// the 20 time steps; the sweeps themselves then only read): per time step the clearance, the lane distance and
// heading term.
//   slots 0-3   (forward sweep): clearance, lane distance, heading term 1 - cos, speed
//   slots 4-12  (adjoint): the clearance's partials (x, y, heading), the heading term as the adjoint's own evaluation yields
//                          it and the lane distance's partials (x, y) and the heading term's, then cos, sin of the heading
// (The winners of the hard minima stay inside stl_geometry: the adjoint's partials are evaluated right there.)  Every slot has
// one reader, and the stage that has read it writes its own result over it (k_guidance_iter, SPLIT: the chains' values and the
// weights in step 0's slots, the direct partials over the clearance partials, the gradients over the forward values) -- no
// buffer of its own for any hand-over between the stages.
// The adjoint's state is the forward sweep's, bit for bit: stl_eval_grad re-derives the states of a 4-step block from the
// block's checkpoint with the forward sweep's own operations, the exact sincosf included.  (Round 4 tried the hardware
// v_sin_f32 / v_cos_f32 there -- the adjoint only reaches a gradient -- and the reference's autograd gradients were missed:
// a position that is 1e-5 m off moves a soft-min exponent by tau * 1e-5 = 1e-3, i.e. the per-step weights by 0.1 % against
// normalisers the forward sweep computed at the exact states; test_stl_backward_matches_reference_autograd[stl_mixed] failed
// its 5e-3.  The divisions, square roots and the heading term's sin / cos of the adjoint are another matter: they scale a
// partial derivative, not an exponent.)
constexpr int kGeoSlots = 13;
struct GeoPre {
  const float* p;   // element (t, c) of this lane at p[(kGeoSlots t + c) * stride]
  int stride;
  PSTL_HD float at(int t, int c) const { return p[(kGeoSlots * t + c) * stride]; }
};
// Steps [t0, t1) of the row whose states `src` yields (the states of the steps before t0 are generated and dropped: the
// dynamics are a handful of operations per step).  The very calls of stl_eval_rec<false, ., true, .>: same values, bit for bit.
// ... and of stl_eval_grad's adjoint for the same step (the partials at the recorded winners; K > kRecMaxK: ranked again).
// ADJ = false: the forward values only (slots 0-2; value-only callers -- scoring -- run clearance_eval without the record).
// one step of the geometry, written to the slots of step t (o = its slot 0)
template <bool ADJ = true>
PSTL_HD void stl_geometry_step(const StlEnv& env, const f4* sel_lane, const float* nei, int K, int t, float x, float y, float th,
                               float v, float c, float s, float* o, int stride) {
  const bool use_rec = K <= kRecMaxK;
  ClearHit ch;
  clearance_eval<false, ADJ>(env, nei, K, t, x, y, c, s, ch);
  LaneHit h;
  lane_eval<false>(sel_lane, x, y, th, h);
  o[0 * stride] = ch.dn;
  o[1 * stride] = h.d;
  o[2 * stride] = h.th;
  o[3 * stride] = v;      // (the sweeps read the speed here instead of running the dynamics -- 20 sincosf -- once more)
  if (!ADJ) return;
  ClearHit cg;
  if (use_rec) clearance_from_winner(env, nei, t, x, y, c, s, ch.win, cg);
  else clearance_eval<true>(env, nei, K, t, x, y, c, s, cg);
  LaneHit hg;
  lane_eval<true>(sel_lane, x, y, th, hg, use_rec ? h.jb : -1);
  o[4 * stride] = cg.d_dx;
  o[5 * stride] = cg.d_dy;
  o[6 * stride] = cg.d_dth;
  o[7 * stride] = hg.th;
  o[8 * stride] = hg.dd_dx;
  o[9 * stride] = hg.dd_dy;
  o[10 * stride] = hg.dth_dth;
  o[11 * stride] = c;
  o[12 * stride] = s;
}

template <bool ADJ = true, class Src>
PSTL_HD void stl_geometry(const StlEnv& env, const f4* sel_lane, const float* nei, int K, Src src, int t0, int t1, float* out,
                          int stride) {
  PSTL_NOUNROLL
  for (int t = 0; t < t1; ++t) {
    float x, y, th, v, c, s;
    src.get(t, x, y, th, v, c, s);
    if (t < t0) continue;
    stl_geometry_step<ADJ>(env, sel_lane, nei, K, t, x, y, th, v, c, s, out + (kGeoSlots * t) * stride, stride);
  }
}

struct FwdOut {  // what the adjoint needs from the forward sweep
  float Lv1, Lv2, Ls, L1, L2, L3, Lfb, Lft, score;
};

// Evaluates the formulas of one row.
//   ALL3 = true : all three formulas -> out3[0..2]; returns the mode-selected score   (scratch: kScratchFwd3)
//   ALL3 = false: only the formula of r.mode                                           (scratch: kScratchFwd)
//   XY != -1    : additionally parks the state of every 4th step at scratch[XY ...] (x, y, th, v; 5 each)  (adjoint)
//   REC         : records the winners of the hard minima (lane segment, neighbour, circle pair) per step in `rec`
//                 (for the adjoint; !ALL3)
// (The latency layout, whose waves compute the geometry of every step ahead, has sweeps of its own: stl_pre_chain, adj_pre_*.)
template <bool ALL3, int XY, bool REC, bool NORM, class Src>
PSTL_HD float stl_eval_rec(const StlEnv& env, const StlRow& r, const f4* lanes, const float* nei, int K, Src src, Scratch st,
                           int tab, float* out3, FwdOut* fo, Rec& rec) {
  static_assert(!(ALL3 && REC), "winners are recorded for the selected formula only");
  const float tau = env.tau;
  Lse gv1, gv2, gsafe, g1, g2, g3;
  gv1.init();
  gv2.init();
  gsafe.init();
  g1.init();
  g2.init();
  g3.init();
  ReachAcc R1, R2;
  R1.init();
  R2.init();
  const int mode = r.mode;
  const f4* sel_lane = lanes + (mode < 3 ? mode : 0) * kNseg;
  // (thmax - th) / thmax, twenty times per row: the division once per row, then a product (1 ulp from the quotient)
  const float ith = 1.0f / r.thmax;
  PSTL_NOUNROLL
  for (int t = 0; t < kT; ++t) {
    float x, y, th, v, c, s;
    src.get(t, x, y, th, v, c, s);
    if (XY >= 0 && (t & (kCkStride - 1)) == 0) {   // state checkpoints for the adjoint (every 4th step)
      const int k = t / kCkStride;
      st.at(XY + k) = x;
      st.at(XY + kCk + k) = y;
      st.at(XY + 2 * kCk + k) = th;
      st.at(XY + 3 * kCk + k) = v;
    }
    PSTL_ST_KEEP(x); PSTL_ST_KEEP(y); PSTL_ST_KEEP(c); PSTL_ST_KEEP(s); PSTL_ST_KEEP(v); PSTL_ST_KEEP(th);
    PSTL_ST(ST_DYN);
    gv1.add(-over<NORM>(v - r.vmin, r.vf) * tau);
    gv2.add(-over<NORM>(-v + r.vmax, r.vf) * tau);
    PSTL_ST_KEEP(gv1.s); PSTL_ST_KEEP(gv2.s); PSTL_ST_KEEP(gv1.m); PSTL_ST_KEEP(gv2.m);
    PSTL_ST(ST_LSE);
    ClearHit ch;
    LaneHit h;
    clearance_eval<false, REC>(env, nei, K, t, x, y, c, s, ch);
    PSTL_ST_KEEP(ch.dn);
    PSTL_ST(ST_CLEAR);
    gsafe.add(-over<NORM>(ch.dn - r.dsafe, r.sf) * tau);
    PSTL_ST_KEEP(gsafe.s); PSTL_ST_KEEP(gsafe.m);
    PSTL_ST(ST_LSE);
    lane_eval<false>(ALL3 ? lanes : sel_lane, x, y, th, h);
    if (REC) rec_put(rec, t, (unsigned)h.jb, ch.win);
    PSTL_ST_KEEP(h.d); PSTL_ST_KEEP(h.th);
    PSTL_ST(ST_LANE_TAIL);
    {
      const float s1 = over<NORM>(h.d - r.dmin, r.df), s2 = over<NORM>(-h.d + r.dmax, r.df);
      const float a3 = -((r.thmax - h.th) * ith) * tau;
      // (the selected formula only: lane keeping reads the three "always" terms, the lane changes the two "reach" terms;
      // the kernels that map a wavefront to ONE (scene, mode) take one side of each branch as a whole)
      if (ALL3 || mode == 0) {
        g1.add(-s1 * tau);
        g2.add(-s2 * tau);
        g3.add(a3);
      }
      if (!ALL3 && (mode == 1 || mode == 2)) R1.step(tau, t, s1, s2, a3, st, tab);
    }
    if (ALL3) {
      lane_eval<false>(lanes + kNseg, x, y, th, h);
      R1.step(tau, t, over<NORM>(h.d - r.dmin, r.df), over<NORM>(-h.d + r.dmax, r.df), -((r.thmax - h.th) * ith) * tau, st, tab);
      lane_eval<false>(lanes + 2 * kNseg, x, y, th, h);
      R2.step(tau, t, over<NORM>(h.d - r.dmin, r.df), over<NORM>(-h.d + r.dmax, r.df), -((r.thmax - h.th) * ith) * tau, st,
              tab + 2 * kFwin);
    }
    PSTL_ST_KEEP(g1.s); PSTL_ST_KEEP(g2.s); PSTL_ST_KEEP(g3.s); PSTL_ST_KEEP(R1.tailb.s); PSTL_ST_KEEP(R1.tailt.s);
    PSTL_ST(ST_LSE);
  }
  const float Lv1 = gv1.value(), Lv2 = gv2.value(), Ls = gsafe.value();
  const float Vv1 = -(Lv1 / tau), Vv2 = -(Lv2 / tau), Vs = -(Ls / tau);
  const float L1 = g1.value(), L2 = g2.value(), L3 = g3.value();
  float Lfb = 0.0f, Lft = 0.0f;
  float sc[3] = {0.0f, 0.0f, 0.0f};
  if (ALL3 || mode == 0) {
    const float v6[6] = {Vv1, Vv2, -(L1 / tau), -(L2 / tau), -(L3 / tau), Vs};
    sc[0] = conj6(v6, 6, tau);
  }
  if (ALL3) {
    R1.finish(tau, st, tab, Lfb, Lft);
    const float v1[5] = {Vv1, Vv2, Lfb / tau, Lft / tau, Vs};
    sc[1] = conj6(v1, 5, tau);
    R2.finish(tau, st, tab + 2 * kFwin, Lfb, Lft);
    const float v2[5] = {Vv1, Vv2, Lfb / tau, Lft / tau, Vs};
    sc[2] = conj6(v2, 5, tau);
  } else if (mode == 1 || mode == 2) {
    R1.finish(tau, st, tab, Lfb, Lft);
    const float v1[5] = {Vv1, Vv2, Lfb / tau, Lft / tau, Vs};
    sc[mode] = conj6(v1, 5, tau);
  }
  if (ALL3 && out3) {
    out3[0] = sc[0];
    out3[1] = sc[1];
    out3[2] = sc[2];
  }
  float score;
  if (ALL3) {
    // get_stl_scores (nusc_train.py:150-151): masked sum of the three formulas plus the constant 1 for outliers
    score = sc[0] * (mode == 0 ? 1.0f : 0.0f) + sc[1] * (mode == 1 ? 1.0f : 0.0f) + sc[2] * (mode == 2 ? 1.0f : 0.0f) +
            1.0f * (mode == 3 ? 1.0f : 0.0f);
  } else {
    score = mode == 3 ? 1.0f : sc[mode];
  }
  if (fo) {
    fo->Lv1 = Lv1, fo->Lv2 = Lv2, fo->Ls = Ls, fo->L1 = L1, fo->L2 = L2, fo->L3 = L3, fo->Lfb = Lfb, fo->Lft = Lft;
    fo->score = score;
  }
  PSTL_ST(ST_FWD_FINISH);
  return score;
}

template <bool ALL3, int XY, bool NORM = false, class Src>
PSTL_HD float stl_eval(const StlEnv& env, const StlRow& r, const f4* lanes, const float* nei, int K, Src src, Scratch st,
                       int tab, float* out3, FwdOut* fo) {
  Rec none;
  return stl_eval_rec<ALL3, XY, false, NORM, Src>(env, r, lanes, nei, K, src, st, tab, out3, fo, none);
}

// ---------------------------------------------------------------------------------------------------------------
// Forward + adjoint of one row: returns the score and calls emit(t, gw, ga, w, a) once for every t in [0,T) with
// (gw, ga) = dscore_fn(score) * d score / d (u[2t], u[2t+1]) and (w, a) = the stored values u[2t], u[2t+1] themselves
// (the adjoint has them in registers, so an optimiser step needs no further read; u = the 40 control values, scaled by wscale/ascale
// inside the dynamics).  Only the formula of r.mode carries gradient (the others are multiplied by a 0 mask in the
// reference).  Scratch: kScratchGrad = 40 floats per lane = the state (x, y, th, v) at every 4th step (4*5) + the two
// suffix tables (2*10).  The states in between are re-derived block by block from the checkpoint with the forward
// sweep's own operations -- bit-identical, and the sin/cos they need are the ones the adjoint needs anyway -- instead of
// being stored (80 floats): 10 KB of LDS per wavefront instead of 25.6 KB, i.e. 12 resident wavefronts per CU, not 6.
// (These kernels are latency-bound at that occupancy: measured 0.94 / 1.02 / 1.32 / 1.65 ms at 7 / 6 / 4 / 3 per CU.)
// ---------------------------------------------------------------------------------------------------------------
PSTL_HD float pick4(const float (&a)[kCkStride], int i) {   // register array, dynamic index: three selects
  static_assert(kCkStride == 4, "pick4");
  return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
}

// What the adjoint takes over from the forward sweep besides the scratch (checkpoints and suffix tables): 17 registers.
// (A two-launch form of k_guidance_iter that handed both through device memory, the rows with an active loss compacted
// between the sweeps, was measured in round 4 and not kept: profiles/r4/guidance_two_launch_compaction.patch.)
struct AdjState {
  FwdOut fo;
  Rec rec;
};

// 16 bytes = the control pairs of steps 2q, 2q + 1 of a row-major row (us == 1)
PSTL_HD ctrl4 ctrl_quad(const float* u, int q) { return *reinterpret_cast<const ctrl4*>(u + 4 * q); }

// emit() is always called for t = T-1 ... 0, in that order; a row without gradient hands it exact zeros.
// Row-major rows: the loop used to read one control pair per step right in front of the emit() that needs it -- and emit()
// stores to the same buffer, so no load could be moved up: twenty gather latencies in a row, a quarter of the lifetime of a
// k_guidance_iter wavefront (stamp build, round 6).  Now four steps per iteration, their two 16-byte pieces requested one
// iteration ahead: one exposed latency per row, ten loads instead of twenty.
template <class EmitFn>
PSTL_HD void stl_grad_zero(const float* u, long us, EmitFn emit) {
  static_assert(kT % 4 == 0, "four steps per iteration");
  if (us == 1) {
    ctrl4 hi = ctrl_quad(u, kT / 2 - 1), lo = ctrl_quad(u, kT / 2 - 2);
    PSTL_NOUNROLL
    for (int it = kT / 4 - 1; it >= 0; --it) {
      ctrl4 nhi = hi, nlo = lo;
      if (it > 0) {   // (the pieces of the four steps below: nothing this iteration stores to)
        nhi = ctrl_quad(u, 2 * it - 1);
        nlo = ctrl_quad(u, 2 * it - 2);
      }
      emit(4 * it + 3, 0.0f, 0.0f, hi.z, hi.w);
      emit(4 * it + 2, 0.0f, 0.0f, hi.x, hi.y);
      emit(4 * it + 1, 0.0f, 0.0f, lo.z, lo.w);
      emit(4 * it, 0.0f, 0.0f, lo.x, lo.y);
      hi = nhi;
      lo = nlo;
    }
  } else {
    PSTL_NOUNROLL
    for (int t = kT - 1; t >= 0; --t) {
      float w, a;
      ctrl_pair(u, us, t, w, a);
      emit(t, 0.0f, 0.0f, w, a);
    }
  }
  PSTL_ST(ST_ZERO_EMIT);
}

// scratch: [checkpoints 4 x 5 | suffix tables 2 x 10]
template <bool NORM = false>
PSTL_HD float stl_grad_forward(const StlEnv& env, const StlRow& r, const f4* lanes, const float* nei, int K, const float* s0,
                               const float* u, Scratch st, float wscale, float ascale, long us, AdjState& S) {
  const int LB = 4 * kCk;
  FwdOut fo;
  Rec rec;
  rec_clear(rec);
  const float score = stl_eval_rec<false, 0, true, NORM, DynSrc>(env, r, lanes, nei, K, DynSrc(s0, u, wscale, ascale, env.dt, us), st,
                                                                 LB, nullptr, &fo, rec);
  S.fo = fo;
  S.rec = rec;
  return score;
}

template <bool NORM = false, class EmitFn>
PSTL_HD void stl_grad_adjoint(const StlEnv& env, const StlRow& r, const f4* lanes, const float* nei, int K, const float* u,
                              Scratch st, float wscale, float ascale, long us, const AdjState& S, float dscore_in, EmitFn emit) {
  const float tau = env.tau;
  const int mode = r.mode;
  const f4* lane = lanes + mode * kNseg;
  const int CKP = 0, LB = 4 * kCk, LT = LB + kFwin;
  const bool use_rec = K <= kRecMaxK;   // uniform; with more neighbours the record is written but not trusted
  const Rec rec = S.rec;
  const float score = S.fo.score;
  const float Lv1 = S.fo.Lv1, Lv2 = S.fo.Lv2, Ls = S.fo.Ls, L1 = S.fo.L1, L2 = S.fo.L2, L3 = S.fo.L3, Lfb = S.fo.Lfb, Lft = S.fo.Lft;
  float V[6];
  int n;
  if (mode == 0) {
    V[0] = -(Lv1 / tau), V[1] = -(Lv2 / tau), V[2] = -(L1 / tau), V[3] = -(L2 / tau), V[4] = -(L3 / tau), V[5] = -(Ls / tau);
    n = 6;
  } else {
    V[0] = -(Lv1 / tau), V[1] = -(Lv2 / tau), V[2] = Lfb / tau, V[3] = Lft / tau, V[4] = -(Ls / tau);
    n = 5;
  }
  // (row-major rows: the 16-byte pieces of the last block's controls, requested before anything else -- see the block loop)
  ctrl4 qa = ctrl4{0.0f, 0.0f, 0.0f, 0.0f}, qb = qa;
  if (us == 1) {
    qa = ctrl_quad(u, 2 * kCk - 2);
    qb = ctrl_quad(u, 2 * kCk - 1);
  }
  const float Lout = -score * tau;  // logsumexp of (-V_i tau)
  float om[6];
  PSTL_UNROLL
  for (int i = 0; i < 6; ++i) om[i] = i < n ? PSTL_EXP(-V[i] * tau - Lout) * dscore_in : 0.0f;  // d score / d V_i
  const float o_v1 = om[0], o_v2 = om[1], o_s = (mode == 0) ? om[5] : om[4];
  {
    float w, a;
    if (us == 1) w = qb.z, a = qb.w;
    else ctrl_pair(u, us, kT - 1, w, a);
    emit(kT - 1, 0.0f, 0.0f, w, a);  // the last control never reaches a scored state
  }
  if (mode != 0) {
    // d F10(G s) / d s_u = sum_{k <= m} q_k exp(a_u - L_k), m = min(u, 9), q_k = exp(tau g_k - Lf).  With
    // S_m = sum_{k<=m} q_k exp(L_m - L_k) (a 10-step recurrence; L_k decreases with k so every exponent is <= 0)
    // this is exp(a_u - Lambda_m), Lambda_m = L_m - log S_m: one exp per time step instead of one per (u, k).
    float sb = 0.0f, sth = 0.0f, lb_prev = 0.0f, lt_prev = 0.0f;
    PSTL_NOUNROLL
    for (int k = 0; k < kFwin; ++k) {
      const float lb = st.at(LB + k), lt = st.at(LT + k);
      const float qb = PSTL_EXP(-lb - Lfb), qt = PSTL_EXP(-lt - Lft);   // exp(tau g_k - Lf), g_k = -(L_k / tau): a gradient weight
      sb = (k == 0) ? qb : sb * PSTL_EXP(lb - lb_prev) + qb;
      sth = (k == 0) ? qt : sth * PSTL_EXP(lt - lt_prev) + qt;
      st.at(LB + k) = lb - PSTL_LOG(sb);
      st.at(LT + k) = lt - PSTL_LOG(sth);
      lb_prev = lb;
      lt_prev = lt;
    }
  }
  PSTL_ST(ST_ADJ_HEAD);
  // ---- adjoint, backwards in time -----------------------------------------------------------------------------
  float lx = 0.0f, ly = 0.0f, lth = 0.0f, lv = 0.0f;  // lambda_{t+1}
  const float dt = env.dt;
  const float inv_thmax = 1.0f / r.thmax;   // (the forward sweep's own reciprocal: s3 below is its value, bit for bit)
  // The states are not stored per step: block by block (4 steps), they are re-derived from the block's checkpoint with the
  // forward sweep's own operations (bit-identical), kept in registers, and consumed in reverse order.  The controls a
  // block reads (steps < its last one) have not been rewritten yet by emit(), which has only reached later steps.
  PSTL_NOUNROLL
  for (int blk = kCk - 1; blk >= 0; --blk) {
    float bx[kCkStride], by[kCkStride], bth[kCkStride], bv[kCkStride], bc[kCkStride], bs[kCkStride];
    float bw[kCkStride], ba[kCkStride];   // stored controls of steps 4blk-1, 4blk, 4blk+1, 4blk+2: the ones emit() is called for
    {
      float x = st.at(CKP + blk), y = st.at(CKP + kCk + blk), th = st.at(CKP + 2 * kCk + blk), v = st.at(CKP + 3 * kCk + blk);
      float wr[kCkStride - 1], ar[kCkStride - 1];   // stored controls of steps 4blk, 4blk+1, 4blk+2
      if (us == 1) {
        // Row-major rows: this block's two 16-byte pieces were requested while the block above was computed (qa, qb), and the
        // ones of the block below are requested here -- emit() has only rewritten later steps, and stores nothing below step
        // 4blk-1 before they land.  (One gather latency per block used to sit in front of the dynamics: a fifth of a wavefront's
        // lifetime in k_guidance_iter.)  Step 4blk-1, which this block's last emit() takes, is the upper half of the second of them.
        wr[0] = qa.x, ar[0] = qa.y, wr[1] = qa.z, ar[1] = qa.w, wr[2] = qb.x, ar[2] = qb.y;
        if (blk > 0) {
          qa = ctrl_quad(u, 2 * blk - 2);
          qb = ctrl_quad(u, 2 * blk - 1);
          bw[0] = qb.z, ba[0] = qb.w;
        }
      } else {
        CtrlReader rd(u, us);
        if (blk > 0) ctrl_pair(u, us, blk * kCkStride - 1, bw[0], ba[0]);
        PSTL_UNROLL
        for (int i = 0; i + 1 < kCkStride; ++i) rd.get(blk * kCkStride + i, wr[i], ar[i]);
      }
      PSTL_UNROLL
      for (int i = 0; i < kCkStride; ++i) {
        float c, s;
        PSTL_SINCOS(th, &s, &c);   // (exact: see the note at kGeoSlots)
        bx[i] = x, by[i] = y, bth[i] = th, bv[i] = v, bc[i] = c, bs[i] = s;
        if (i + 1 < kCkStride) {
          bw[i + 1] = wr[i], ba[i + 1] = ar[i];
          const float w = wr[i] * wscale;
          const float a = ar[i] * ascale;
          const float dx = v * c;
          const float dy = v * s;
          x = x + dx * dt;
          y = y + dy * dt;
          th = th + w * dt;
          v = v + a * dt;
        }
      }
    }
  PSTL_ST(ST_ADJ_REDERIVE);
  PSTL_UNROLL   // (unrolled: the block's register arrays are then indexed by constants -- 24 selects per step otherwise)
  for (int i = kCkStride - 1; i >= 0; --i) {
    const int t = blk * kCkStride + i;
    if (t == 0) break;
    const float x = pick4(bx, i), y = pick4(by, i), th = pick4(bth, i), v = pick4(bv, i), c = pick4(bc, i), s = pick4(bs, i);
    // direct partials of the score w.r.t. state t
    float gx, gy, gth, gv;
    // (--norm_stl: the predicates are a / f; the chain rule adds the factor 1 / f, as autograd's division does)
    gv = o_v1 * PSTL_EXP(-over<NORM>(v - r.vmin, r.vf) * tau - Lv1) - o_v2 * PSTL_EXP(-over<NORM>(-v + r.vmax, r.vf) * tau - Lv2);
    if (NORM) gv = gv / r.vf;
    PSTL_ST(ST_ADJ_WEIGHTS);
    ClearHit ch;
    if (use_rec) {
      clearance_from_winner(env, nei, t, x, y, c, s, rec_win(rec, t), ch);
    } else {
      clearance_eval<true>(env, nei, K, t, x, y, c, s, ch);
    }
    PSTL_ST(ST_ADJ_CLEAR);
    float gs = o_s * PSTL_EXP(-over<NORM>(ch.dn - r.dsafe, r.sf) * tau - Ls);
    if (NORM) gs = gs / r.sf;
    gx = gs * ch.d_dx;
    gy = gs * ch.d_dy;
    gth = gs * ch.d_dth;
    LaneHit h;
    PSTL_ST(ST_ADJ_WEIGHTS);
    lane_eval<true>(lane, x, y, th, h, use_rec ? (int)rec_seg(rec, t) : -1);
    PSTL_ST(ST_ADJ_LANE);
    const float s1 = over<NORM>(h.d - r.dmin, r.df), s2 = over<NORM>(-h.d + r.dmax, r.df), s3 = (r.thmax - h.th) * inv_thmax;
    float gd, gsth;  // d score / d d_t , d score / d s3_t
    if (mode == 0) {
      gd = om[2] * PSTL_EXP(-s1 * tau - L1) - om[3] * PSTL_EXP(-s2 * tau - L2);
      gsth = om[4] * PSTL_EXP(-s3 * tau - L3);
    } else {
      const float a1 = -s1 * tau, a2 = -s2 * tau;
      const float lp = lse2(a1, a2);
      const float ab = lp, a3 = -s3 * tau;   // (ab = -band tau with band = -(lp / tau))
      const int m = t < kFwin - 1 ? t : kFwin - 1;
      const float wb = PSTL_EXP(ab - st.at(LB + m));
      const float wt = PSTL_EXP(a3 - st.at(LT + m));
      gd = om[2] * wb * (PSTL_EXP(a1 - lp) - PSTL_EXP(a2 - lp));
      gsth = om[3] * wt;
    }
    if (NORM) gd = gd / r.df;
    gx += gd * h.dd_dx;
    gy += gd * h.dd_dy;
    gth += gsth * (-inv_thmax) * h.dth_dth;
    PSTL_ST(ST_ADJ_WEIGHTS);
    // lambda_t = direct_t + J_t^T lambda_{t+1}
    const float nlth = gth + lth + lx * (-(v * s) * dt) + ly * ((v * c) * dt);
    const float nlv = gv + lv + lx * (c * dt) + ly * (s * dt);
    lx = gx + lx;
    ly = gy + ly;
    lth = nlth;
    lv = nlv;
    PSTL_ST(ST_ADJ_COSTATE);
    // state_t = f(state_{t-1}, u_{t-1}):  th_t = th_{t-1} + w dt ; v_t = v_{t-1} + a dt
    emit(t - 1, lth * dt * wscale, lv * dt * ascale, pick4(bw, i), pick4(ba, i));
    PSTL_ST(ST_EMIT);
  }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Both sweeps over precomputed geometry, in parts (latency layout of k_guidance_iter).  What one wavefront used to walk in
// 20 + 19 dependent steps -- six running log-sum-exps forwards, ~110 instructions of soft-min weights and partial
// derivatives per step backwards, then the costate -- splits into
//   (0) the forward sweep's chains, which share nothing but their inputs: four wavefronts take one group each;
//   (1) the score and the weights of the formula's terms, once per row;
//   (2) the direct partials of the score with respect to the state of step t, which need nothing of the other steps: the
//       waves of the workgroup take two steps each;
//   (3) the costate recursion, ~20 instructions per step.
// The expressions are stl_eval_rec's and stl_grad_adjoint's, operation for operation, each chain in its own order: same bits.
// Scratch layout of the precomputed-geometry form: suffix tables LB = 0, LT = kFwin.
// ---------------------------------------------------------------------------------------------------------------
struct AdjCtx {   // 12 words per row
  float Lv1, Lv2, Ls, L1, L2, L3;   // (lane changes: L1 = Lfb, L3 = Lft, L2 unused)
  float om[6];
};

// part 0, chain group `which` of a row of mode 0, 1 or 2: 0 = speed (o0 = Lv1, o1 = Lv2), 1 = clearance (o0 = Ls),
// 2 = lane distance (mode 0: o0 = L1, o1 = L2; lane changes: the band -- o0 = Lfb, table LB left as the adjoint's Lambda_m),
// 3 = heading (mode 0: o0 = L3; lane changes: o0 = Lft, table LT).  (The Lambda recurrence is the adjoint's; with LAMBDA it
// runs for every row, its result unread when the row has no gradient.)
struct ChainOut {
  float o0, o1;
};
template <bool NORM = false, bool LAMBDA = true>
PSTL_HD ChainOut stl_pre_chain(int which, const StlEnv& env, const StlRow& r, GeoPre pre, Scratch st) {
  const float tau = env.tau;
  const int mode = r.mode;
  const float ith = 1.0f / r.thmax;   // (as stl_eval_rec)
  float o0 = 0.0f, o1 = 0.0f;
  if (which == 0) {
    Lse gv1, gv2;
    gv1.init();
    gv2.init();
    PSTL_NOUNROLL
    for (int t = 0; t < kT; ++t) {
      const float v = pre.at(t, 3);
      gv1.add(-over<NORM>(v - r.vmin, r.vf) * tau);
      gv2.add(-over<NORM>(-v + r.vmax, r.vf) * tau);
    }
    o0 = gv1.value(), o1 = gv2.value();
  } else if (which == 1) {
    Lse gsafe;
    gsafe.init();
    PSTL_NOUNROLL
    for (int t = 0; t < kT; ++t) gsafe.add(-over<NORM>(pre.at(t, 0) - r.dsafe, r.sf) * tau);
    o0 = gsafe.value();
  } else if (mode == 0) {
    if (which == 2) {
      Lse g1, g2;
      g1.init();
      g2.init();
      PSTL_NOUNROLL
      for (int t = 0; t < kT; ++t) {
        const float d = pre.at(t, 1);
        const float s1 = over<NORM>(d - r.dmin, r.df), s2 = over<NORM>(-d + r.dmax, r.df);
        g1.add(-s1 * tau);
        g2.add(-s2 * tau);
      }
      o0 = g1.value(), o1 = g2.value();
    } else {
      Lse g3;
      g3.init();
      PSTL_NOUNROLL
      for (int t = 0; t < kT; ++t) g3.add(-((r.thmax - pre.at(t, 2)) * ith) * tau);
      o0 = g3.value();
    }
  } else {
    // one half of ReachAcc (the band or the heading term): first kFwin values parked, the tail as one running log-sum-exp,
    // the suffix scan, the outer soft-max, then the adjoint's Lambda recurrence over the same table
    const int base = which == 2 ? 0 : kFwin;
    Lse tail, f;
    tail.init();
    f.init();
    // (the values of the 20 steps are independent of one another: made in two unrolled batches of ten, so that their
    // exp / log latencies overlap, then fed to the table and the running tail in the sweep's order)
    PSTL_UNROLL
    for (int half = 0; half < 2; ++half) {
      float av[kFwin];
      PSTL_UNROLL
      for (int k = 0; k < kFwin; ++k) {
        const int t = half * kFwin + k;
        if (which == 2) {
          const float d = pre.at(t, 1);
          const float s1 = over<NORM>(d - r.dmin, r.df), s2 = over<NORM>(-d + r.dmax, r.df);
          const float band = -(lse2(-s1 * tau, -s2 * tau) / tau);
          av[k] = -band * tau;
        } else {
          av[k] = -((r.thmax - pre.at(t, 2)) * ith) * tau;
        }
      }
      PSTL_UNROLL
      for (int k = 0; k < kFwin; ++k) {
        if (half == 0) st.at(base + k) = av[k];
        else tail.add(av[k]);
      }
    }
    PSTL_NOUNROLL
    for (int k = kFwin - 1; k >= 0; --k) {
      tail.add(st.at(base + k));
      const float l = tail.value();
      st.at(base + k) = l;
      f.add(-(l / tau) * tau);
    }
    const float Lf = f.value();
    o0 = Lf;
    if (!LAMBDA) return ChainOut{o0, o1};   // (value-only callers: scoring)
    float sb = 0.0f, l_prev = 0.0f;
    PSTL_NOUNROLL
    for (int k = 0; k < kFwin; ++k) {
      const float l = st.at(base + k);
      const float q = PSTL_EXP(-l - Lf);
      sb = (k == 0) ? q : sb * PSTL_EXP(l - l_prev) + q;
      st.at(base + k) = l - PSTL_LOG(sb);
      l_prev = l;
    }
  }
  return ChainOut{o0, o1};
}

// part 1: the score from the chains' values (stl_eval_rec's tail) and d score / d V_i times the loss's derivative
// (dscore_fn(score), 0 = no gradient: om is not written then); returns the score
template <class DScoreFn>
PSTL_HD float adj_pre_weights(const StlEnv& env, int mode, AdjCtx& C, DScoreFn dscore_fn, float& dscore_in) {
  const float tau = env.tau;
  const float Lv1 = C.Lv1, Lv2 = C.Lv2, Ls = C.Ls;
  const float Vv1 = -(Lv1 / tau), Vv2 = -(Lv2 / tau), Vs = -(Ls / tau);
  float V[6];
  int n;
  float score;
  if (mode == 0) {
    const float v6[6] = {Vv1, Vv2, -(C.L1 / tau), -(C.L2 / tau), -(C.L3 / tau), Vs};
    score = conj6(v6, 6, tau);
    V[0] = -(Lv1 / tau), V[1] = -(Lv2 / tau), V[2] = -(C.L1 / tau), V[3] = -(C.L2 / tau), V[4] = -(C.L3 / tau), V[5] = -(Ls / tau);
    n = 6;
  } else {
    const float Lfb = C.L1, Lft = C.L3;
    const float v1[5] = {Vv1, Vv2, Lfb / tau, Lft / tau, Vs};
    score = conj6(v1, 5, tau);
    V[0] = -(Lv1 / tau), V[1] = -(Lv2 / tau), V[2] = Lfb / tau, V[3] = Lft / tau, V[4] = -(Ls / tau);
    n = 5;
  }
  dscore_in = dscore_fn(score);
  if (dscore_in == 0.0f) return score;
  const float Lout = -score * tau;
  PSTL_UNROLL
  for (int i = 0; i < 6; ++i) C.om[i] = i < n ? PSTL_EXP(-V[i] * tau - Lout) * dscore_in : 0.0f;
  return score;
}

// part 2: direct partials of the score with respect to (x, y, heading, speed) of step t >= 1, from the geometry slots of that step
template <bool NORM = false>
PSTL_HD void adj_pre_direct(const StlEnv& env, const StlRow& r, const AdjCtx& C, GeoPre pre, Scratch st, int t, float& gx,
                            float& gy, float& gth, float& gv) {
  const float tau = env.tau;
  const int mode = r.mode;
  const int LB = 0, LT = kFwin;
  const float inv_thmax = 1.0f / r.thmax;
  const float o_v1 = C.om[0], o_v2 = C.om[1], o_s = (mode == 0) ? C.om[5] : C.om[4];
  const float v = pre.at(t, 3);
  gv = o_v1 * PSTL_EXP(-over<NORM>(v - r.vmin, r.vf) * tau - C.Lv1) - o_v2 * PSTL_EXP(-over<NORM>(-v + r.vmax, r.vf) * tau - C.Lv2);
  if (NORM) gv = gv / r.vf;
  ClearHit ch;
  ch.dn = pre.at(t, 0), ch.d_dx = pre.at(t, 4), ch.d_dy = pre.at(t, 5), ch.d_dth = pre.at(t, 6);
  float gs = o_s * PSTL_EXP(-over<NORM>(ch.dn - r.dsafe, r.sf) * tau - C.Ls);
  if (NORM) gs = gs / r.sf;
  gx = gs * ch.d_dx;
  gy = gs * ch.d_dy;
  gth = gs * ch.d_dth;
  LaneHit h;
  h.d = pre.at(t, 1), h.th = pre.at(t, 7), h.dd_dx = pre.at(t, 8), h.dd_dy = pre.at(t, 9), h.dth_dth = pre.at(t, 10);
  const float s1 = over<NORM>(h.d - r.dmin, r.df), s2 = over<NORM>(-h.d + r.dmax, r.df), s3 = (r.thmax - h.th) * inv_thmax;
  float gd, gsth;
  if (mode == 0) {
    gd = C.om[2] * PSTL_EXP(-s1 * tau - C.L1) - C.om[3] * PSTL_EXP(-s2 * tau - C.L2);
    gsth = C.om[4] * PSTL_EXP(-s3 * tau - C.L3);
  } else {
    const float a1 = -s1 * tau, a2 = -s2 * tau;
    const float lp = lse2(a1, a2);
    const float ab = lp, a3 = -s3 * tau;
    const int m = t < kFwin - 1 ? t : kFwin - 1;
    const float wb = PSTL_EXP(ab - st.at(LB + m));
    const float wt = PSTL_EXP(a3 - st.at(LT + m));
    gd = C.om[2] * wb * (PSTL_EXP(a1 - lp) - PSTL_EXP(a2 - lp));
    gsth = C.om[3] * wt;
  }
  if (NORM) gd = gd / r.df;
  gx += gd * h.dd_dx;
  gy += gd * h.dd_dy;
  gth += gsth * (-inv_thmax) * h.dth_dth;
}

// part 3: lambda_t = direct_t + J_t^T lambda_{t+1}, t = T-1 ... 1; direct(t, gx, gy, gth, gv) hands back part 2's values,
// emit(t, gw, ga) takes the gradient with respect to the control pair of step t, t = T-1 ... 0
template <class DirectFn, class EmitFn>
PSTL_HD void adj_pre_costate(const StlEnv& env, GeoPre pre, float wscale, float ascale, DirectFn direct, EmitFn emit) {
  const float dt = env.dt;
  emit(kT - 1, 0.0f, 0.0f);  // the last control never reaches a scored state
  float lx = 0.0f, ly = 0.0f, lth = 0.0f, lv = 0.0f;
  PSTL_NOUNROLL
  for (int t = kT - 1; t >= 1; --t) {
    const float v = pre.at(t, 3), c = pre.at(t, 11), s = pre.at(t, 12);
    float gx, gy, gth, gv;
    direct(t, gx, gy, gth, gv);
    const float nlth = gth + lth + lx * (-(v * s) * dt) + ly * ((v * c) * dt);
    const float nlv = gv + lv + lx * (c * dt) + ly * (s * dt);
    lx = gx + lx;
    ly = gy + ly;
    lth = nlth;
    lv = nlv;
    emit(t - 1, lth * dt * wscale, lv * dt * ascale);
  }
}

template <bool NORM = false, bool WAVE_ZERO = false, class DScoreFn, class EmitFn>
PSTL_HD float stl_eval_grad(const StlEnv& env, const StlRow& r, const f4* lanes, const float* nei, int K, const float* s0,
                            const float* u, Scratch st, float wscale, float ascale, DScoreFn dscore_fn, EmitFn emit,
                            long us = 1, bool inert = false) {
  // inert: the caller knows that this row's score cannot reach the result (its loss weight is zero -- an invalid lane):
  // every gradient is exactly zero, so neither sweep is needed (the returned score is then meaningless).
  if (r.mode >= 3 || inert) {
    stl_grad_zero(u, us, emit);
    return 1.0f;
  }
  AdjState S;
  const float score = stl_grad_forward<NORM>(env, r, lanes, nei, K, s0, u, st, wscale, ascale, us, S);
  const float dscore_in = dscore_fn(score);
  // A lane is a row and a wavefront executes both sides of a branch its lanes disagree on: where SOME rows of a wavefront are
  // past their hinge and some are not, the per-row branch below runs the adjoint for the one AND the whole of stl_grad_zero for
  // the other, one after the other.  WAVE_ZERO (device code only; the callers whose emit() stores lane-contiguously -- the
  // traj-opt and --refinement loops, element-major work buffers): only a wavefront in which NO row has a gradient takes the zero
  // path; in a mixed one every row that ran the forward sweep walks the adjoint, a satisfied one with dscore_in = 0, and what
  // reaches emit() for it is the literal 0 the zero path hands in (same values: Adam on an exact zero leaves what the zero path
  // leaves).  Traj-opt loop +5 %.  NOT for the callers that scatter 16-byte pieces of row-major rows (k_guidance_iter, the
  // training adjoint): measured there (profiles/r6/wave_level_zero_decision_ab.txt), the launches got 1.7 % faster but wrote
  // TWICE the bytes to HBM (227 -> 455 MB per launch) -- with every row's stores spread over the adjoint's duration the partly
  // written lines of the rows in flight (~9 MB per XCD) no longer fit the 4 MB L2 and leave it piece by piece.
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (WAVE_ZERO) {
    if (__builtin_amdgcn_ballot_w64(dscore_in != 0.0f) == 0ull) {
      stl_grad_zero(u, us, emit);
      return score;
    }
    const bool has_grad = dscore_in != 0.0f;
    stl_grad_adjoint<NORM>(env, r, lanes, nei, K, u, st, wscale, ascale, us, S, dscore_in,
                           [&emit, has_grad](int t, float gw, float ga, float w, float a) {
                             emit(t, has_grad ? gw : 0.0f, has_grad ? ga : 0.0f, w, a);
                           });
    return score;
  }
#endif
  if (dscore_in == 0.0f) {   // e.g. a hinge loss on a satisfied row: the adjoint would produce exact zeros
    stl_grad_zero(u, us, emit);
    return score;
  }
  stl_grad_adjoint<NORM>(env, r, lanes, nei, K, u, st, wscale, ascale, us, S, dscore_in, emit);
  return score;
}

}  // namespace pstl
