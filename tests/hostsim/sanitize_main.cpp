// sanitize_main.cpp -- TEST INFRASTRUCTURE: the per-row STL math (csrc/stl_core.hpp) driven on deterministic
// pseudo-random scenes under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; GPU sanitizers are not
// available on the pool).  Covers: all-zero (invalid) side lanes, invalid all-zero neighbours, K = 1..8, every
// high-level mode including the outlier constant, tiny and huge controls, forward (all three formulas / selected)
// and the adjoint.  Exit code 0 = no report.
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <vector>

#include "../../pstl_diffusion_policy_amd/csrc/stl_core.hpp"

using namespace pstl;

static unsigned long long g_state = 0x9E3779B97F4A7C15ull;
static float urand() {
  g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((g_state >> 40) & 0xFFFFFF) / 16777216.0f;
}

int main() {
  int checked = 0;
  for (int K = 1; K <= 8; K += (K == 1 ? 1 : 2)) {   // the reference (torch.min over neighbours) needs K >= 1
    for (int rep = 0; rep < 40; ++rep) {
      std::vector<float> nei_raw((size_t)(K ? K : 1) * kT * 7, 0.0f), nei_prep((size_t)(K ? K : 1) * kT * kNeiPrep, 0.0f);
      for (int k = 0; k < K; ++k) {
        const bool valid = urand() < 0.6f;
        const float x0 = 10 + 40 * urand(), y0 = -4 + 8 * urand(), v = 3 + 4 * urand(), th = 0.1f * (urand() - 0.5f);
        for (int t = 0; t < kT; ++t) {
          float* p = &nei_raw[((size_t)k * kT + t) * 7];
          if (valid) {
            p[0] = 1, p[1] = x0 + v * 0.5f * t, p[2] = y0, p[3] = th, p[4] = v, p[5] = 4.5f, p[6] = 1.9f;
          }
          prep_neighbor(p, &nei_prep[((size_t)k * kT + t) * kNeiPrep]);
        }
      }
      std::vector<f4> lanes(3 * kNseg);
      for (int m = 0; m < 3; ++m) {
        const bool ok = (m == 0) || urand() < 0.6f;
        for (int j = 0; j < kNseg; ++j)
          lanes[m * kNseg + j] = ok ? f4{-5.0f + 60.0f * j / (kNseg - 1), (m == 1 ? 4.0f : m == 2 ? -4.0f : 0.0f), 0.0f, 0.0f}
                                    : f4{0.0f, 0.0f, 0.0f, 0.0f};
      }
      const float scale = (rep % 5 == 0) ? 50.0f : (rep % 5 == 1) ? 1e-6f : 0.2f;
      float s0[4] = {urand(), urand() - 0.5f, 0.05f * (urand() - 0.5f), 5 + 3 * urand()};
      alignas(16) float u[2 * kT];   // contiguous control rows are read 16 bytes at a time
      for (int i = 0; i < 2 * kT; ++i) u[i] = scale * (urand() - 0.5f);
      const StlEnv env = make_env(100.0f, 0.5f, 4.084f, 1.73f);
      for (int mode = 0; mode <= 3; ++mode) {
        const StlRow r = {s0[3] - 3, s0[3] + 3, -2.0f, 2.0f, 0.3f, 0.4f, mode};
        std::vector<float> sc3(kScratchFwd3), scg(kScratchGrad);
        float o3[3];
        const float a3 = stl_eval<true, -1>(env, r, lanes.data(), nei_prep.data(), K, DynSrc(s0, u, 1.0f, 1.0f, 0.5f),
                                            Scratch{sc3.data(), 1}, 0, o3, nullptr);
        const float a1 = stl_eval<false, -1>(env, r, lanes.data(), nei_prep.data(), K, DynSrc(s0, u, 1.0f, 1.0f, 0.5f),
                                             Scratch{sc3.data(), 1}, 0, nullptr, nullptr);
        float du[2 * kT];
        const float ag = stl_eval_grad(
            env, r, lanes.data(), nei_prep.data(), K, s0, u, Scratch{scg.data(), 1}, 1.0f, 1.0f, [](float) { return 1.0f; },
            [&](int t, float gw, float ga, float, float) {
              du[2 * t] = gw;
              du[2 * t + 1] = ga;
            });
        if (!(a3 == a1) || !(a1 == ag) || (mode == 3 && a1 != 1.0f)) {
          fprintf(stderr, "inconsistent scores K=%d rep=%d mode=%d: %g %g %g\n", K, rep, mode, a3, a1, ag);
          return 2;
        }
        if (mode < 3) {   // the same row through the parts of the latency layout: same score, same gradient, bit for bit
          std::vector<float> geo((size_t)kGeoSlots * kT), scp(kScratchGradPre);
          for (int q = 0; q < 10; ++q)
            stl_geometry(env, lanes.data() + mode * kNseg, nei_prep.data(), K, DynSrc(s0, u, 1.0f, 1.0f, 0.5f), 2 * q, 2 * q + 2,
                         geo.data(), 1);
          const GeoPre pre = {geo.data(), 1};
          const Scratch sp = {scp.data(), 1};
          AdjCtx C;
          const ChainOut c0 = stl_pre_chain(0, env, r, pre, sp), c1 = stl_pre_chain(1, env, r, pre, sp),
                         c2 = stl_pre_chain(2, env, r, pre, sp), c3 = stl_pre_chain(3, env, r, pre, sp);
          C.Lv1 = c0.o0, C.Lv2 = c0.o1, C.Ls = c1.o0, C.L1 = c2.o0, C.L2 = c2.o1, C.L3 = c3.o0;
          float dsc, dp[2 * kT], part[kT][4];
          const float ap = adj_pre_weights(env, mode, C, [](float) { return 1.0f; }, dsc);
          for (int t = kT - 1; t >= 1; --t) adj_pre_direct(env, r, C, pre, sp, t, part[t][0], part[t][1], part[t][2], part[t][3]);
          adj_pre_costate(
              env, pre, 1.0f, 1.0f,
              [&](int t, float& gx, float& gy, float& gth, float& gv) { gx = part[t][0], gy = part[t][1], gth = part[t][2], gv = part[t][3]; },
              [&](int t, float gw, float ga) {
                dp[2 * t] = gw;
                dp[2 * t + 1] = ga;
              });
          if (!(ap == ag) || memcmp(dp, du, sizeof(dp)) != 0) {
            fprintf(stderr, "the parts disagree with the fused sweeps K=%d rep=%d mode=%d: %g %g\n", K, rep, mode, ap, ag);
            return 4;
          }
        }
        for (int i = 0; i < 2 * kT; ++i)
          if (!(du[i] == du[i])) {
            fprintf(stderr, "NaN gradient K=%d rep=%d mode=%d i=%d\n", K, rep, mode, i);
            return 3;
          }
        ++checked;
      }
    }
  }
  printf("sanitize_main: %d row evaluations clean\n", checked);
  return 0;
}
