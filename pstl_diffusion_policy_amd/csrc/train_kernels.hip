// train_kernels.hip -- backward pass of RefineNet under the STL loss (the training step of
// config 5, reference nusc_train.py:1400-1427,1522-1525 and compute_policy_loss :370-478).
//
//   forward (pstl_refine_train_forward, mlp_kernels.hip) keeps h1 = relu(L1), h2 = relu(L2), pre = L3 output.
//   d loss / d rect_controls comes from pstl_stl_backward (the STL adjoint already used by guidance).
//   here: interval/tanh head backward (fused elementwise kernel), ReLU masks + column sums (fused, deterministic
//   two-stage reduction), per-scene reduction for the 224 scene-constant input columns, the three weight-gradient
//   contractions over the rows (dW3 = dO^T h2, dW2 = dH2^T h1, dW1x = dH1^T [hl|stlp|init]) as a hand-written split-K
//   fp32-MFMA kernel (k_wgrad: rocBLAS ran these K = 786 432, tiny-M-by-N shapes at 22 TFLOP/s, 4.7 ms each), and the two
//   activation-gradient products dH2 = (dO W3) * [h2 > 0], dH1 = (dH2 W2) * [h1 > 0] as k_dgrad: the transposed weights
//   register-stationary as split-bf16 MFMA A operands (the layout of the forward chain), the gradient rows staged
//   through LDS as B operands, ReLU mask and bias-gradient column sums fused into the epilogue.  No vendor BLAS.

#include "pstl_common.hpp"

namespace pstl {
namespace {

constexpr int kHid = PSTL_HID, kCtrl = PSTL_CTRL, kFeat = PSTL_FEAT;
constexpr int kX47 = 47;   // hl 1 | stlp 6 | init 40  = input columns 224..270 of rect_net layer 1
constexpr int kIn = kFeat + kX47;  // 271

// dO = dcontrols * [prev_score < 0] * d interval / d raw * (1 - raw^2), raw = tanh(pre)   (nusc_model.py:212-229)
// x47 = [hl | stlp | init]
// With the merge_net architecture (pooled != null) the last 40 input columns are init + pooled[scene, mode, shard].
__global__ void k_head_bwd(long N, float w_max, float a_max, const float* dctrl, const float* pre, const float* init,
                           const float* prev_scores, const float* hl, const float* stlp, const float* pooled, int S,
                           int n_shards, float* dO, float* x47) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * kCtrl) return;
  const long row = i / kCtrl;
  const int f = (int)(i % kCtrl);
  const float sc = (f & 1) ? a_max : w_max;
  const float raw = tanhf(pre[i]);
  const float in0 = init[i];
  const float slope = raw >= 0.0f ? (sc - in0) : (in0 - (-sc));
  const float viol = prev_scores[row] < 0.0f ? 1.0f : 0.0f;
  dO[i] = dctrl[i] * viol * slope * (1.0f - raw * raw);
  float fused = in0;
  if (pooled) {   // row = (b*S + s)*3 + mode ; shard = s / (S/n_shards)
    const long bs_ = row / 3;
    const int mode = (int)(row % 3), s_ = (int)(bs_ % S);
    const long b = bs_ / S;
    fused = in0 + pooled[((b * 3 + mode) * n_shards + s_ / (S / n_shards)) * kCtrl + f];
  }
  x47[row * kX47 + 7 + f] = fused;
  if (f == 0) x47[row * kX47] = hl[row];
  if (f < 6) x47[row * kX47 + 1 + f] = stlp[row * 6 + f];
}

// Column sums of the 40-column head gradient (no mask): 25 row-lanes x 10 column quads per block, four rows per row-lane
// in flight, row-lanes added in a fixed order.
__global__ __launch_bounds__(256) void k_colsum40(long N, const float* G, float* partial) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int q = threadIdx.x % 10, rl = threadIdx.x / 10;
  long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  rows_per_block = (rows_per_block + 99) / 100 * 100;
  const long r0 = blockIdx.x * rows_per_block, r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  f4 acc = f4{0.0f, 0.0f, 0.0f, 0.0f};
  if (rl < 25)
    for (long r = r0 + rl; r < r1; r += 100) {
      f4 g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        g[u] = (r + 25 * u < r1) ? *reinterpret_cast<const f4*>(G + (r + 25 * u) * kCtrl + 4 * q) : f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += g[u];
    }
  __shared__ f4 red[25][10];
  if (rl < 25) red[rl][q] = acc;
  __syncthreads();
  if (threadIdx.x < 10) {
    f4 t = red[0][q];
    for (int i = 1; i < 25; ++i) t += red[i][q];
    *reinterpret_cast<f4*>(partial + (long)blockIdx.x * kCtrl + 4 * q) = t;
  }
}

// out[c] = sum over the blocks of partial[b][c]: one 64-lane block per column, lane l adds blocks l, l+64, ... and the
// 64 lane sums are added in lane order (fixed order: deterministic)
__global__ __launch_bounds__(64) void k_colsum_final(int nblocks, int ncol, const float* partial, float* out) {
  const int c = blockIdx.x, l = threadIdx.x;
  float acc = 0.0f;
  for (int b = l; b < nblocks; b += 64) acc += partial[(long)b * ncol + c];
  __shared__ float red[64];
  red[l] = acc;
  __syncthreads();
  if (l == 0) {
    float t = red[0];
    for (int i = 1; i < 64; ++i) t += red[i];
    out[c] = t;
  }
}

// S[scene][f] = sum over the scene's rows of G[row][f]
__global__ void k_scene_sum(int rows_per_scene, const float* G, float* S) {
  const long b = blockIdx.x;
  const int f = threadIdx.x;
  float acc = 0.0f;
  int r = 0;
  for (; r + 8 <= rows_per_scene; r += 8) {   // eight loads in flight, added in row order (the same sum as a plain loop)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = G[(b * rows_per_scene + r + u) * kHid + f];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; r < rows_per_scene; ++r) acc += G[(b * rows_per_scene + r) * kHid + f];
  S[b * kHid + f] = acc;
}

// dscore[r] = -grad_scale * valid[r] * [thres - score[r] > 0] ; loss_parts[block] = sum relu(thres - score) * valid
__global__ void k_loss_grad(long N, const float* scores, const float* valid, float thres, float grad_scale, float* dscore,
                            float* loss_parts) {
  __shared__ float red[256];
  float acc = 0.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const float m = thres - scores[i];
    const float v = valid[i];
    dscore[i] = m > 0.0f ? -(grad_scale * v) : 0.0f;
    acc += fmaxf(m, 0.0f) * v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_parts[blockIdx.x] = red[0];
}

// ---- DPP diversity loss of e7 training (reference nusc_train.py:442-464) ----------------------------------------------
// One wavefront per (scene, mode, shard) group of n = S/n_shards samples.  x_i = rect_controls_i / (w_max, a_max);
// L = diag(q) exp(-scale * |x_i - x_j|) diag(q), q_i = exp(score_i) [score_i > 0] (or [score_i > 0] with
// --diverse_detach); M = (L + I)^-1; diversity = tr(I - M); loss = weight * mean_g(-diversity).
// L + I is symmetric positive definite with eigenvalues >= 1, so the in-place Gauss-Jordan inversion below needs no
// pivoting; it runs in float64 (full-rate on MI355X), lane j owning column j.  Backward, with c = weight / #groups:
//   d loss / d L = -c (M M)^T ;  d/d sim_ij = (.)_ij q_i q_j ;  d/d q_i = sum_j ((.)_ij + (.)_ji) sim_ij q_j ;
//   d/d dist_ij = -scale sim_ij d/d sim_ij ;  d/d x_i = sum_j (d/d dist_ij + d/d dist_ji) (x_i - x_j)/dist_ij  (0 at dist 0,
//   as torch.norm's backward defines it) ;  d/d score_i = d/d q_i * q_i   (0 with --diverse_detach).
constexpr int kDppWave = 64;
// LDS of one group: A, Q (n x (n+1) doubles), X (n x 40), SIM, DST (n x (n+1)), q (n, padded to even), trace terms (n doubles)
__host__ __device__ inline size_t dpp_group_bytes(int n) {
  const size_t ld = n + 1;
  size_t b = 2 * n * ld * sizeof(double) + ((size_t)n * kCtrl + 2 * n * ld + ((n + 1) & ~1)) * sizeof(float) + n * sizeof(double);
  return (b + 15) & ~(size_t)15;
}

struct DppArgs {
  int bs, S, n_shards;
  float w_max, a_max, scale, c;
  int detach;
  const float* rect;     // (N,40)
  const float* scores;   // (N,)
  float* group_div;      // (bs*3*n_shards,)
  float* dcontrols;      // (N,40) written
  float* dscore;         // (N,) written
};

// A wavefront carries gpw = 64 / n groups side by side (n = 16: four; lane = sub * n + l, l = the group's column / row),
// each with its own LDS block: with one group per wavefront three quarters of the lanes of the default shape idled.
__global__ __launch_bounds__(64) void k_dpp(DppArgs a, int n_groups) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = a.S / a.n_shards;
  const int ld = n + 1;                                   // padded leading dimension (doubles)
  const int gpw = kDppWave / n;                            // groups per wavefront
  const size_t per = dpp_group_bytes(n);
  const int sub = threadIdx.x / n, l = threadIdx.x % n;
  const int g = blockIdx.x * gpw + sub;
  const bool own = sub < gpw && g < n_groups;
  unsigned char* mine = smem + (size_t)(sub < gpw ? sub : 0) * per;
  double* A = reinterpret_cast<double*>(mine);            // n x ld : L + I, then M
  double* Q = A + n * ld;                                 // n x ld : M M
  float* X = reinterpret_cast<float*>(Q + n * ld);        // n x 40 normalised controls
  float* SIM = X + n * kCtrl;                             // n x ld
  float* DST = SIM + n * ld;                              // n x ld
  float* qv = DST + n * ld;                               // n
  double* trs = reinterpret_cast<double*>(qv + ((n + 1) & ~1));   // n: the diagonal terms of the trace
  const int gg = own ? g : 0;
  const int shard = gg % a.n_shards, bm = gg / a.n_shards, mode = bm % 3, b = bm / 3;
  const long row = ((long)b * a.S + shard * n + l) * 3 + mode;
  float score = 0.0f, q = 0.0f;
  if (own) {
    score = a.scores[row];
    const float pos = score > 0.0f ? 1.0f : 0.0f;
    q = a.detach ? pos : expf(score) * pos;
    qv[l] = q;
    const float* src = a.rect + row * kCtrl;
#pragma unroll 8
    for (int f = 0; f < kCtrl; ++f) X[l * kCtrl + f] = src[f] / ((f & 1) ? a.a_max : a.w_max);
  }
  __syncthreads();
  if (own) {   // column j = l
    for (int i = 0; i < n; ++i) {
      float acc = 0.0f;
      for (int f = 0; f < kCtrl; ++f) {
        const float d = X[i * kCtrl + f] - X[l * kCtrl + f];
        acc += d * d;
      }
      const float dist = sqrtf(acc);
      const float sim = expf(-a.scale * dist);
      DST[i * ld + l] = dist;
      SIM[i * ld + l] = sim;
      A[i * ld + l] = (double)((qv[i] * sim) * q) + (i == l ? 1.0 : 0.0);
    }
  }
  __syncthreads();
  // in-place Gauss-Jordan inversion: after step k, column/row k of A hold those of the partial inverse
  for (int k = 0; k < n; ++k) {
    const double p = A[k * ld + k];
    __syncthreads();
    if (own) A[k * ld + l] = (l == k ? 1.0 : A[k * ld + l]) / p;
    __syncthreads();
    if (own) {
      const double rk = A[k * ld + l];
      for (int i = 0; i < n; ++i) {
        if (i == k) continue;
        const double f = A[i * ld + k];                 // column k is only rewritten by lane k, after it has read f
        const double cur = (l == k) ? 0.0 : A[i * ld + l];
        A[i * ld + l] = cur - f * rk;
      }
    }
    __syncthreads();
  }
  if (own) {   // Q = M M (column j = l), trace
    for (int i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int k = 0; k < n; ++k) acc += A[i * ld + k] * A[k * ld + l];
      Q[i * ld + l] = acc;
    }
    trs[l] = 1.0 - A[l * ld + l];
  }
  __syncthreads();
  if (own && l == 0) {
    double tr = 0.0;
    for (int i = 0; i < n; ++i) tr += trs[i];
    a.group_div[g] = (float)tr;
  }
  if (own) {   // row i = l
    const int i = l;
    double dq = 0.0;
    float dx[kCtrl];
#pragma unroll
    for (int f = 0; f < kCtrl; ++f) dx[f] = 0.0f;
    for (int j = 0; j < n; ++j) {
      const double gij = -(double)a.c * Q[j * ld + i], gji = -(double)a.c * Q[i * ld + j];   // (M M)^T_ij = Q_ji
      const float sim = SIM[i * ld + j], dist = DST[i * ld + j];
      dq += (gij + gji) * (double)sim * (double)qv[j];
      if (dist != 0.0f) {
        const double dd = (gij + gji) * (double)q * (double)qv[j] * (-(double)a.scale * (double)sim);   // both orientations
        const float coef = (float)(dd / (double)dist);
#pragma unroll
        for (int f = 0; f < kCtrl; ++f) dx[f] += coef * (X[i * kCtrl + f] - X[j * kCtrl + f]);
      }
    }
    float* out = a.dcontrols + row * kCtrl;
#pragma unroll
    for (int f = 0; f < kCtrl; ++f) out[f] = dx[f] / ((f & 1) ? a.a_max : a.w_max);
    a.dscore[row] = a.detach ? 0.0f : (float)(dq * (double)q);
  }
}

// loss_reg = mask_mean(square(rect - init), [score >= 0]) (reference nusc_train.py:466): partial sums per block of
// (sum of squares over masked rows, number of masked rows); the gradient kernel needs the masked-row count first.
__global__ void k_reg_partials(long N, const float* rect, const float* init, const float* scores, double* parts) {
  __shared__ double r0[256], r1[256];
  double sq = 0.0, cnt = 0.0;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (long)gridDim.x * blockDim.x) {
    if (scores[r] >= 0.0f) {
      cnt += 1.0;
      for (int f = 0; f < kCtrl; ++f) {
        const float d = rect[r * kCtrl + f] - init[r * kCtrl + f];
        sq += (double)(d * d);
      }
    }
  }
  r0[threadIdx.x] = sq;
  r1[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      r0[threadIdx.x] += r0[threadIdx.x + o];
      r1[threadIdx.x] += r1[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    parts[2 * blockIdx.x] = r0[0];
    parts[2 * blockIdx.x + 1] = r1[0];
  }
}
__global__ void k_reg_final(int nblocks, long N, double* parts, float* reg_out /* [0] loss_reg, [1] 1/clip(mean m) */) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double sq = 0.0, cnt = 0.0;
  for (int b = 0; b < nblocks; ++b) {
    sq += parts[2 * b];
    cnt += parts[2 * b + 1];
  }
  const float mean_m = (float)(cnt / (double)N);
  const float den = fmaxf(mean_m, 1e-2f);
  reg_out[0] = (float)(sq / ((double)N * kCtrl)) / den;
  reg_out[1] = 1.0f / den;
}
// dcontrols += weight * 2 (rect - init) [score >= 0] / (N*40) / clip(mean m)
__global__ void k_reg_grad(long N, float weight, const float* rect, const float* init, const float* scores,
                           const float* reg_out, float* dcontrols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * kCtrl) return;
  const long r = i / kCtrl;
  if (scores[r] >= 0.0f)
    dcontrols[i] += (weight * reg_out[1] / (float)(N * kCtrl)) * (2.0f * (rect[i] - init[i]));
}

constexpr int kRedBlocks = 512;
constexpr long kWtPackWords = 16L * 8 * 8 * 64;   // split-bf16 A operands of one transposed 256 x 256 weight matrix (k_dgrad)

// ---- weight gradient: D[f][c] = sum over rows of G[row][f] * H[row][c]  (split-K over workgroups) -------------------
// v_mfma_f32_16x16x4_f32 with the ROW index as the contraction: A[i = f][k = row], B[k = row][j = c], 16 rows (4 k-steps)
// per staged chunk.  FT x CT output tiles of 16x16 are spread over 8 waves as WF x WC; every wave keeps its
// (FT/WF) x (CT/WC) accumulator tiles in registers for its whole row range, then writes one partial slab;
// k_slab_reduce adds the slabs in a fixed order (deterministic).  Chunks are staged global -> registers -> LDS with the
// next chunk's loads in flight during the MFMAs; LDS rows are padded so that the 4 rows a ds_read_b32 touches fall
// into different bank halves.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kWgRows = 16;

template <int FT, int CT, int WF, int WC>
__global__ __launch_bounds__(512, 2) void k_wgrad(long N, const float* G, int ldg, int fvalid, const float* H, int ldh,
                                                  int cvalid, float* slabs) {
  static_assert(WF * WC == 8 && FT % WF == 0 && CT % WC == 0, "tile split");
  constexpr int F = 16 * FT, C = 16 * CT;
  constexpr int SG = (F % 32 == 16) ? F : F + 16, SH = (C % 32 == 16) ? C : C + 16;   // padded LDS row strides
  constexpr int MF = FT / WF, MC = CT / WC;                                           // tiles per wave
  constexpr int VG = (kWgRows * F / 4 + 511) / 512, VH = (kWgRows * C / 4 + 511) / 512;  // float4 loads per thread
  __shared__ __attribute__((aligned(16))) float lg[2][kWgRows * SG];
  __shared__ __attribute__((aligned(16))) float lh[2][kWgRows * SH];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wf = w / WC, wc = w % WC;
  const int li = lane & 15, kk = lane >> 4;
  const long n_chunks = (N + kWgRows - 1) / kWgRows;
  const long per = (n_chunks + gridDim.x - 1) / gridDim.x;
  const long c0 = blockIdx.x * per, c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;

  f32x4 acc[MF][MC];
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < MC; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  f32x4 rg[VG], rh[VH];
  auto load_chunk = [&](long ch) {
#pragma unroll
    for (int v = 0; v < VG; ++v) {
      const int e = tid + v * 512;                 // float4 index inside the chunk: (row, col4)
      const int r = e / (F / 4), c4 = e % (F / 4);
      const long row = ch * kWgRows + r;
      f32x4 val = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (e < kWgRows * F / 4 && row < N) {
        const float* src = G + row * ldg + 4 * c4;
        if (4 * c4 + 3 < fvalid && (ldg % 4 == 0)) {
          val = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) val[q] = (4 * c4 + q < fvalid) ? src[q] : 0.0f;
        }
      }
      rg[v] = val;
    }
#pragma unroll
    for (int v = 0; v < VH; ++v) {
      const int e = tid + v * 512;
      const int r = e / (C / 4), c4 = e % (C / 4);
      const long row = ch * kWgRows + r;
      f32x4 val = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (e < kWgRows * C / 4 && row < N) {
        const float* src = H + row * ldh + 4 * c4;
        if (4 * c4 + 3 < cvalid && (ldh % 4 == 0)) {
          val = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) val[q] = (4 * c4 + q < cvalid) ? src[q] : 0.0f;
        }
      }
      rh[v] = val;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int v = 0; v < VG; ++v) {
      const int e = tid + v * 512;
      if (e < kWgRows * F / 4) *reinterpret_cast<f32x4*>(&lg[buf][(e / (F / 4)) * SG + 4 * (e % (F / 4))]) = rg[v];
    }
#pragma unroll
    for (int v = 0; v < VH; ++v) {
      const int e = tid + v * 512;
      if (e < kWgRows * C / 4) *reinterpret_cast<f32x4*>(&lh[buf][(e / (C / 4)) * SH + 4 * (e % (C / 4))]) = rh[v];
    }
  };

  if (c0 < c1) {
    load_chunk(c0);
    store_chunk(0);
  }
  __syncthreads();
  int buf = 0;
  for (long ch = c0; ch < c1; ++ch) {
    if (ch + 1 < c1) load_chunk(ch + 1);          // in flight during the MFMAs below
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float av[MF], bv[MC];
#pragma unroll
      for (int a = 0; a < MF; ++a) av[a] = lg[buf][(4 * s4 + kk) * SG + 16 * (wf * MF + a) + li];
#pragma unroll
      for (int b = 0; b < MC; ++b) bv[b] = lh[buf][(4 * s4 + kk) * SH + 16 * (wc * MC + b) + li];
#pragma unroll
      for (int a = 0; a < MF; ++a)
#pragma unroll
        for (int b = 0; b < MC; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    if (ch + 1 < c1) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial slab of this workgroup: [F][C] row-major; accumulator tile: row f = 16 ft + 4 (lane>>4) + r, col = 16 ct + (lane&15)
  float* out = slabs + (long)blockIdx.x * F * C;
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < MC; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(16 * (wf * MF + a) + 4 * kk + r) * C + 16 * (wc * MC + b) + li] = acc[a][b][r];
}

// D[f][c] (row-major, leading dimension ldd) = sum over slabs, for f < fvalid, c < cvalid
__global__ void k_slab_reduce(int nslabs, int F, int C, int fvalid, int cvalid, const float* slabs, float* D, int ldd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * C) return;
  const int f = i / C, c = i % C;
  if (f >= fvalid || c >= cvalid) return;
  // eight loads in flight, added in slab order (the same sum as a plain loop; as a plain loop every add waited for its
  // own load: 0.107 ms per 256 x 256 reduction of 512 slabs)
  float acc = 0.0f;
  int s = 0;
  for (; s + 8 <= nslabs; s += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slabs[(long)(s + u) * F * C + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; s < nslabs; ++s) acc += slabs[(long)s * F * C + i];
  D[(long)f * ldd + c] = acc;
}

template <int FT, int CT, int WF, int WC>
int wgrad(long N, const float* G, int ldg, int fvalid, const float* H, int ldh, int cvalid, float* slabs, float* D, int ldd,
          hipStream_t st) {
  const long n_chunks = (N + kWgRows - 1) / kWgRows;
  const int nb = (int)(n_chunks < kRedBlocks ? n_chunks : kRedBlocks);
  hipLaunchKernelGGL((k_wgrad<FT, CT, WF, WC>), dim3(nb), dim3(512), 0, st, N, G, ldg, fvalid, H, ldh, cvalid, slabs);
  hipLaunchKernelGGL(k_slab_reduce, dim3((16 * FT * 16 * CT + 255) / 256), dim3(256), 0, st, nb, 16 * FT, 16 * CT, fvalid,
                     cvalid, slabs, D, ldd);
  return launch_status();
}

// ---- the same contraction on the bfloat16 matrix pipe (round 3) ----------------------------------------------------------
// Every fp32 operand as two bfloat16 pieces (hi = bf16(v), lo = bf16(v - hi): 2^-17 per operand, fp32's exponent range --
// the gradients are ~1e-7), three v_mfma_f32_16x16x32_bf16 per fp32 product (hi hi + lo hi + hi lo), fp32 accumulators: the
// scheme of k_dgrad and of the forward chain's bfloat16 form, against a 5e-3 gate on the gradients.  The contraction index is
// the ROW: a 32-row chunk is one k-block.  Staging turns the row-major chunk into MFMA operand order on the way into LDS: a
// thread takes a 4-row x 4-column block (four 16-byte loads, rows of a wave coalesced), splits it, and writes per column the
// 4 rows' pieces as one 8-byte store to [tile = col / 16][g = row / 8][i = col % 16][row % 8] -- the slot a lane (i, g) of
// the MFMA reads as ONE ds_read_b128 (tile blocks padded by 16 bytes: the stores of a wave otherwise fall 16-fold into the
// same banks).  Double-buffered, next chunk's loads in flight during the MFMAs; the pieces of G serve as A operands, those of
// H as B operands, same k order on both sides.  Slabs and their fixed-order reduction as above (deterministic).
typedef __bf16 wg_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf4 __attribute__((ext_vector_type(4)));
constexpr int kWbRows = 32;
constexpr int kWbTile = 1024 + 16;   // bytes of one (16 columns x 32 rows) tile block of 2-byte pieces, padded

template <int FT, int CT>
constexpr size_t wgrad_bf_lds() { return (size_t)2 * (2 * FT + 2 * CT) * kWbTile; }

template <int FT, int CT, int WF, int WC>
__global__ __launch_bounds__(512, 1) void k_wgrad_bf(long N, const float* G, int ldg, int fvalid, const float* H, int ldh,
                                                     int cvalid, float* slabs) {
  static_assert(WF * WC == 8 && FT % WF == 0 && CT % WC == 0, "tile split");
  constexpr int F = 16 * FT, C = 16 * CT;
  constexpr int MF = FT / WF, MC = CT / WC;
  constexpr int IG = 8 * (F / 4), IH = 8 * (C / 4);                  // 4x4 blocks per chunk
  constexpr int VG = (IG + 511) / 512, VH = (IH + 511) / 512;        // ... per thread
  extern __shared__ __attribute__((aligned(16))) char wlds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wf = w / WC, wc = w % WC;
  const int li = lane & 15, kk = lane >> 4;
  const long n_chunks = (N + kWbRows - 1) / kWbRows;
  const long per = (n_chunks + gridDim.x - 1) / gridDim.x;
  const long c0 = blockIdx.x * per, c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;
  auto tile = [&](int buf, int t) { return wlds + ((size_t)buf * (2 * FT + 2 * CT) + t) * kWbTile; };   // t: G hi | G lo | H hi | H lo

  f32x4 acc[MF][MC];
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < MC; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  f32x4 rg[VG][4], rh[VH][4];
  auto load_block = [&](const float* X, int ld, int valid, int W, long ch, int item, f32x4 (&dst)[4]) {
    const int rgp = item / (W / 4), c4 = item % (W / 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long row = ch * kWbRows + 4 * rgp + q;
      f32x4 val = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      if (item < 8 * (W / 4) && row < N) {
        const float* src = X + row * ld + 4 * c4;
        if (4 * c4 + 3 < valid && (ld % 4 == 0)) {
          val = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = (4 * c4 + e < valid) ? src[e] : 0.0f;
        }
      }
      dst[q] = val;
    }
  };
  auto store_block = [&](int buf, int thi, int NT_, int W, int item, const f32x4 (&src)[4]) {
    if (item >= 8 * (W / 4)) return;
    const int rgp = item / (W / 4), c4 = item % (W / 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = 4 * c4 + j;
      wg_bf4 hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        hi[q] = (__bf16)src[q][j];
        lo[q] = (__bf16)(src[q][j] - (float)hi[q]);
      }
      const int off = ((rgp >> 1) * 16 + (col & 15)) * 16 + (rgp & 1) * 8;
      *reinterpret_cast<wg_bf4*>(tile(buf, thi + (col >> 4)) + off) = hi;
      *reinterpret_cast<wg_bf4*>(tile(buf, thi + NT_ + (col >> 4)) + off) = lo;
    }
  };
  auto load_chunk = [&](long ch) {
#pragma unroll
    for (int v = 0; v < VG; ++v) load_block(G, ldg, fvalid, F, ch, tid + v * 512, rg[v]);
#pragma unroll
    for (int v = 0; v < VH; ++v) load_block(H, ldh, cvalid, C, ch, tid + v * 512, rh[v]);
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int v = 0; v < VG; ++v) store_block(buf, 0, FT, F, tid + v * 512, rg[v]);
#pragma unroll
    for (int v = 0; v < VH; ++v) store_block(buf, 2 * FT, CT, C, tid + v * 512, rh[v]);
  };

  if (c0 < c1) {
    load_chunk(c0);
    store_chunk(0);
  }
  __syncthreads();
  int buf = 0;
  for (long ch = c0; ch < c1; ++ch) {
    if (ch + 1 < c1) load_chunk(ch + 1);          // in flight during the MFMAs below
    wg_bf8 bh[MC], bl[MC];
#pragma unroll
    for (int b = 0; b < MC; ++b) {
      bh[b] = *reinterpret_cast<const wg_bf8*>(tile(buf, 2 * FT + wc * MC + b) + lane * 16);
      bl[b] = *reinterpret_cast<const wg_bf8*>(tile(buf, 2 * FT + CT + wc * MC + b) + lane * 16);
    }
#pragma unroll
    for (int a = 0; a < MF; ++a) {
      const wg_bf8 ah = *reinterpret_cast<const wg_bf8*>(tile(buf, wf * MF + a) + lane * 16);
      const wg_bf8 al = *reinterpret_cast<const wg_bf8*>(tile(buf, FT + wf * MF + a) + lane * 16);
#pragma unroll
      for (int b = 0; b < MC; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b], acc[a][b], 0, 0, 0);
      }
    }
    if (ch + 1 < c1) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  float* out = slabs + (long)blockIdx.x * F * C;
#pragma unroll
  for (int a = 0; a < MF; ++a)
#pragma unroll
    for (int b = 0; b < MC; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(16 * (wf * MF + a) + 4 * kk + r) * C + 16 * (wc * MC + b) + li] = acc[a][b][r];
}

// Used for dW2 = dH2^T h1 (256 x 256 over N rows): 0.81 -> 0.34 ms per 786 432 rows = 1.6 GB of operands at 4.7 TB/s, i.e. the
// contraction now runs at the speed its two inputs stream from HBM.  The thin ones (dW3: 40 x 256, dW1's 47 columns) were
// already there on the exact fp32 kernel (0.19 ms for 0.93 GB) and stay on it, as do the short contractions of the scene
// encoders and the per-scene sums.
template <int FT, int CT, int WF, int WC>
int wgrad_bf(long N, const float* G, int ldg, int fvalid, const float* H, int ldh, int cvalid, float* slabs, float* D, int ldd,
             hipStream_t st) {
  const long n_chunks = (N + kWbRows - 1) / kWbRows;
  const int nb = (int)(n_chunks < kRedBlocks ? n_chunks : kRedBlocks);
  auto fn = k_wgrad_bf<FT, CT, WF, WC>;
  const size_t lds = wgrad_bf_lds<FT, CT>();
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PSTL_ERR_LAUNCH;
  hipLaunchKernelGGL(fn, dim3(nb), dim3(512), lds, st, N, G, ldg, fvalid, H, ldh, cvalid, slabs);
  hipLaunchKernelGGL(k_slab_reduce, dim3((16 * FT * 16 * CT + 255) / 256), dim3(256), 0, st, nb, 16 * FT, 16 * CT, fvalid,
                     cvalid, slabs, D, ldd);
  return launch_status();
}

// ---- activation gradients: out[row][f] = [H[row][f] > 0] * sum_k G[row][k] * W[k][f],  f < 256 -----------------------
// (dH2 = dO W3 with K = 40, dH1 = dH2 W2 with K = 256; W is the layer's (out = K, in = 256) weight matrix, so the product
// runs over the layer's OUTPUT index.)  Same scheme as the forward chain kernel (mlp_kernels.hip): eight waves, wave w
// owns output features [32 w, 32 w + 32) and keeps its slice of W^T in registers as MFMA A operands for the whole
// launch; a tile of 16 gradient rows is the B operand.  Arithmetic: every fp32 operand as two bfloat16 pieces
// (hi = bf16(v), lo = bf16(v - hi)), three v_mfma_f32_16x16x32_bf16 products per fp32 product, fp32 accumulation --
// operands good to 2^-17, far inside the 5e-3 the gradients are held to, and bfloat16 keeps fp32's exponent range
// (gradients here are ~1e-7: half pieces would underflow).  The tile's pieces are made once (each thread splits the
// 16-byte quads it loaded) and laid out in LDS in B-operand order, double buffered: the loads of tile t + 1 are in
// flight during the MFMAs of tile t.  Epilogue per lane: four consecutive features of one row -- the ReLU mask from H
// (16-byte load), the 16-byte store, and a running column sum whose 16 row-lanes are added in a fixed order at the end
// (partial[block][f]; k_colsum_final adds the blocks: deterministic).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bf16_pair(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}

// A operands of W^T: word m of lane l for block (T, kb) at dst[(((T*nkb + kb)*2 + (m>>2))*64 + l)*4 + (m&3)], m < 4 the hi
// pieces of slots 2m, 2m+1, m >= 4 the lo pieces; slot s of lane group g = l>>4 is k = 32 kb + 16 (s>>2) + 4 g + (s&3),
// the lane's feature is f = 16 T + (l&15); element = W[k*ld + f] (k < kvalid and f < fvalid, else 0)
__global__ void k_pack_wt_bf(const float* W, int ld, int kvalid, int fvalid, int nkb, unsigned* dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16L * nkb * 512) return;
  const int lane = (int)((i >> 2) & 63), m = (int)((i & 3) | (((i >> 8) & 1) << 2));
  const long tk = i >> 9;
  const int kb = (int)(tk % nkb), T = (int)(tk / nkb);
  const int f = 16 * T + (lane & 15), g = lane >> 4;
  float v[2];
  for (int e = 0; e < 2; ++e) {
    const int sl = 2 * (m & 3) + e;
    const int k = 32 * kb + 16 * (sl >> 2) + 4 * g + (sl & 3);
    const float wv = (k < kvalid && f < fvalid) ? W[(long)k * ld + f] : 0.0f;
    const float hi = (float)(__bf16)wv;
    v[e] = m < 4 ? hi : wv - hi;
  }
  dst[i] = bf16_pair(v[0], v[1]);
}

template <int NKB>
__global__ __launch_bounds__(512, 2) void k_dgrad(long N, const float* G, int ldg, int kvalid, const unsigned* Wp,
                                                  const float* H, float* out, int ldo, int fvalid, float* partial) {
  static_assert(NKB <= 8, "one staging wave per k-block");
  __shared__ __attribute__((aligned(16))) u32x4 pieces[2][NKB * 2 * 64];   // [buffer][kb][hi | lo][lane]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  bf16x8 wh[2][NKB], wl[2][NKB];
  {
    const u32x4* q4 = reinterpret_cast<const u32x4*>(Wp);
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const long blk = (long)(2 * w + ot) * NKB + kb;
        wh[ot][kb] = __builtin_bit_cast(bf16x8, q4[(blk * 2 + 0) * 64 + lane]);
        wl[ot][kb] = __builtin_bit_cast(bf16x8, q4[(blk * 2 + 1) * 64 + lane]);
      }
  }
  const long n_tiles = (N + 15) / 16;
  // Staging: wave kb (< NKB) makes k-block kb of the tile; its lane (g, c) loads the two 16-byte quads of row c that
  // are the lane's own B operand -- columns 32 kb + 4 g .. +3 (slots 0..3) and 32 kb + 16 + 4 g .. +3 (slots 4..7) --
  // and writes its hi and lo pieces as two lane-linear 16-byte LDS stores (conflict-free).
  f32x4 rq[2];
  auto load_tile = [&](long t) {
    rq[0] = rq[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (w < NKB) {
      const long row = t * 16 + c;
      if (row < N) {
        const int k0 = 32 * w + 4 * g;
        if (k0 < kvalid) rq[0] = *reinterpret_cast<const f32x4*>(G + row * ldg + k0);
        if (k0 + 16 < kvalid) rq[1] = *reinterpret_cast<const f32x4*>(G + row * ldg + k0 + 16);
      }
    }
  };
  auto store_tile = [&](int buf) {
    if (w < NKB) {
      float hi[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) hi[q] = (float)(__bf16)rq[0][q], hi[4 + q] = (float)(__bf16)rq[1][q];
      const u32x4 ph = u32x4{bf16_pair(hi[0], hi[1]), bf16_pair(hi[2], hi[3]), bf16_pair(hi[4], hi[5]), bf16_pair(hi[6], hi[7])};
      const u32x4 pl = u32x4{bf16_pair(rq[0][0] - hi[0], rq[0][1] - hi[1]), bf16_pair(rq[0][2] - hi[2], rq[0][3] - hi[3]),
                             bf16_pair(rq[1][0] - hi[4], rq[1][1] - hi[5]), bf16_pair(rq[1][2] - hi[6], rq[1][3] - hi[7])};
      pieces[buf][(2 * w) * 64 + lane] = ph;
      pieces[buf][(2 * w + 1) * 64 + lane] = pl;
    }
  };
  f32x4 cs[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
  long t = blockIdx.x;
  if (t < n_tiles) {
    load_tile(t);
    store_tile(0);
  }
  __syncthreads();
  int buf = 0;
  for (; t < n_tiles; t += gridDim.x) {
    const long tn = t + gridDim.x;
    if (tn < n_tiles) load_tile(tn);               // in flight during the MFMAs below
    // this lane's mask source: features 16 (2w + ot) + 4g .. +3 of row t*16 + c
    const long row = t * 16 + c;
    f32x4 hm[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
      hm[ot] = !H ? f32x4{1.0f, 1.0f, 1.0f, 1.0f}     // no mask (the product feeds a linear input, not a ReLU)
               : row < N ? *reinterpret_cast<const f32x4*>(H + row * kHid + 16 * (2 * w + ot) + 4 * g)
                         : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 acc[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
    const u32x4* pb = &pieces[buf][lane];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, pb[(2 * kb) * 64]), bl = __builtin_bit_cast(bf16x8, pb[(2 * kb + 1) * 64]);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ot][kb], bh, acc[ot], 0, 0, 0);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ot][kb], bh, acc[ot], 0, 0, 0);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ot][kb], bl, acc[ot], 0, 0, 0);
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = hm[ot][r] > 0.0f ? acc[ot][r] : 0.0f;
      if (row < N && 16 * (2 * w + ot) + 4 * g < fvalid) {   // fvalid is a multiple of 4: whole quads
        *reinterpret_cast<f32x4*>(out + row * ldo + 16 * (2 * w + ot) + 4 * g) = o;
        cs[ot] += o;
      }
    }
    if (tn < n_tiles) store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // column sums of this workgroup: the 16 row-lanes of a feature quad are added in a fixed (butterfly) order
#pragma unroll
  for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = cs[ot][r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
      cs[ot][r] = v;
    }
    if (c == 0) *reinterpret_cast<f32x4*>(partial + (long)blockIdx.x * kHid + 16 * (2 * w + ot) + 4 * g) = cs[ot];
  }
}

// out (N, ldo)[:, :fvalid] = (G[:, :kvalid] W[:kvalid, :fvalid]) * [H > 0] (H null: no mask); colsum (null: skipped) = its
// column sums.  W[k][f] = W[k*ldw + f].  fvalid is a multiple of 4, ldo too (16-byte stores).
template <int NKB>
int dgrad(long N, const float* G, int ldg, int kvalid, const float* W, int ldw, int fvalid, unsigned* wpack, const float* H,
          float* out, int ldo, float* partial, float* colsum, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_wt_bf, dim3(16 * NKB * 2), dim3(256), 0, st, W, ldw, kvalid, fvalid, NKB, wpack);
  const long n_tiles = (N + 15) / 16;
  const int nb = (int)(n_tiles < 256 ? n_tiles : 256);
  hipLaunchKernelGGL(k_dgrad<NKB>, dim3(nb), dim3(512), 0, st, N, G, ldg, kvalid, (const unsigned*)wpack, H, out, ldo, fvalid,
                     partial);
  if (colsum) hipLaunchKernelGGL(k_colsum_final, dim3(fvalid), dim3(64), 0, st, nb, kHid, partial, colsum);
  return launch_status();
}

template <int NKB>
int dgrad(long N, const float* G, int ldg, int kvalid, const float* W, unsigned* wpack, const float* H, float* out,
          float* partial, float* colsum, hipStream_t st) {
  return dgrad<NKB>(N, G, ldg, kvalid, W, kHid, kHid, wpack, H, out, kHid, partial, colsum, st);
}

// ---- RefineNet's backward, one pass per layer (round 4): every saved activation is read ONCE ---------------------------
// Until round 3 each layer ran an activation-gradient launch (k_dgrad) AND a weight-gradient launch (k_wgrad*), each
// streaming the same (N,256) activations from HBM: h2 and h1 were read twice, dH2 written once and read twice, dH1 written and
// read twice (per-scene sums, dW1) -- 2.4 GB of the backward's ~5 GB at 786 432 rows.  Both contractions of a layer need the
// same rows at the same time, so they share one pass:
//   k_bwd_l2:  dH2 = (dO W3) * [h2 > 0]  +  db2 partials  +  dW3 = dO^T h2            (h2 read once: its mask quads ARE the operand)
//   k_bwd_l1:  dH1 = (dH2 W2) * [h1 > 0] +  db1 partials  +  per-scene sums S  +  dW1[:, 224:] = dH1^T x47
//              (dH1 never leaves the chip unless the encoders' backward follows: PSTL_FLAG_KEEP_DH1)
// dW2 = dH2^T h1 stays a launch of its own (k_wgrad_bf): its 256 x 256 accumulators (128 registers per lane) do not fit beside
// the register-stationary W2^T (128) of the activation gradient.
// Scheme: k_dgrad's (transposed weights register-stationary as split-bf16 A operands, 16-row gradient tiles staged through
// LDS as B operands, mask + column sums in the epilogue) on 32-row chunks = two tiles = ONE k-block of the weight-gradient
// MFMAs, whose contraction index is the row.  A lane of the epilogue holds four consecutive features of one row; it writes
// them (k_bwd_l2: the h2 quad it just used as mask) as bfloat16 hi / lo pieces into the operand layout of k_wgrad_bf
// ([16-column tile][row / 8][column % 16][row % 8]) -- and the tiles a wave writes are exactly the tiles its own weight-
// gradient MFMAs read (wave w owns features [32 w, 32 w + 32) in both roles), so that hand-over needs no barrier.  The small
// second operand (dO^T: 48 x 32, x47: 32 x 48) is staged by all threads from the contiguous chunk.  Arithmetic: split-bf16
// products (2^-17 per operand, fp32 range and accumulation) for both contractions; cfg->chain_waves 8 / 4 (exact request)
// keeps the unfused launches with the fp32-MFMA weight gradients.  Deterministic: slabs and column-sum partials per
// workgroup, added in a fixed order.
constexpr int kFcRows = 32;
__device__ __forceinline__ int wg_off(int col16, int row32) { return ((row32 >> 3) * 16 + col16) * 16 + (row32 & 7) * 2; }
__device__ __forceinline__ void put_pieces(char* tile_hi, char* tile_lo, int col16, int row32, float v) {
  const __bf16 hi = (__bf16)v;
  const int o = wg_off(col16, row32);
  *reinterpret_cast<__bf16*>(tile_hi + o) = hi;
  *reinterpret_cast<__bf16*>(tile_lo + o) = (__bf16)(v - (float)hi);
}

constexpr size_t bwd_l2_lds() { return (size_t)2 * 2 * 2 * 2 * 64 * 16 + (size_t)2 * 3 * 2 * kWbTile + (size_t)16 * 2 * kWbTile; }
constexpr size_t bwd_l1_lds() { return (size_t)2 * 8 * 2 * 64 * 16 + (size_t)2 * 3 * 2 * kWbTile + (size_t)16 * 2 * kWbTile; }

// dH2 (N,256), partial [grid][256] (column sums of dH2), slabs [grid][48][256] (dW3 partials; rows 40..47 are zero)
__global__ __launch_bounds__(512, 1) void k_bwd_l2(long N, const float* dO, const unsigned* Wp, const float* h2, float* dH2,
                                                   float* partial, float* slabs) {
  constexpr int NKB = 2;
  extern __shared__ __attribute__((aligned(16))) char flds[];
  u32x4* pieces = reinterpret_cast<u32x4*>(flds);                       // [buf][tile][kb][hi | lo][lane]
  char* dot = flds + (size_t)2 * 2 * NKB * 2 * 64 * 16;                 // [buf][3 column tiles of dO][hi | lo]: dO^T, A operand
  char* ht = dot + (size_t)2 * 3 * 2 * kWbTile;                         // [16 column tiles of h2][hi | lo]: B operand, wave-private
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  for (int i = tid; i < (int)(2 * 3 * 2 * kWbTile / 16); i += 512) reinterpret_cast<u32x4*>(dot)[i] = u32x4{0u, 0u, 0u, 0u};   // columns 40..47
  bf16x8 wh[2][NKB], wl[2][NKB];
  {
    const u32x4* q4 = reinterpret_cast<const u32x4*>(Wp);
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const long blk = (long)(2 * w + ot) * NKB + kb;
        wh[ot][kb] = __builtin_bit_cast(bf16x8, q4[(blk * 2 + 0) * 64 + lane]);
        wl[ot][kb] = __builtin_bit_cast(bf16x8, q4[(blk * 2 + 1) * 64 + lane]);
      }
  }
  const long n_chunks = (N + kFcRows - 1) / kFcRows;
  const long per = (n_chunks + gridDim.x - 1) / gridDim.x;
  const long c0 = blockIdx.x * per, c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;
  f32x4 rq[2];
  float rd[3];
  f32x4 hn[2][2];   // the chunk's h2 quads of this lane (mask and dW3 operand), fetched one chunk ahead like everything else
  auto load_chunk = [&](long ch) {
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
      const long row = ch * kFcRows + tile * 16 + c;
#pragma unroll
      for (int ot = 0; ot < 2; ++ot)
        hn[tile][ot] = row < N ? *reinterpret_cast<const f32x4*>(h2 + row * kHid + 16 * (2 * w + ot) + 4 * g) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    rq[0] = rq[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (w < 2 * NKB) {   // waves 0..3: tile w >> 1, k-block w & 1 of the dO tile (the activation gradient's B operand)
      const long row = ch * kFcRows + (w >> 1) * 16 + c;
      if (row < N) {
        const int k0 = 32 * (w & 1) + 4 * g;
        if (k0 < kCtrl) rq[0] = *reinterpret_cast<const f32x4*>(dO + row * kCtrl + k0);
        if (k0 + 16 < kCtrl) rq[1] = *reinterpret_cast<const f32x4*>(dO + row * kCtrl + k0 + 16);
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {   // the chunk of dO once more, element-wise (contiguous: 1 280 floats), for dO^T
      const int e = tid + 512 * u;
      const long row = ch * kFcRows + e / kCtrl;
      rd[u] = (e < kFcRows * kCtrl && row < N) ? dO[ch * kFcRows * kCtrl + e] : 0.0f;
    }
  };
  auto store_chunk = [&](int buf) {
    if (w < 2 * NKB) {
      float hi[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) hi[q] = (float)(__bf16)rq[0][q], hi[4 + q] = (float)(__bf16)rq[1][q];
      u32x4* pb = pieces + ((size_t)(buf * 2 + (w >> 1)) * NKB + (w & 1)) * 2 * 64;
      pb[lane] = u32x4{bf16_pair(hi[0], hi[1]), bf16_pair(hi[2], hi[3]), bf16_pair(hi[4], hi[5]), bf16_pair(hi[6], hi[7])};
      pb[64 + lane] = u32x4{bf16_pair(rq[0][0] - hi[0], rq[0][1] - hi[1]), bf16_pair(rq[0][2] - hi[2], rq[0][3] - hi[3]),
                            bf16_pair(rq[1][0] - hi[4], rq[1][1] - hi[5]), bf16_pair(rq[1][2] - hi[6], rq[1][3] - hi[7])};
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + 512 * u;
      if (e < kFcRows * kCtrl) {
        const int row = e / kCtrl, col = e % kCtrl;
        char* t = dot + (size_t)(buf * 3 + (col >> 4)) * 2 * kWbTile;
        put_pieces(t, t + kWbTile, col & 15, row, rd[u]);
      }
    }
  };
  f32x4 wacc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a) wacc[a][0] = wacc[a][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 cs[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
  if (c0 < c1) {
    load_chunk(c0);
    __syncthreads();   // (the zero fill of dot)
    store_chunk(0);
  }
  __syncthreads();
  int buf = 0;
  for (long ch = c0; ch < c1; ++ch) {
    f32x4 hc[2][2];
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) hc[tile][0] = hn[tile][0], hc[tile][1] = hn[tile][1];
    if (ch + 1 < c1) load_chunk(ch + 1);          // in flight during the MFMAs below
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
      const long row = ch * kFcRows + tile * 16 + c;
      const f32x4 hm[2] = {hc[tile][0], hc[tile][1]};
      f32x4 acc[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
      const u32x4* pb = pieces + (size_t)(buf * 2 + tile) * NKB * 2 * 64 + lane;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, pb[(2 * kb) * 64]), bl = __builtin_bit_cast(bf16x8, pb[(2 * kb + 1) * 64]);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ot][kb], bh, acc[ot], 0, 0, 0);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ot][kb], bh, acc[ot], 0, 0, 0);
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ot][kb], bl, acc[ot], 0, 0, 0);
      }
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = hm[ot][r] > 0.0f ? acc[ot][r] : 0.0f;
        if (row < N) {
          *reinterpret_cast<f32x4*>(dH2 + row * kHid + 16 * (2 * w + ot) + 4 * g) = o;
          cs[ot] += o;
        }
        char* t = ht + (size_t)(2 * w + ot) * 2 * kWbTile;      // this wave's own operand tiles (rows past N: zeros)
#pragma unroll
        for (int r = 0; r < 4; ++r) put_pieces(t, t + kWbTile, 4 * g + r, tile * 16 + c, hm[ot][r]);
      }
    }
    // dW3 partials: (dO^T: 3 tiles of 16 columns) x (this wave's two 16-column tiles of h2), contraction over the 32 rows
    wg_bf8 bh[2], bl[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      bh[b] = *reinterpret_cast<const wg_bf8*>(ht + (size_t)((2 * w + b) * 2 + 0) * kWbTile + lane * 16);
      bl[b] = *reinterpret_cast<const wg_bf8*>(ht + (size_t)((2 * w + b) * 2 + 1) * kWbTile + lane * 16);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const wg_bf8 ah = *reinterpret_cast<const wg_bf8*>(dot + (size_t)((buf * 3 + a) * 2 + 0) * kWbTile + lane * 16);
      const wg_bf8 al = *reinterpret_cast<const wg_bf8*>(dot + (size_t)((buf * 3 + a) * 2 + 1) * kWbTile + lane * 16);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        wacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], wacc[a][b], 0, 0, 0);
        wacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], wacc[a][b], 0, 0, 0);
        wacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b], wacc[a][b], 0, 0, 0);
      }
    }
    if (ch + 1 < c1) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = cs[ot][r];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
      cs[ot][r] = v;
    }
    if (c == 0) *reinterpret_cast<f32x4*>(partial + (long)blockIdx.x * kHid + 16 * (2 * w + ot) + 4 * g) = cs[ot];
  }
  float* out = slabs + (long)blockIdx.x * 48 * kHid;      // accumulator tile: row f = 16 a + 4 (lane >> 4) + r, column 16 (2w + b) + (lane & 15)
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * a + 4 * g + r) * kHid + 16 * (2 * w + b) + c] = wacc[a][b][r];
}

// Workgroup b owns the scenes [b * spw, (b + 1) * spw): rows_per_scene is a multiple of 32, so chunks never straddle a scene and
// N = bs * rows_per_scene has no tail.  S (bs,256) per-scene sums of dH1, partial [grid][256] its column sums per workgroup,
// slabs [grid][256][48] the partials of dH1^T x47 (column 47 is zero), dH1 (N,256) only with WRITE.
template <bool WRITE>
__global__ __launch_bounds__(512, 1) void k_bwd_l1(int bs, int rows_per_scene, int spw, const float* dH2, const unsigned* Wp,
                                                   const float* h1, const float* x47, float* dH1, float* S, float* slabs) {
  constexpr int NKB = 8;
  extern __shared__ __attribute__((aligned(16))) char flds[];
  u32x4* pieces = reinterpret_cast<u32x4*>(flds);                       // [buf][kb][hi | lo][lane]: ONE 16-row tile per buffer
  char* xt = flds + (size_t)2 * NKB * 2 * 64 * 16;                      // [buf][3 column tiles of x47][hi | lo]: B operand, per chunk
  char* at = xt + (size_t)2 * 3 * 2 * kWbTile;                          // [16 feature tiles of dH1][hi | lo]: dH1^T, wave-private
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  for (int i = tid; i < (int)(2 * 3 * 2 * kWbTile / 16); i += 512) reinterpret_cast<u32x4*>(xt)[i] = u32x4{0u, 0u, 0u, 0u};   // column 47
  bf16x8 wh[2][NKB], wl[2][NKB];
  {
    const u32x4* q4 = reinterpret_cast<const u32x4*>(Wp);
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const long blk = (long)(2 * w + ot) * NKB + kb;
        wh[ot][kb] = __builtin_bit_cast(bf16x8, q4[(blk * 2 + 0) * 64 + lane]);
        wl[ot][kb] = __builtin_bit_cast(bf16x8, q4[(blk * 2 + 1) * 64 + lane]);
      }
  }
  // The pass is pipelined by 16-row TILE (the activation gradient's unit): while tile T is in the matrix pipe, the dH2 quads AND
  // the h1 mask quads of tile T + 1 are in flight -- fetched at the top of a tile's own iteration, the mask cost one HBM round
  // trip per tile (2.5 us: the whole kernel's pace).  The weight gradient runs after every second tile (32 rows = one k-block).
  const int tps = rows_per_scene / 16;               // tiles per scene (even)
  const long s0 = (long)blockIdx.x * spw, s1 = (s0 + spw < bs) ? s0 + spw : bs;
  const long t0 = s0 * tps, t1 = s1 * tps;
  f32x4 rq[2], hn[2];
  float rx[3];
  auto load_tile = [&](long t) {   // wave w makes k-block w of the tile (k_dgrad's staging) and fetches its own mask quads
    const long row = t * 16 + c;
    const int k0 = 32 * w + 4 * g;
    rq[0] = *reinterpret_cast<const f32x4*>(dH2 + row * kHid + k0);
    rq[1] = *reinterpret_cast<const f32x4*>(dH2 + row * kHid + k0 + 16);
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) hn[ot] = *reinterpret_cast<const f32x4*>(h1 + row * kHid + 16 * (2 * w + ot) + 4 * g);
  };
  auto store_tile = [&](int buf) {
    float hi[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) hi[q] = (float)(__bf16)rq[0][q], hi[4 + q] = (float)(__bf16)rq[1][q];
    u32x4* pb = pieces + ((size_t)buf * NKB + w) * 2 * 64;
    pb[lane] = u32x4{bf16_pair(hi[0], hi[1]), bf16_pair(hi[2], hi[3]), bf16_pair(hi[4], hi[5]), bf16_pair(hi[6], hi[7])};
    pb[64 + lane] = u32x4{bf16_pair(rq[0][0] - hi[0], rq[0][1] - hi[1]), bf16_pair(rq[0][2] - hi[2], rq[0][3] - hi[3]),
                          bf16_pair(rq[1][0] - hi[4], rq[1][1] - hi[5]), bf16_pair(rq[1][2] - hi[6], rq[1][3] - hi[7])};
  };
  auto load_x = [&](long ch) {   // the chunk of x47 (contiguous: 32 x 47 floats)
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + 512 * u;
      rx[u] = e < kFcRows * kX47 ? x47[ch * kFcRows * kX47 + e] : 0.0f;
    }
  };
  auto store_x = [&](int xbuf) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = tid + 512 * u;
      if (e < kFcRows * kX47) {
        const int row = e / kX47, col = e % kX47;
        char* t = xt + (size_t)(xbuf * 3 + (col >> 4)) * 2 * kWbTile;
        put_pieces(t, t + kWbTile, col & 15, row, rx[u]);
      }
    }
  };
  f32x4 wacc[2][3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) wacc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 css[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
  if (t0 < t1) {
    load_tile(t0);
    load_x(t0 / 2);
    __syncthreads();   // (the zero fill of xt)
    store_tile(0);
    store_x(0);
  }
  __syncthreads();
  int buf = 0, xbuf = 0, in_scene = 0;
  long scene = s0;
  for (long t = t0; t < t1; ++t) {
    const int half = (int)(t & 1);            // which half of the 32-row chunk (t0 is even)
    const f32x4 hm[2] = {hn[0], hn[1]};
    if (t + 1 < t1) load_tile(t + 1);         // dH2 and mask quads of the next tile: in flight during the MFMAs below
    if (half == 0 && t + 2 < t1) load_x(t / 2 + 1);
    const long row = t * 16 + c;
    f32x4 acc[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
    const u32x4* pb = pieces + (size_t)buf * NKB * 2 * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, pb[(2 * kb) * 64]), bl = __builtin_bit_cast(bf16x8, pb[(2 * kb + 1) * 64]);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ot][kb], bh, acc[ot], 0, 0, 0);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ot][kb], bh, acc[ot], 0, 0, 0);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ot][kb], bl, acc[ot], 0, 0, 0);
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = hm[ot][r] > 0.0f ? acc[ot][r] : 0.0f;
      if (WRITE) *reinterpret_cast<f32x4*>(dH1 + row * kHid + 16 * (2 * w + ot) + 4 * g) = o;
      css[ot] += o;
      char* tl = at + (size_t)(2 * w + ot) * 2 * kWbTile;      // this wave's own operand tiles
#pragma unroll
      for (int r = 0; r < 4; ++r) put_pieces(tl, tl + kWbTile, 4 * g + r, half * 16 + c, o[r]);
    }
    if (half == 1) {
      // dW1[:, 224:271] partials: (this wave's two 16-feature tiles of dH1^T) x (3 tiles of 16 columns of x47), over the 32 rows
      // (operands fetched pair by pair: the register file is full of W2^T)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const wg_bf8 ah = *reinterpret_cast<const wg_bf8*>(at + (size_t)((2 * w + a) * 2 + 0) * kWbTile + lane * 16);
        const wg_bf8 al = *reinterpret_cast<const wg_bf8*>(at + (size_t)((2 * w + a) * 2 + 1) * kWbTile + lane * 16);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const wg_bf8 bh = *reinterpret_cast<const wg_bf8*>(xt + (size_t)((xbuf * 3 + b) * 2 + 0) * kWbTile + lane * 16);
          const wg_bf8 bl = *reinterpret_cast<const wg_bf8*>(xt + (size_t)((xbuf * 3 + b) * 2 + 1) * kWbTile + lane * 16);
          wacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, wacc[a][b], 0, 0, 0);
          wacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, wacc[a][b], 0, 0, 0);
          wacc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, wacc[a][b], 0, 0, 0);
        }
      }
      if (t + 1 < t1) {   // the next chunk's x47 (loaded a tile ago); the other buffer's last readers are two barriers back
        store_x(xbuf ^ 1);
        xbuf ^= 1;
      }
    }
    if (++in_scene == tps) {   // the scene is complete: its sums (16 row-lanes added in a fixed butterfly order)
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        f32x4 v = css[ot];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) v[r] += __shfl_xor(v[r], m, 64);
        if (c == 0) *reinterpret_cast<f32x4*>(S + scene * kHid + 16 * (2 * w + ot) + 4 * g) = v;
        css[ot] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
      in_scene = 0;
      ++scene;
    }
    if (t + 1 < t1) store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  float* out = slabs + (long)blockIdx.x * kHid * 48;      // accumulator tile: row f = 16 (2w + a) + 4 (lane >> 4) + r, column 16 b + (lane & 15)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * (2 * w + a) + 4 * g + r) * 48 + 16 * b + c] = wacc[a][b][r];
}

// ---- scene-encoder backward (training with --joint) -----------------------------------------------------------------
// feature (bs,224) = [ego 32 | neighbour min 32 | mean 32 | max 32 | 3 lanes x 32] (nusc_model.py:82-93): route d feature
// to the encoder outputs, tokens ordered [bs ego | bs*K neighbours | 3*bs lanes].  min / max send their gradient to the
// first neighbour holding the extreme (torch.min/max(dim)); the mean to all K.
__global__ void k_pool_bwd(int bs, int K, const float* dfeat, const float* tok_out, float* dtok) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)bs * 32) return;
  const long b = i >> 5;
  const int o = (int)(i & 31);
  const float* df = dfeat + b * kFeat;
  dtok[b * 32 + o] = df[o];
  const float* no = tok_out + ((long)bs + b * K) * 32 + o;
  int imin = 0, imax = 0;
  float vmin = no[0], vmax = no[0];
  for (int k = 1; k < K; ++k) {
    const float v = no[(long)k * 32];
    if (v < vmin) vmin = v, imin = k;
    if (v > vmax) vmax = v, imax = k;
  }
  const float gmin = df[32 + o], gmean = df[64 + o] / (float)K, gmax = df[96 + o];
  for (int k = 0; k < K; ++k)
    dtok[((long)bs + b * K + k) * 32 + o] = gmean + (k == imin ? gmin : 0.0f) + (k == imax ? gmax : 0.0f);
  for (int m = 0; m < 3; ++m) dtok[((long)bs * (K + 1) + b * 3 + m) * 32 + o] = df[128 + 32 * m + o];
}

// partial[block][c] = sum over the block's rows of G[row][c], c < ncol <= 64 (one column per thread, 4 row-lanes)
__global__ __launch_bounds__(256) void k_colsum_narrow(long N, const float* G, int ld, int ncol, float* partial) {
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long per = (N + gridDim.x - 1) / gridDim.x;
  const long r0 = blockIdx.x * per, r1 = (r0 + per < N) ? r0 + per : N;
  float acc = 0.0f;
  if (c < ncol)
    for (long r = r0 + rl; r < r1; r += 4) acc += G[r * ld + c];
  __shared__ float red[4][64];
  red[rl][c] = acc;
  __syncthreads();
  if (rl == 0 && c < ncol) partial[(long)blockIdx.x * ncol + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// the work buffer of pstl_refine_backward (pstl_train_work_floats); pstl_encoder_backward reads dH1 and S from it
struct RefineWork {
  float *dO, *x47, *dH2, *dH1, *S, *part, *slabs;
  unsigned* wpack;
  RefineWork(const pstl_cfg* cfg, float* work) {
    const long N = n_rows(cfg);
    dO = work;                                // (N,40)
    x47 = dO + N * kCtrl;                     // (N,47)
    dH2 = x47 + N * kX47;                     // (N,256)
    dH1 = dH2 + N * kHid;                     // (N,256)
    S = dH1 + N * kHid;                       // (bs,256)
    part = S + (long)cfg->bs * kHid;          // (kRedBlocks,256)
    slabs = part + (long)kRedBlocks * kHid;   // (kRedBlocks,256,256) split-K partials of k_wgrad
    wpack = reinterpret_cast<unsigned*>(slabs + (long)kRedBlocks * kHid * kHid);   // kWtPackWords
  }
};

}  // namespace
}  // namespace pstl

using namespace pstl;

extern "C" size_t pstl_train_work_floats(const pstl_cfg* cfg) {
  if (check_cfg(cfg)) return 0;
  const long N = n_rows(cfg);
  return (size_t)(N * (kCtrl + kX47 + 2L * kHid) + (long)cfg->bs * kHid + (long)kRedBlocks * kHid +
                  (long)kRedBlocks * kHid * kHid + kWtPackWords + 64);
}

extern "C" int pstl_loss_grad(const pstl_cfg* cfg, const float* scores, const float* valid, float grad_scale, float* dscore,
                              float* loss_parts /* 256 floats */, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!scores || !valid || !dscore || !loss_parts) return PSTL_ERR_ARG;
  const float thres = (cfg->flags & PSTL_FLAG_MAXIMIZE) ? 100.0f : cfg->thres;
  hipLaunchKernelGGL(k_loss_grad, dim3(256), dim3(256), 0, as_stream(stream), n_rows(cfg), scores, valid, thres, grad_scale,
                     dscore, loss_parts);
  return launch_status();
}

extern "C" int pstl_refine_backward(const pstl_cfg* cfg, const float* w2 /* (256,256) */,
                                    const float* w3 /* (40,256) */, const float* feature /* (bs,224) */, const float* stlp,
                                    const float* hl, const float* init_controls, const float* pooled,
                                    const float* prev_scores, const float* h1, const float* h2, const float* pre,
                                    const float* dcontrols, float* work, float* dw1 /* (256,271) */, float* db1,
                                    float* dw2, float* db2, float* dw3 /* (40,256) */, float* db3, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!w2 || !w3 || !feature || !stlp || !hl || !init_controls || !prev_scores || !h1 || !h2 || !pre ||
      !dcontrols || !work || !dw1 || !db1 || !dw2 || !db2 || !dw3 || !db3)
    return PSTL_ERR_ARG;
  if (cfg->flags & PSTL_FLAG_CLIP_RECT) return PSTL_ERR_SHAPE;  // the reference trains without --clip_rect
  const bool merge = !(cfg->flags & PSTL_FLAG_NO_MERGE);
  if (merge && (!pooled || cfg->rows_per_scene != 3 * cfg->S || cfg->n_shards < 1 || cfg->S % cfg->n_shards != 0))
    return PSTL_ERR_ARG;
  hipStream_t st = as_stream(stream);
  const long N = n_rows(cfg);
  const RefineWork rw(cfg, work);
  float *dO = rw.dO, *x47 = rw.x47, *dH2 = rw.dH2, *dH1 = rw.dH1, *S = rw.S, *part = rw.part, *slabs = rw.slabs;
  unsigned* wpack = rw.wpack;
  const int nb = (int)(N < kRedBlocks ? N : kRedBlocks);

  hipLaunchKernelGGL(k_head_bwd, dim3((unsigned)((N * kCtrl + 255) / 256)), dim3(256), 0, st, N, cfg->w_max, cfg->a_max,
                     dcontrols, pre, init_controls, prev_scores, hl, stlp, merge ? pooled : (const float*)nullptr, cfg->S,
                     cfg->n_shards, dO, x47);
  // layer 3
  hipLaunchKernelGGL(k_colsum40, dim3(nb), dim3(256), 0, st, N, dO, part);
  hipLaunchKernelGGL(k_colsum_final, dim3(kCtrl), dim3(64), 0, st, nb, kCtrl, part, db3);
  const bool exact = cfg->chain_waves == 8 || cfg->chain_waves == 4;
  if (!exact && cfg->rows_per_scene % kFcRows == 0) {
    // one pass per layer (k_bwd_l2 / k_bwd_l1 above): each saved activation is read once
    // (per device: a process may drive several GPUs -- the CU count and the function attributes belong to the current one)
    static DeviceOnce lds_allowed;
    const int dev = current_device();
    if (dev < 0) return PSTL_ERR_LAUNCH;
    int cus = device_cus(dev);
    if (cus > kRedBlocks) cus = kRedBlocks;
    const long n_chunks = (N + kFcRows - 1) / kFcRows;
    const int nb2 = (int)(n_chunks < cus ? n_chunks : cus);
    if (!lds_allowed.done(dev)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_l2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_l2_lds()) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_l1<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_l1_lds()) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(k_bwd_l1<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_l1_lds()) != hipSuccess)
        return PSTL_ERR_LAUNCH;
      lds_allowed.set(dev);
    }
    // layer 2 and dW3:  dH2 = (dO W3) * [h2 > 0], db2, dW3 = dO^T h2
    hipLaunchKernelGGL(k_pack_wt_bf, dim3(16 * 2 * 2), dim3(256), 0, st, w3, kHid, kCtrl, kHid, 2, wpack);
    hipLaunchKernelGGL(k_bwd_l2, dim3(nb2), dim3(512), bwd_l2_lds(), st, N, (const float*)dO, (const unsigned*)wpack, h2, dH2, part, slabs);
    hipLaunchKernelGGL(k_colsum_final, dim3(kHid), dim3(64), 0, st, nb2, kHid, part, db2);
    hipLaunchKernelGGL(k_slab_reduce, dim3((48 * kHid + 255) / 256), dim3(256), 0, st, nb2, 48, kHid, kCtrl, kHid, slabs, dw3, kHid);
    if (int e = wgrad_bf<16, 16, 2, 4>(N, dH2, kHid, kHid, h1, kHid, kHid, slabs, dw2, kHid, st)) return e;  // dW2 = dH2^T h1
    // layer 1:  dH1 = (dH2 W2) * [h1 > 0], db1, per-scene sums S, dW1[:, 224:] = dH1^T x47
    const int spw = (cfg->bs + cus - 1) / cus;
    const int nb1 = (cfg->bs + spw - 1) / spw;
    hipLaunchKernelGGL(k_pack_wt_bf, dim3(16 * 8 * 2), dim3(256), 0, st, w2, kHid, kHid, kHid, 8, wpack);
    if (cfg->flags & PSTL_FLAG_KEEP_DH1)
      hipLaunchKernelGGL(k_bwd_l1<true>, dim3(nb1), dim3(512), bwd_l1_lds(), st, cfg->bs, cfg->rows_per_scene, spw, (const float*)dH2,
                         (const unsigned*)wpack, h1, (const float*)x47, dH1, S, slabs);
    else
      hipLaunchKernelGGL(k_bwd_l1<false>, dim3(nb1), dim3(512), bwd_l1_lds(), st, cfg->bs, cfg->rows_per_scene, spw, (const float*)dH2,
                         (const unsigned*)wpack, h1, (const float*)x47, dH1, S, slabs);
    hipLaunchKernelGGL(k_colsum_final, dim3(kHid), dim3(64), 0, st, cfg->bs, kHid, (const float*)S, db1);   // db1 = the sum of the scenes' sums
    hipLaunchKernelGGL(k_slab_reduce, dim3((kHid * 48 + 255) / 256), dim3(256), 0, st, nb1, kHid, 48, kHid, kX47, slabs, dw1 + kFeat, kIn);
    if (int e = launch_status()) return e;
  } else {
  if (int e = wgrad<3, 16, 1, 8>(N, dO, kCtrl, kCtrl, h2, kHid, kHid, slabs, dw3, kHid, st)) return e;  // dW3 = dO^T h2
  // layer 2: dH2 = (dO W3) * [h2 > 0], db2 = column sums of dH2
  if (int e = dgrad<2>(N, dO, kCtrl, kCtrl, w3, wpack, h2, dH2, part, db2, st)) return e;
  // dW2 = dH2^T h1: the exact-fp32 request (cfg->chain_waves 8 / 4, also what the host falls back to after a split-f16 domain
  // overflow) keeps every weight gradient on the fp32 MFMA form
  if (exact) {
    if (int e = wgrad<16, 16, 4, 2>(N, dH2, kHid, kHid, h1, kHid, kHid, slabs, dw2, kHid, st)) return e;
  } else {
    if (int e = wgrad_bf<16, 16, 2, 4>(N, dH2, kHid, kHid, h1, kHid, kHid, slabs, dw2, kHid, st)) return e;
  }
  // layer 1: dH1 = (dH2 W2) * [h1 > 0], db1 = column sums of dH1
  if (int e = dgrad<8>(N, dH2, kHid, kHid, w2, wpack, h1, dH1, part, db1, st)) return e;
  hipLaunchKernelGGL(k_scene_sum, dim3(cfg->bs), dim3(256), 0, st, cfg->rows_per_scene, dH1, S);
  if (int e = wgrad<16, 3, 8, 1>(N, dH1, kHid, kHid, x47, kX47, kX47, slabs, dw1 + kFeat, kIn, st)) return e;
  }
  // the 224 scene-constant input columns: dW1[:, :224] = S^T feature (contraction over the scenes)
  if (int e = wgrad<16, 14, 4, 2>(cfg->bs, S, kHid, kHid, feature, kFeat, kFeat, slabs, dw1, kIn, st)) return e;
  return launch_status();
}


extern "C" int pstl_diversity_loss(const pstl_cfg* cfg, const float* rect_controls, const float* init_controls,
                                   const float* scores, float diversity_scale, float diversity_weight, int detach,
                                   float rect_reg_weight, float* group_div, float* reg_out, double* reg_work,
                                   float* dcontrols, float* dscore, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!rect_controls || !scores || !group_div || !dcontrols || !dscore) return PSTL_ERR_ARG;
  if (cfg->rows_per_scene != 3 * cfg->S || cfg->n_shards < 1 || cfg->S % cfg->n_shards != 0) return PSTL_ERR_SHAPE;
  const int n = cfg->S / cfg->n_shards;
  if (n > 64) return PSTL_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  const long N = n_rows(cfg);
  const int groups = cfg->bs * 3 * cfg->n_shards;
  DppArgs a;
  a.bs = cfg->bs;
  a.S = cfg->S;
  a.n_shards = cfg->n_shards;
  a.w_max = cfg->w_max;
  a.a_max = cfg->a_max;
  a.scale = diversity_scale;
  a.c = diversity_weight / (float)groups;
  a.detach = detach;
  a.rect = rect_controls;
  a.scores = scores;
  a.group_div = group_div;
  a.dcontrols = dcontrols;
  a.dscore = dscore;
  const int gpw = kDppWave / n;                     // groups per wavefront (n <= 64 is checked above)
  const size_t lds = dpp_group_bytes(n) * gpw;
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k_dpp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess)
    return PSTL_ERR_LAUNCH;
  hipLaunchKernelGGL(k_dpp, dim3((unsigned)((groups + gpw - 1) / gpw)), dim3(64), lds, st, a, groups);
  if (rect_reg_weight != 0.0f || reg_out) {
    if (!init_controls || !reg_out || !reg_work) return PSTL_ERR_ARG;
    const int nb = 256;
    hipLaunchKernelGGL(k_reg_partials, dim3(nb), dim3(256), 0, st, N, rect_controls, init_controls, scores, reg_work);
    hipLaunchKernelGGL(k_reg_final, dim3(1), dim3(1), 0, st, nb, N, reg_work, reg_out);
    if (rect_reg_weight != 0.0f)
      hipLaunchKernelGGL(k_reg_grad, dim3((unsigned)((N * kCtrl + 255) / 256)), dim3(256), 0, st, N, rect_reg_weight,
                         rect_controls, init_controls, scores, reg_out, dcontrols);
  }
  return launch_status();
}


// ---- --joint: gradients of the three scene encoders (and d fused for merge_net) -------------------------------------
extern "C" size_t pstl_encoder_backward_work_floats(const pstl_cfg* cfg) {
  if (check_cfg(cfg)) return 0;
  const long bs = cfg->bs, K = cfg->K;
  const long tmax = bs * (K > 3 ? K : 3);
  return (size_t)(bs * kFeat + bs * (K + 4) * 32 + 2 * tmax * kHid + 64);
}

extern "C" int pstl_encoder_backward(const pstl_cfg* cfg, float* refine_work, const float* rect_w1, const float* const* enc_w1,
                                     const float* const* enc_w2, const float* tok_in, const float* tok_h1, const float* tok_h2,
                                     const float* tok_out, float* work, float* const* d_w0, float* const* d_b0,
                                     float* const* d_w1, float* const* d_b1, float* const* d_w2, float* const* d_b2,
                                     float* dfused, void* stream) {
  if (int e = check_cfg(cfg)) return e;
  if (!refine_work || !rect_w1 || !enc_w1 || !enc_w2 || !tok_in || !tok_h1 || !tok_h2 || !tok_out || !work || !d_w0 ||
      !d_b0 || !d_w1 || !d_b1 || !d_w2 || !d_b2)
    return PSTL_ERR_ARG;
  for (int e = 0; e < 3; ++e)
    if (!enc_w1[e] || !enc_w2[e] || !d_w0[e] || !d_b0[e] || !d_w1[e] || !d_b1[e] || !d_w2[e] || !d_b2[e]) return PSTL_ERR_ARG;
  hipStream_t st = as_stream(stream);
  const long N = n_rows(cfg), bs = cfg->bs, K = cfg->K;
  const RefineWork rw(cfg, refine_work);
  const long tmax = bs * (K > 3 ? K : 3);
  float* dfeat = work;                  // (bs,224)
  float* dtok = dfeat + bs * kFeat;     // (T,32)
  float* dh2 = dtok + bs * (K + 4) * 32;  // (tmax,256)
  float* dh1 = dh2 + tmax * kHid;       // (tmax,256)
  // d fused = dH1 W1[:, 231:271] (the rect_net input columns merge_net feeds); no mask: a linear input.  dH1 is in the work
  // buffer only when pstl_refine_backward was asked to leave it there: without the flag this would read stale memory.
  if (dfused && !(cfg->flags & PSTL_FLAG_KEEP_DH1)) return PSTL_ERR_ARG;
  if (dfused)
    if (int e = dgrad<8>(N, rw.dH1, kHid, kHid, rect_w1 + kFeat + 7, kIn, kCtrl, rw.wpack, nullptr, dfused, kCtrl, rw.part,
                         nullptr, st))
      return e;
  // d feature = S W1[:, :224], S = per-scene sums of dH1 (left in the work buffer by pstl_refine_backward)
  if (int e = dgrad<8>(bs, rw.S, kHid, kHid, rect_w1, kIn, kFeat, rw.wpack, nullptr, dfeat, kFeat, rw.part, nullptr, st)) return e;
  hipLaunchKernelGGL(k_pool_bwd, dim3((unsigned)((bs * 32 + 255) / 256)), dim3(256), 0, st, (int)bs, (int)K, dfeat, tok_out, dtok);
  const long t0[3] = {0, bs, bs * (K + 1)}, tn[3] = {bs, bs * K, 3 * bs};
  const int nin[3] = {6, 7, 45};
  for (int e = 0; e < 3; ++e) {
    const long T = tn[e];
    const float* g_out = dtok + t0[e] * 32;
    const float* h1 = tok_h1 + t0[e] * kHid;
    const float* h2 = tok_h2 + t0[e] * kHid;
    const float* xin = tok_in + t0[e] * 48;
    // layer 2 (256 -> 32): db2, dW2 = d out^T h2, dh2 = (d out W2) * [h2 > 0] with db1 = its column sums
    const int nb = (int)(T < 256 ? T : 256);
    hipLaunchKernelGGL(k_colsum_narrow, dim3(nb), dim3(256), 0, st, T, g_out, 32, 32, rw.part);
    hipLaunchKernelGGL(k_colsum_final, dim3(32), dim3(64), 0, st, nb, 32, rw.part, d_b2[e]);
    if (int er = wgrad<2, 16, 1, 8>(T, g_out, 32, 32, h2, kHid, kHid, rw.slabs, d_w2[e], kHid, st)) return er;
    if (int er = dgrad<1>(T, g_out, 32, 32, enc_w2[e], rw.wpack, h2, dh2, rw.part, d_b1[e], st)) return er;
    // layer 1 (256 -> 256)
    if (int er = wgrad<16, 16, 4, 2>(T, dh2, kHid, kHid, h1, kHid, kHid, rw.slabs, d_w1[e], kHid, st)) return er;
    if (int er = dgrad<8>(T, dh2, kHid, kHid, enc_w1[e], rw.wpack, h1, dh1, rw.part, d_b0[e], st)) return er;
    // layer 0 (nin -> 256): dW0 = dh1^T token inputs
    if (int er = wgrad<16, 3, 8, 1>(T, dh1, kHid, kHid, xin, 48, nin[e], rw.slabs, d_w0[e], nin[e], st)) return er;
  }
  return launch_status();
}
