// stl_program.hip -- generic differentiable STL robustness on gfx950: the operator set of the reference's stl_d_lib.py
// (softmax/softmin :6-26, And :87, ListAnd :97, Or :113, Not :125, Imply :132, Eventually :144, Always :157, Once :171,
// UntimedUntil :183, Until :195) evaluated for a whole formula tree by ONE kernel launch, and its adjoint by a second.
// Imply, Once and the two Until forms are lowered by the host onto the seven node types below (UntimedUntil(l, r) =
// Eventually[0,T)( And( r, Always[-T,1)(l) ) ) with the two cumulative operators marked "always soft", as the
// reference computes them with logcumsumexp regardless of the "hard" switch).
//
// The host flattens a formula into a postfix node list (children before parents).  One signal row (one trajectory) per
// lane; the value of node k at time t for row r lives at vals[(k*T + t)*n + r], so the 64 lanes of a wavefront always
// touch 64 consecutive floats (fully coalesced) whatever the tree looks like.  The node list is read with uniform
// (scalar) loads.  The three fixed formulas of the sampling path have their own fused kernel (stl_kernels.hip); this
// interpreter is the drop-in for everything else a user of stl_d_lib can write.
//
// Numerics follow the reference: soft max = logsumexp(tau*x)/tau with the maximum subtracted (torch.logsumexp), soft
// min = -softmax(-x); an empty window gives -inf for both (stl_d_lib.py:7-8,16-17); "hard" replaces logsumexp by max.
#include "pstl_common.hpp"

namespace pstl {
namespace {

struct ProgArgs {
  const pstl_stl_node* nodes;
  const int32_t* lists;
  int n_nodes, T, hard;
  long n;
  float tau;
  const float* signals;  // (n_sig, n, T)
  float* vals;           // (n_nodes, T, n)
  float* out;            // (n, T) value of the last node
  const float* dout;     // (n, T)
  float* adj;            // (n_nodes, T, n)
  float* dsignals;       // (n_sig, n, T), zero-initialised by the caller
};

__device__ __forceinline__ int clipi(int x, int a, int b) { return x < a ? a : (x > b ? b : x); }

// soft max over v[t0..t1) of node `src` (sign = -1: soft min), row-strided access
struct Acc {
  float m, s;
  __device__ __forceinline__ void init() { m = -INFINITY; s = 0.0f; }
  __device__ __forceinline__ void scan(float z) { m = fmaxf(m, z); }
  __device__ __forceinline__ void add(float z) { s += (z == m) ? 1.0f : expf(z - m); }   // exp(0) == 1 also for m = +-inf
  __device__ __forceinline__ float lse() const { return m + logf(s); }
};

// value of the soft max of sign*x over the given (node, t) list; tau-scaled.  Two passes (max, then sum), as
// torch.logsumexp does.  `get(i)` returns the i-th operand.
template <class Get>
__device__ __forceinline__ float soft_extreme(int cnt, float sign, float tau, bool hard, Get get) {
  if (cnt <= 0) return -INFINITY;                       // the reference returns -inf for an empty window, max or min
  Acc a;
  a.init();
  if (hard) {                                           // torch.max of the operands themselves (no tau scaling)
    for (int i = 0; i < cnt; ++i) a.scan(sign * get(i));
    return sign * a.m;
  }
  for (int i = 0; i < cnt; ++i) a.scan(sign * get(i) * tau);
  for (int i = 0; i < cnt; ++i) a.add(sign * get(i) * tau);
  return sign * (a.lse() / tau);
}

// d result / d operand i  (result = sign * LSE(sign*tau*x)/tau  =>  softmax weight, the sign cancels)
template <class Get, class Put>
__device__ __forceinline__ void soft_extreme_grad(int cnt, float sign, float tau, bool hard, float g, Get get, Put put) {
  if (cnt <= 0 || g == 0.0f) return;
  Acc a;
  a.init();
  if (hard) {                                            // torch.max: the gradient goes to the first maximal element
    for (int i = 0; i < cnt; ++i) a.scan(sign * get(i));
    for (int i = 0; i < cnt; ++i)
      if (sign * get(i) == a.m) {
        put(i, g);
        return;
      }
    return;
  }
  for (int i = 0; i < cnt; ++i) a.scan(sign * get(i) * tau);
  for (int i = 0; i < cnt; ++i) a.add(sign * get(i) * tau);
  const float inv = 1.0f / a.s;
  for (int i = 0; i < cnt; ++i) {
    const float z = sign * get(i) * tau;
    const float w = (z == a.m ? 1.0f : expf(z - a.m)) * inv;
    if (w != 0.0f) put(i, g * w);
  }
}

#define VAL(k, t) vals[((long)(k) * T + (t)) * n + row]
#define ADJ(k, t) adj[((long)(k) * T + (t)) * n + row]

__global__ __launch_bounds__(256) void k_stl_program_forward(ProgArgs p) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= p.n) return;
  const int T = p.T;
  const long n = p.n;
  const float tau = p.tau;
  const bool hard = p.hard != 0;
  float* vals = p.vals;
  for (int k = 0; k < p.n_nodes; ++k) {
    const pstl_stl_node nd = p.nodes[k];
    switch (nd.op) {
      case PSTL_STL_SIGNAL: {
        const float* src = p.signals + ((long)nd.a * n + row) * T;
        for (int t = 0; t < T; ++t) VAL(k, t) = src[t];
      } break;
      case PSTL_STL_NOT:
        for (int t = 0; t < T; ++t) VAL(k, t) = -VAL(nd.a, t);
        break;
      case PSTL_STL_AND:
      case PSTL_STL_OR: {
        const float sign = nd.op == PSTL_STL_AND ? -1.0f : 1.0f;
        for (int t = 0; t < T; ++t)
          VAL(k, t) = soft_extreme(2, sign, tau, hard, [&](int i) { return VAL(i == 0 ? nd.a : nd.b, t); });
      } break;
      case PSTL_STL_LISTAND: {
        const int32_t* ch = p.lists + nd.list_off;
        for (int t = 0; t < T; ++t)
          VAL(k, t) = soft_extreme(nd.n_list, -1.0f, tau, hard, [&](int i) { return VAL(ch[i], t); });
      } break;
      case PSTL_STL_ALWAYS:
      case PSTL_STL_EVENTUALLY: {   // Once is Eventually with a window that reaches into the past
        const float sign = nd.op == PSTL_STL_ALWAYS ? -1.0f : 1.0f;
        const bool h = hard && !(nd.flags & PSTL_STL_FLAG_SOFT);
        for (int t = 0; t < T; ++t) {
          const int t0 = clipi(t + nd.ts, 0, T), t1 = clipi(t + nd.te, 0, T);
          VAL(k, t) = soft_extreme(t1 - t0, sign, tau, h, [&](int i) { return VAL(nd.a, t0 + i); });
        }
      } break;
      default:
        break;
    }
  }
  float* o = p.out + row * T;
  for (int t = 0; t < T; ++t) o[t] = VAL(p.n_nodes - 1, t);
}

__global__ __launch_bounds__(256) void k_stl_program_backward(ProgArgs p) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= p.n) return;
  const int T = p.T;
  const long n = p.n;
  const float tau = p.tau;
  const bool hard = p.hard != 0;
  const float* vals = p.vals;
  float* adj = p.adj;
  for (int k = 0; k < p.n_nodes - 1; ++k)
    for (int t = 0; t < T; ++t) ADJ(k, t) = 0.0f;
  for (int t = 0; t < T; ++t) ADJ(p.n_nodes - 1, t) = p.dout[row * T + t];
  for (int k = p.n_nodes - 1; k >= 0; --k) {
    const pstl_stl_node nd = p.nodes[k];
    switch (nd.op) {
      case PSTL_STL_SIGNAL: {
        float* dst = p.dsignals + ((long)nd.a * n + row) * T;
        for (int t = 0; t < T; ++t) dst[t] += ADJ(k, t);   // the same signal may feed several leaves
      } break;
      case PSTL_STL_NOT:
        for (int t = 0; t < T; ++t) ADJ(nd.a, t) -= ADJ(k, t);
        break;
      case PSTL_STL_AND:
      case PSTL_STL_OR: {
        const float sign = nd.op == PSTL_STL_AND ? -1.0f : 1.0f;
        for (int t = 0; t < T; ++t)
          soft_extreme_grad(2, sign, tau, hard, ADJ(k, t), [&](int i) { return VAL(i == 0 ? nd.a : nd.b, t); },
                            [&](int i, float g) { ADJ(i == 0 ? nd.a : nd.b, t) += g; });
      } break;
      case PSTL_STL_LISTAND: {
        const int32_t* ch = p.lists + nd.list_off;
        for (int t = 0; t < T; ++t)
          soft_extreme_grad(nd.n_list, -1.0f, tau, hard, ADJ(k, t), [&](int i) { return VAL(ch[i], t); },
                            [&](int i, float g) { ADJ(ch[i], t) += g; });
      } break;
      case PSTL_STL_ALWAYS:
      case PSTL_STL_EVENTUALLY: {
        const float sign = nd.op == PSTL_STL_ALWAYS ? -1.0f : 1.0f;
        const bool h = hard && !(nd.flags & PSTL_STL_FLAG_SOFT);
        for (int t = 0; t < T; ++t) {
          const int t0 = clipi(t + nd.ts, 0, T), t1 = clipi(t + nd.te, 0, T);
          soft_extreme_grad(t1 - t0, sign, tau, h, ADJ(k, t), [&](int i) { return VAL(nd.a, t0 + i); },
                            [&](int i, float g) { ADJ(nd.a, t0 + i) += g; });
        }
      } break;
      default:
        break;
    }
  }
}

#undef VAL
#undef ADJ

static int check_program(const pstl_stl_node* nodes, int n_nodes, long n, int T) {
  if (!nodes || n_nodes < 1 || n < 1 || T < 1 || T > PSTL_STL_MAX_T) return PSTL_ERR_ARG;
  return PSTL_OK;
}

}  // namespace
}  // namespace pstl

using namespace pstl;

extern "C" int pstl_stl_program_forward(const pstl_stl_node* nodes, int n_nodes, const int32_t* lists, int64_t n, int T,
                                        const float* signals, float tau, int hard, float* vals, float* out,
                                        void* stream) {
  if (int e = check_program(nodes, n_nodes, n, T)) return e;
  if (!signals || !vals || !out) return PSTL_ERR_ARG;
  ProgArgs p{};
  p.nodes = nodes;
  p.lists = lists;
  p.n_nodes = n_nodes;
  p.T = T;
  p.hard = hard;
  p.n = n;
  p.tau = tau;
  p.signals = signals;
  p.vals = vals;
  p.out = out;
  hipLaunchKernelGGL(k_stl_program_forward, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), p);
  return launch_status();
}

extern "C" int pstl_stl_program_backward(const pstl_stl_node* nodes, int n_nodes, const int32_t* lists, int64_t n, int T,
                                         const float* vals, float tau, int hard, const float* dout, float* adj,
                                         float* dsignals, void* stream) {
  if (int e = check_program(nodes, n_nodes, n, T)) return e;
  if (!vals || !dout || !adj || !dsignals) return PSTL_ERR_ARG;
  ProgArgs p{};
  p.nodes = nodes;
  p.lists = lists;
  p.n_nodes = n_nodes;
  p.T = T;
  p.hard = hard;
  p.n = n;
  p.tau = tau;
  p.vals = const_cast<float*>(vals);
  p.dout = dout;
  p.adj = adj;
  p.dsignals = dsignals;
  hipLaunchKernelGGL(k_stl_program_backward, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), p);
  return launch_status();
}
