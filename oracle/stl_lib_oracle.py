"""CPU restatement of the reference's STL operator semantics (stl_d_lib.py), TEST INFRASTRUCTURE only.

Evaluates the nested-list formula specs of tests/stl_specs.py directly with torch-CPU float32 ops (autograd gives the
gradients).  Semantics restated (reference stl_d_lib.py):
  soft max = logsumexp(tau x)/tau, soft min = -softmax(-x); empty window -> -inf for both (:6-20); hard -> torch.max;
  And/Or = pairwise soft min/max (:87,:113); ListAnd = soft min over the children, per time step (:97-111);
  Always/Eventually/Once: out[t] = soft min/max of s[clip(t+ts,0,T) : clip(t+te,0,T)] (:144-181);
  UntimedUntil: suffix-softmax( softmin2( rhs, prefix-softmin(lhs) ) ) with logcumsumexp, never "hard" (:183-193);
  Until(ts>0) = And(Eventually(ts,te,rhs), Always(0,ts,UntimedUntil)) (:195-204); Imply = Or(Not lhs, rhs) (:132).
Pinned by tests/golden/stl_lib.npz (values and autograd gradients produced by the reference's own classes).
"""
import torch


def _smax(x, tau, hard, dim):
    if x.shape[dim] == 0:
        shape = list(x.shape)
        shape[dim] = 1
        return torch.full(shape, float("-inf"))
    if hard:
        return torch.max(x, dim=dim, keepdim=True)[0]
    return torch.logsumexp(x * tau, dim=dim, keepdim=True) / tau


def _smin(x, tau, hard, dim):
    if x.shape[dim] == 0:
        shape = list(x.shape)
        shape[dim] = 1
        return torch.full(shape, float("-inf"))
    return -_smax(-x, tau, hard, dim)


def _window(s, ts, te, tau, hard, fn):
    T = s.shape[1]
    cl = lambda v: max(min(v, T), 0)
    return torch.cat([fn(s[:, cl(t + ts):cl(t + te)], tau, hard, 1) for t in range(T)], dim=1)


def _untimed_until(ls, rs, tau, hard):
    inf_ls = -torch.logcumsumexp(-ls * tau, dim=1) / tau
    mid = _smin(torch.stack([rs, inf_ls], dim=1), tau, hard, 1).squeeze(1)
    return (torch.logcumsumexp(mid.flip(1) * tau, dim=1) / tau).flip(1)


def evaluate(spec, x, tau, hard=False):
    """x: (n_sig, n, T) tensor -> robustness (n, T)."""
    ev = lambda s: evaluate(s, x, tau, hard)
    k = spec[0]
    if k == "ap":
        return x[spec[1]]
    if k == "not":
        return -ev(spec[1])
    if k == "and":
        return _smin(torch.stack([ev(spec[1]), ev(spec[2])], dim=1), tau, hard, 1).squeeze(1)
    if k == "or":
        return _smax(torch.stack([ev(spec[1]), ev(spec[2])], dim=1), tau, hard, 1).squeeze(1)
    if k == "imply":
        return _smax(torch.stack([-ev(spec[1]), ev(spec[2])], dim=1), tau, hard, 1).squeeze(1)
    if k == "listand":
        return _smin(torch.stack([ev(s) for s in spec[1]], dim=1), tau, hard, 1)[:, 0]
    if k == "alw":
        return _window(ev(spec[3]), spec[1], spec[2], tau, hard, _smin)
    if k in ("ev", "once"):
        return _window(ev(spec[3]), spec[1], spec[2], tau, hard, _smax)
    if k == "uu":
        return _untimed_until(ev(spec[1]), ev(spec[2]), tau, hard)
    if k == "until":
        ts, te = spec[1], spec[2]
        uu = _untimed_until(ev(spec[3]), ev(spec[4]), tau, hard)
        if ts == 0:
            return uu
        a = _window(ev(spec[4]), ts, te, tau, hard, _smax)
        b = _window(uu, 0, ts, tau, hard, _smin)
        return _smin(torch.stack([a, b], dim=1), tau, hard, 1).squeeze(1)
    raise ValueError(k)
