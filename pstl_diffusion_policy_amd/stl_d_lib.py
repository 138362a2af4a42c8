"""Differentiable STL formulas on MI355X -- the surface of the reference's stl_d_lib.py (same class names, constructor
arguments and call signature `formula(x, tau, d=None)` -> robustness (n, T)), evaluated by libpstl_hip.so.

How it runs: a formula tree is flattened once into a postfix node list (`pstl_stl_node`, include/pstl_hip.h).  A call
evaluates the leaves (AP expressions: user lambdas on torch GPU tensors, exactly as in the reference), stacks the
resulting (n, T) signals and hands the whole tree to ONE kernel launch (csrc/stl_program.hip); the adjoint is a second
launch behind a torch.autograd.Function, so `.backward()` keeps working through the AP expressions.
Reference lines: softmax/softmin stl_d_lib.py:6-26; STLFormula :28-68; AP :70-84; And :87; ListAnd :97; Or :113;
Not :125; Imply :132; Eventually :144; Always :157; Once :171; UntimedUntil :183; Until :195.

There is no CPU path: tensors must live on the GPU and the library must be built (ffi.lib() raises otherwise).
"""
import ctypes

import numpy as np
import torch

from . import ffi

OP_SIGNAL, OP_NOT, OP_AND, OP_OR, OP_LISTAND, OP_ALWAYS, OP_EVENTUALLY = range(7)
FLAG_SOFT = 1
MAX_T = 1024


def clip(x, a, b):
    return max(min(x, b), a)


# ---------------------------------------------------------------------------------------------------------------
# program = flattened formula
# ---------------------------------------------------------------------------------------------------------------
class _Program:
    """Postfix node list of one formula + the leaves whose expressions produce the input signals."""

    def __init__(self, root):
        self.nodes = []      # rows of (op, a, b, ts, te, n_list, list_off, flags)
        self.lists = []
        self.leaves = []     # leaf objects, signal i = leaves[i](x, tau, d)
        self._leaf_id = {}
        self.root_children = None
        self.root = self._lower(root)
        self._dev = {}

    def _emit(self, op, a=0, b=0, ts=0, te=0, children=None, flags=0):
        n_list, off = 0, 0
        if children is not None:
            n_list, off = len(children), len(self.lists)
            self.lists.extend(children)
        self.nodes.append((op, a, b, ts, te, n_list, off, flags))
        return len(self.nodes) - 1

    def _untimed_until(self, lhs, rhs):
        ls, rs = self._lower(lhs), self._lower(rhs)
        inf_ls = self._emit(OP_ALWAYS, ls, ts=-MAX_T, te=1, flags=FLAG_SOFT)       # running soft min of lhs
        mid = self._emit(OP_AND, rs, inf_ls)
        return self._emit(OP_EVENTUALLY, mid, ts=0, te=MAX_T, flags=FLAG_SOFT)     # suffix soft max

    def _lower(self, f):
        if isinstance(f, And):
            return self._emit(OP_AND, self._lower(f.lhs), self._lower(f.rhs))
        if isinstance(f, Or):
            return self._emit(OP_OR, self._lower(f.lhs), self._lower(f.rhs))
        if isinstance(f, Imply):
            return self._emit(OP_OR, self._emit(OP_NOT, self._lower(f.lhs)), self._lower(f.rhs))
        if isinstance(f, Not):
            return self._emit(OP_NOT, self._lower(f.node))
        if isinstance(f, ListAnd):
            ch = [self._lower(c) for c in f.lists]
            return self._emit(OP_LISTAND, children=ch)
        if isinstance(f, Always):
            return self._emit(OP_ALWAYS, self._lower(f.node), ts=f.ts, te=f.te)
        if isinstance(f, (Eventually, Once)):
            return self._emit(OP_EVENTUALLY, self._lower(f.node), ts=f.ts, te=f.te)
        if isinstance(f, UntimedUntil):
            return self._untimed_until(f.lhs, f.rhs)
        if isinstance(f, Until):
            if f.ts == 0:
                return self._untimed_until(f.lhs, f.rhs)
            ev = self._emit(OP_EVENTUALLY, self._lower(f.rhs), ts=f.ts, te=f.te)
            al = self._emit(OP_ALWAYS, self._untimed_until(f.lhs, f.rhs), ts=0, te=f.ts)
            return self._emit(OP_AND, ev, al)
        if isinstance(f, STLFormula):
            raise NotImplementedError("no lowering for %s" % type(f).__name__)
        # a leaf: AP, or any callable with the (x, tau, d) signature
        key = id(f)
        if key not in self._leaf_id:
            self._leaf_id[key] = len(self.leaves)
            self.leaves.append(f)
        return self._emit(OP_SIGNAL, self._leaf_id[key])

    def device_tables(self, dev):
        if dev not in self._dev:
            nodes = torch.tensor(np.asarray(self.nodes, dtype=np.int32).reshape(-1, 8), dtype=torch.int32, device=dev)
            lists = torch.tensor(self.lists if self.lists else [0], dtype=torch.int32, device=dev)
            self._dev[dev] = (nodes.contiguous(), lists.contiguous())
        return self._dev[dev]


class _StlProgramFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prog, tau, hard, keep, signals):
        dev = signals.device
        n_sig, n, T = signals.shape
        nodes, lists = prog.device_tables(dev)
        n_nodes = nodes.shape[0]
        vals = torch.empty(n_nodes, T, n, dtype=torch.float32, device=dev)
        out = torch.empty(n, T, dtype=torch.float32, device=dev)
        ffi.check(ffi.lib().pstl_stl_program_forward(ffi.ptr(nodes, torch.int32), int(n_nodes), ffi.ptr(lists, torch.int32),
                                                     ctypes.c_int64(n), int(T), ffi.ptr(signals), ctypes.c_float(tau),
                                                     int(hard), ffi.ptr(vals), ffi.ptr(out), ffi.stream()),
                  "stl_program_forward")
        ctx.prog, ctx.tau, ctx.hard, ctx.shape = prog, float(tau), int(hard), (n_sig, n, T)
        ctx.save_for_backward(vals)
        if keep is not None:
            keep["vals"] = vals
        return out

    @staticmethod
    def backward(ctx, dout):
        (vals,) = ctx.saved_tensors
        n_sig, n, T = ctx.shape
        dev = vals.device
        nodes, lists = ctx.prog.device_tables(dev)
        dsig = torch.zeros(n_sig, n, T, dtype=torch.float32, device=dev)
        adj = torch.empty_like(vals)
        dout_c = ffi.f32(dout, dev)          # named: a contiguous copy must outlive the launch
        ffi.check(ffi.lib().pstl_stl_program_backward(ffi.ptr(nodes, torch.int32), int(nodes.shape[0]),
                                                      ffi.ptr(lists, torch.int32), ctypes.c_int64(n), int(T), ffi.ptr(vals),
                                                      ctypes.c_float(ctx.tau), ctx.hard,
                                                      ffi.ptr(dout_c), ffi.ptr(adj), ffi.ptr(dsig), ffi.stream()),
                  "stl_program_backward")
        return None, None, None, None, dsig


def _evaluate(formula, x, tau, d, want_children=False):
    prog = formula.__dict__.get("_pstl_program")
    if prog is None:
        prog = _Program(formula)
        formula.__dict__["_pstl_program"] = prog
    sigs = [leaf(x, tau, d) for leaf in prog.leaves]
    if not sigs[0].is_cuda:
        raise RuntimeError("stl_d_lib formulas are evaluated by libpstl_hip.so on the GPU; got a %s tensor" % sigs[0].device)
    sigs = torch.broadcast_tensors(*sigs)
    if sigs[0].dim() != 2:
        raise ValueError("AP expressions must produce (n, T) signals, got %s" % (tuple(sigs[0].shape),))
    if sigs[0].shape[1] > MAX_T:
        raise ValueError("T = %d exceeds PSTL_STL_MAX_T" % sigs[0].shape[1])
    signals = torch.stack([s.float() for s in sigs], dim=0).contiguous()
    hard = 1 if (d is not None and d.get("hard", False)) else 0
    keep = {} if want_children else None
    out = _StlProgramFn.apply(prog, float(tau), hard, keep, signals)
    if want_children:
        ch = prog.lists[prog.nodes[prog.root][6]:prog.nodes[prog.root][6] + prog.nodes[prog.root][5]]
        v = keep["vals"][ch].permute(2, 0, 1).contiguous()      # (n, k, T), what torch.stack(v, dim=1) gives
        return out, v
    return out


# ---------------------------------------------------------------------------------------------------------------
# the reference's class surface
# ---------------------------------------------------------------------------------------------------------------
class STLFormula:
    def __init__(self, ts=None, te=None, node=None, lhs=None, rhs=None, lists=None, operator=None):
        self.ts, self.te, self.node, self.lhs, self.rhs, self.lists = ts, te, node, lhs, rhs, lists
        self.operator = operator
        self.format = "symbol"

    def __call__(self, x, tau, d=None):
        return _evaluate(self, x, tau, d)

    def __str__(self):
        ops = self.operator[self.format]
        if self.ts is not None:
            ops = "%s[%d:%d]" % (ops, self.ts, self.te + 1)
        if self.node is not None:
            return "%s (%s)" % (ops, self.node)
        if self.lhs is not None:
            return "(%s) %s (%s)" % (self.lhs, ops, self.rhs)
        if self.lists is not None:
            return "%s {%s}" % (ops, ",".join("|%s|" % c for c in self.lists))
        raise NotImplementedError

    def children(self):
        if self.node is not None:
            return [self.node]
        if self.lists is not None:
            return list(self.lists)
        return [self.lhs, self.rhs]

    def update_format(self, format):
        self.format = format
        for child in self.children():
            if hasattr(child, "update_format"):
                child.update_format(format)


class AP:
    n_aps = 0

    def __init__(self, expression, comment=None):
        self.expression, self.comment = expression, comment
        self.apid = AP.n_aps
        AP.n_aps += 1

    def __call__(self, x, tau=None, d=None):
        return self.expression(x)

    def __str__(self):
        return "AP%d" % self.apid if self.comment is None else self.comment


class And(STLFormula):
    def __init__(self, lhs, rhs):
        super().__init__(lhs=lhs, rhs=rhs, operator={"symbol": "&", "word": "AND"})


class ListAnd(STLFormula):
    def __init__(self, lists):
        super().__init__(lists=lists, operator={"symbol": "&", "word": "AND"})

    def __call__(self, x, tau, d=None, full=False):
        if full:
            return _evaluate(self, x, tau, d, want_children=True)
        return _evaluate(self, x, tau, d)


class Or(STLFormula):
    def __init__(self, lhs, rhs):
        super().__init__(lhs=lhs, rhs=rhs, operator={"symbol": "|", "word": "OR"})


class Not(STLFormula):
    def __init__(self, node):
        super().__init__(node=node, operator={"symbol": "¬", "word": "NOT"})


class Imply(STLFormula):
    def __init__(self, lhs, rhs):
        super().__init__(lhs=lhs, rhs=rhs, operator={"symbol": "->", "word": "IMPLY"})
        self.eval = Or(Not(self.lhs), self.rhs)


class Eventually(STLFormula):
    def __init__(self, ts, te, node):
        super().__init__(ts=ts, te=te, node=node, operator={"symbol": "♢", "word": "EVENTUALLY"})


class Always(STLFormula):
    def __init__(self, ts, te, node):
        super().__init__(ts=ts, te=te, node=node, operator={"symbol": "◻", "word": "ALWAYS"})


class Once(STLFormula):
    def __init__(self, ts, te, node):
        super().__init__(ts=ts, te=te, node=node, operator={"symbol": "O", "word": "ONCE"})
        assert ts < 0 and te >= ts and te <= 0


class UntimedUntil(STLFormula):
    def __init__(self, lhs, rhs):
        super().__init__(lhs=lhs, rhs=rhs, operator={"symbol": "U", "word": "UNTIL"})


class Until(STLFormula):
    def __init__(self, ts, te, lhs, rhs):
        super().__init__(ts=ts, te=te, lhs=lhs, rhs=rhs, operator={"symbol": "U", "word": "UNTIL"})
        if ts == 0:
            self.eval = UntimedUntil(lhs, rhs)
        else:
            self.eval = And(Eventually(ts, te, rhs), Always(0, ts, UntimedUntil(lhs, rhs)))


# ---------------------------------------------------------------------------------------------------------------
# the four module-level helpers of the reference (stl_d_lib.py:6-26): the same kernel, as one-node programs
# ---------------------------------------------------------------------------------------------------------------
_ROW_MAX, _PAIR_MAX = {}, None


def softmax(x, tau, d, dim=1):
    """Soft (d['hard']: hard) maximum of x (n, m) over its second axis -> (n, 1): an Eventually node spanning the whole
    row, read at t = 0.  An empty row gives -inf, as in the reference."""
    if x.dim() != 2 or dim != 1:
        raise NotImplementedError("softmax is evaluated by the STL kernel for (n, m) inputs over dim 1")
    m = x.shape[1]
    if m == 0:
        return torch.ones(x.shape[0], 1).to(x.device) * -float("inf")
    if m not in _ROW_MAX:
        _ROW_MAX[m] = Eventually(0, m, AP(lambda s: s, comment="x"))
    return _evaluate(_ROW_MAX[m], x, tau, d)[:, :1]


def softmin(x, tau, d, dim=1):
    if x.shape[1] == 0:
        return torch.ones(x.shape[0], 1).to(x.device) * -float("inf")
    return -softmax(-x, tau, d, dim)


def softmax_pairs(x, y, tau, d):
    """Elementwise soft maximum of two (n, T) signals: an Or node."""
    global _PAIR_MAX
    if _PAIR_MAX is None:
        _PAIR_MAX = Or(AP(lambda s: s[0], comment="x"), AP(lambda s: s[1], comment="y"))
    return _evaluate(_PAIR_MAX, (x, y), tau, d)


def softmin_pairs(x, y, tau, d):
    return -softmax_pairs(-x, -y, tau, d)
