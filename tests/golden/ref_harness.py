"""Import the reference (CPU, this container only) so golden vectors can be generated from it.

The reference cannot travel to the GPU box; this module is only used by make_golden.py and by
tests that are skipped when /root/reference is absent.  Recipe = SURVEY.md section 8(c):
  (1) empty stub modules for the packages the image lacks (nuscenes devkit, imageio),
  (2) Tensor.cuda / Module.cuda patched to identity (the reference hard-codes .cuda()),
  (3) sys.argv set, generate_parser() called, nusc_train.args assigned (it is a module global there).
Nothing from the reference is copied; its functions are *called*.
"""
import contextlib
import os
import sys
import types

REF_ROOT = os.environ.get("PSTL_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "nusc_train.py"))


def _install_stubs():
    import matplotlib
    matplotlib.use("Agg")

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Dummy:  # base class for NuscenesPkl(NuScenes) in the reference's data tooling
        def __init__(self, *a, **k):
            pass

    stub("imageio")
    stub("nuscenes")
    stub("nuscenes.nuscenes", NuScenes=_Dummy, NuScenesExplorer=_Dummy)
    stub("nuscenes.map_expansion")
    stub("nuscenes.map_expansion.map_api", NuScenesMap=_Dummy)
    stub("nuscenes.map_expansion.arcline_path_utils")
    sys.modules["nuscenes.map_expansion"].arcline_path_utils = sys.modules["nuscenes.map_expansion.arcline_path_utils"]
    stub("nuscenes.utils")
    stub("nuscenes.utils.map_mask", MapMask=_Dummy)
    stub("nuscenes.utils.color_map", get_colormap=lambda *a, **k: {})


_REF = None


def load_reference():
    """Returns the imported reference modules as a namespace (nusc_train, nusc_model, stl_d_lib, nusc_api, utils)."""
    global _REF
    if _REF is not None:
        return _REF
    import torch
    _install_stubs()
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    old_argv = sys.argv
    sys.argv = ["nusc_train.py"]
    try:
        import nusc_train
        import nusc_model
        import stl_d_lib
        import nusc_api
        import utils as ref_utils
    finally:
        sys.argv = old_argv
    _REF = types.SimpleNamespace(nusc_train=nusc_train, nusc_model=nusc_model, stl_d_lib=stl_d_lib,
                                 nusc_api=nusc_api, utils=ref_utils)
    return _REF


def parse_reference_args(argv):
    """Run the reference's own parser (incl. its post-parse overrides) on argv and install the result as its global."""
    ref = load_reference()
    old_argv = sys.argv
    sys.argv = ["nusc_train.py"] + list(argv)
    try:
        args = ref.nusc_train.generate_parser()
    finally:
        sys.argv = old_argv
    ref.nusc_train.args = args
    return args


@contextlib.contextmanager
def record_randn_like(store):
    """Record every tensor torch.randn_like returns (x_T and each per-step z of the reference rollout)."""
    import torch
    orig = torch.randn_like

    def tapped(x, *a, **k):
        out = orig(x, *a, **k)
        store.append(out.detach().clone())
        return out

    torch.randn_like = tapped
    try:
        yield store
    finally:
        torch.randn_like = orig


@contextlib.contextmanager
def record_adam_grads(store):
    """Record the gradient every torch.optim.Adam.step() consumes (the guidance block builds a fresh Adam over mu_opt
    per guided step, reference nusc_train.py:606-623): one tensor per step() call, in call order."""
    import torch
    orig = torch.optim.Adam.step

    def tapped(self, *a, **k):
        for grp in self.param_groups:
            for p in grp["params"]:
                if p.grad is not None:
                    store.append(p.grad.detach().clone())
        return orig(self, *a, **k)

    torch.optim.Adam.step = tapped
    try:
        yield store
    finally:
        torch.optim.Adam.step = orig
