"""CPU tests of the host side: library export table vs the header, schedule, guidance trigger, metric arithmetic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, golden_meta


def test_library_builds_loads_and_exports_every_declared_symbol():
    from pstl_diffusion_policy_amd import build, ffi
    build.build(verbose=False)
    L = ffi.lib()
    header = open(os.path.join(ROOT, "include", "pstl_hip.h")).read()
    assert L.pstl_version() == ffi.ABI_VERSION == int(re.search(r"#define PSTL_ABI_VERSION (\d+)", header).group(1))
    declared = sorted(set(re.findall(r"^(?:int|size_t|const char\*)\s+(pstl_\w+)\s*\(", header, flags=re.M)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(ffi.EXPORTS) == declared
    assert float(re.search(r"#define PSTL_SPLIT_F16_WMAX ([0-9.]+)f", header).group(1)) == ffi.SPLIT_F16_WMAX
    assert L.pstl_packed_weight_floats() > 540952          # at least the reference parameter count
    assert L.pstl_error_string(-2).decode().startswith("shape")
    assert ctypes.sizeof(ffi.PstlCfg) == 96      # ABI 4: + the pstl_dyn device pointer; ABI 6: + plan_rows


def test_binding_signatures_match_the_header_prototypes():
    """ffi.SIGNATURES (restype + argtypes of every export) against the prototypes of include/pstl_hip.h: argument count
    and kind (pointer / int / float / int64), so that the table cannot drift from the C side."""
    from pstl_diffusion_policy_amd import ffi
    header = open(os.path.join(ROOT, "include", "pstl_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = re.findall(r"^(int|size_t|const char\*)\s+(pstl_\w+)\s*\(([^;]*?)\)\s*;", header, flags=re.M | re.S)
    assert len(protos) == len(ffi.EXPORTS)
    table = {n: (r, a) for n, r, a in ffi.SIGNATURES}
    kinds = {ffi._I: "i", ffi._F: "f", ffi._L: "l", ffi._P: "p", ffi._C: "p", ctypes.POINTER(ffi.WeightPtrs): "p"}
    for ret, name, args in protos:
        want = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            want.append("p" if "*" in a else "f" if a.startswith("float") else "l" if a.startswith("int64_t") else "i")
        restype, argtypes = table[name]
        assert [kinds[t] for t in argtypes] == want, name
        assert restype is {"int": ffi._I, "size_t": ffi._Z, "const char*": ctypes.c_char_p}[ret], name


def test_wrong_argument_types_are_a_typeerror():
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    L = ffi.lib()
    cfg = ffi.make_cfg(2, 24, 8, 3, 10, default_hparams())
    null = ctypes.c_void_p(0)
    with pytest.raises((TypeError, ctypes.ArgumentError)):
        L.pstl_fill_normal(ctypes.byref(cfg), 1.5, null, null)          # a float where `int step` is expected
    with pytest.raises((TypeError, ctypes.ArgumentError)):
        L.pstl_fill_normal(ctypes.byref(cfg), 1, null)                  # an argument short
    with pytest.raises((TypeError, ctypes.ArgumentError)):
        L.pstl_fill_normal(ffi.Mlp3(), 1, null, null)                   # not a pstl_cfg


def test_graft_entry_build_succeeds():
    """__graft_entry__.build() asserted a stale ABI version.  It must compile everything and import cleanly."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    ge = importlib.import_module("__graft_entry__")
    ge.build()
    assert os.path.exists(os.path.join(ROOT, "tests", "hostsim", "libpstl_hostsim.so"))


def test_unknown_chain_arithmetic_is_refused():
    """the timing-only diagnostic instantiations (chain_waves 108 / 116 / 708 / 716 / 1008) are not in the shipped
    library (-DPSTL_DIAG builds only)."""
    src = open(os.path.join(ROOT, "pstl_diffusion_policy_amd", "csrc", "mlp_kernels.hip")).read()
    body = src[src.index("int launch_chain_nw("):]
    body = body[:body.index("}  // namespace")]
    diag = body[body.index("#ifdef PSTL_DIAG"):body.index("#else")]
    for v in ("108", "116", "708", "716", "1008"):
        assert "case %s:" % v in diag and "case %s:" % v not in body.replace(diag, "")
    from pstl_diffusion_policy_amd import build
    assert not any("PSTL_DIAG" in f for unit in build.UNITS for f in unit[1])


def test_missing_library_fails_loudly(monkeypatch):
    from pstl_diffusion_policy_amd import ffi
    monkeypatch.setattr(ffi, "_lib", None)
    monkeypatch.setattr(ffi, "LIB_PATH", "/nonexistent/libpstl_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ffi.lib()


def test_null_and_bad_arguments_are_rejected_without_a_gpu():
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    L = ffi.lib()
    cfg = ffi.make_cfg(2, 24, 8, 3, 10, default_hparams())
    null = ctypes.c_void_p(0)
    assert L.pstl_rollout(ctypes.byref(cfg), null, null, null, null, null, null, null, null, null, 9, 1, 0, null, null, 0,
                          null) == -1
    assert L.pstl_stl_forward(ctypes.byref(cfg), null, null, null, 1, null, null, null, null, null, null, null, null,
                              null, null) == -1
    assert L.pstl_pack_weights(None, null, null) == -1


def test_schedule_is_bit_identical_to_the_reference():
    from pstl_diffusion_policy_amd.engine import diffusion_coeffs
    for name in ["e5_steps10", "e5_steps100", "e7_steps50_k8"]:
        d = load_golden(name)
        b, a, ah = diffusion_coeffs(golden_meta(d)["steps"])
        np.testing.assert_array_equal(b.numpy(), d["coef_beta"])
        np.testing.assert_array_equal(a.numpy(), d["coef_alpha"])
        np.testing.assert_array_equal(ah.numpy(), d["coef_alpha_hat"])


def test_guidance_trigger_rule():
    from pstl_diffusion_policy_amd.engine import guidance_triggered as trig
    steps = 100
    assert not trig(5, steps, None)
    g = dict(enabled=True, before=10)
    assert [i for i in range(1, steps) if trig(i, steps, g)] == list(range(1, 11))
    g = dict(enabled=True, freq=25)
    assert [i for i in range(1, steps) if trig(i, steps, g)] == [25, 50, 75]
    g = dict(enabled=True, sets=[3, 7], before=1000)
    assert [i for i in range(1, steps) if trig(i, steps, g)] == [3, 7]
    g = dict(enabled=True, sets=[0, 1], reverse=True)
    assert [i for i in range(1, steps) if trig(i, steps, g)] == [98, 99]
    assert all(trig(i, steps, dict(enabled=True)) for i in range(1, steps))     # reference default: before=1000


def test_metric_arithmetic_matches_mask_mean():
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import acc_from_counts
    g = torch.Generator().manual_seed(0)
    bs, S = 7, 8
    scores = torch.randn(bs * S * 3, generator=g)
    ids = (torch.rand(bs, 3, generator=g) < 0.7).float()
    valid = ids[:, None].repeat(1, S, 1).reshape(-1)
    acc, sacc = orc.stl_metrics(scores, valid, S)
    sat = ((scores > 0) & (valid > 0)).sum().item()
    cube = scores.reshape(bs, S, 3)
    ssat = (((cube.max(dim=1)[0] > 0)) & (ids > 0)).sum().item()
    counts = torch.tensor([sat, int(valid.sum()), bs * S * 3, ssat, int(ids.sum()), bs * 3, 0, 0])
    a, s = acc_from_counts(counts)
    assert a == float(acc) and s == float(scene_acc) if False else True
    assert a == float(acc) and s == float(sacc)


def test_bench_refuses_to_measure_fewer_ranks_than_asked_for():
    """`python bench.py --gpus N` must never print a 1-rank number.  Without GPUs it exits non-zero before
    anything is measured; a launcher that started a different number of ranks is an error too."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "launcher started 1 rank" in r.stderr


def test_packed_weight_key_sees_every_way_a_parameter_can_change():
    """Net.packed() must re-pack when ANY parameter changes -- in place (optimiser step, load_state_dict's copy_),
    by `p.data = t` on a middle parameter, by replacing a Parameter or a sub-module, by load_state_dict(assign=True).  The key
    it compares is checked here on the CPU (the packing itself needs the GPU)."""
    import types

    import torch
    from pstl_diffusion_policy_amd.nusc_model import Net
    args = types.SimpleNamespace(diffusion=True, hiddens=[256, 256], nt=20, n_segs=15, rect_head=True, diverse_loss=True,
                                 no_arch=False, diverse_fuse_type="add", rect_hiddens=[256, 256], use_init_hint=False)
    net = Net(args)
    ps, k0 = net._pack_key()
    assert [id(p) for p in ps] == [id(p) for p in net.parameters()]      # the direct walk IS parameters(), in order
    assert net._pack_key()[1] == k0                                      # stable while nothing changes
    keys = [k0]

    def changed():
        k = net._pack_key()[1]
        assert k not in keys
        keys.append(k)

    with torch.no_grad():
        net.policy_net[2].weight.add_(1.0)                               # in place (an optimiser step)
    changed()
    net.policy_net[2].bias.data = torch.zeros(256)                       # p.data = t on a middle parameter
    changed()
    net.merge_net[0].weight = torch.nn.Parameter(torch.zeros(32, 40))    # a replaced Parameter
    changed()
    net.rect_net = Net(args).rect_net                                    # a replaced sub-module
    changed()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(sd, assign=True)                                 # storages swapped wholesale
    changed()
    net.load_state_dict({k: v + 1 for k, v in sd.items()})               # the usual copy_ into the existing storages
    changed()


def test_bench_guided_gate_counts_and_explains_outlier_rows():
    """bench.py's in-run version of the guided-outlier gate (tests/conftest.py: guided_outlier_rows) on
    synthetic errors: a row that leaves 1e-4 at a guided step on an element whose recorded |g| is in Adam's eps regime is
    excluded and explained; one that leaves it at an un-guided step, or on an element with a solid gradient, is not explained."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    steps, N = 12, 6
    guid = dict(enabled=True, before=3, niters=1, lr=0.01)          # guided reverse steps: 3, 2, 1
    grads = [torch.full((N, 20, 2), 1e-3) for _ in range(3)]        # one Adam.step() per guided step, in rollout order
    grads[0][2, 5, 1] = 3e-8                                          # row 2, element 11 at reverse step 3: eps regime
    err = torch.zeros(steps, N, 40)
    err[steps - 3:, 2, 11] = 5e-3                                     # leaves 1e-4 in the state after reverse step 3, stays off
    bad, explained, tiny = bench.guided_gate(err, grads, steps, guid)
    assert bad.tolist() == [False, False, True, False, False, False] and explained and tiny == 1
    err2 = err.clone()
    err2[steps - 3:, 4, 7] = 2e-4                                     # a solid gradient there: not Adam's discontinuity
    bad, explained, _ = bench.guided_gate(err2, grads, steps, guid)
    assert bad.tolist()[4] and not explained
    err3 = torch.zeros(steps, N, 40)
    err3[4:, 1, 0] = 1e-3                                             # leaves 1e-4 after reverse step 8: not a guided step
    bad, explained, _ = bench.guided_gate(err3, grads, steps, guid)
    assert bad.tolist()[1] and not explained


def test_shard_plan_rows_and_device_identity():
    from pstl_diffusion_policy_amd import shard
    assert shard.plan_rows(4096, 8, 192) == 512 * 192 and shard.plan_rows(7, 8, 192) == 192 and shard.plan_rows(9, 2, 48) == 5 * 48
    ident, text = shard.device_identity(None)
    assert len(ident) == 2 and all(-2 ** 63 <= v < 2 ** 63 for v in ident) and text.startswith("host-pid:")
    assert shard.device_identity(None)[0] == ident                  # stable within a process
