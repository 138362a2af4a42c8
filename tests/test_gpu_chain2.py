"""GPU tests of k_chain2 (csrc/chain2_kernels.hip): the multi-step denoiser launch with the rows stationary in registers
and the split-f16 weights streamed through LDS.  `chain_waves = 2` forces it for every launch it can take (any batch size);
the default `chain_waves = 0` hands it the batches that fill whole rounds of its 256-row workgroups.  Its arithmetic is
k_chain's default form in another summation order, so the two kernels agree to rounding, not bit for bit; everything else
it must share with k_chain: the reference's goldens within 1e-4 (tests/test_gpu_parity.py runs every fixture with
chain_waves = 2 too), the Philox stream, the emitted candidate list, shard invariance, the domain guard."""
import numpy as np
import pytest
import torch

from conftest import golden_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    from pstl_diffusion_policy_amd import ffi
    ffi.lib()
    return torch.device("cuda:0")


def _hp():
    from pstl_diffusion_policy_amd.synthetic import default_hparams
    return default_hparams()


def _setup(dev, bs, S, K, seed=5):
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    scene = make_scene_batch(bs, K=K, S=S, seed=seed, invalid_lane_frac=0.2, stlp_mode="wide")
    return hp, scene, PackedWeights(golden_weights(), dev), SceneBatch(scene, S, hp, dev)


# (scenes, samples): 3 S = rows per scene is a multiple of 16 and >= 48; the row counts are not multiples of the 256 rows a
# workgroup owns (a ragged last workgroup), and workgroups straddle 2..6 scenes
@pytest.mark.parametrize("bs,S", [(7, 16), (5, 64), (11, 32), (3, 48)])
@pytest.mark.parametrize("noise", ["tensor", "kernel"])
def test_chain2_agrees_with_chain(dev, bs, S, noise):
    """Same inputs through k_chain (chain_waves 16) and k_chain2 (2): every intermediate of the full list within 2e-5 (the
    two sum the same products in another order: observed <= 4e-6), identical shapes, nothing non-finite."""
    from pstl_diffusion_policy_amd.engine import Sampler
    hp, scene, w, sb = _setup(dev, bs, S, 3)
    steps = 12
    g = torch.Generator(device=dev).manual_seed(3)
    x_T = torch.randn(sb.N, 40, device=dev, generator=g) if noise == "tensor" else None
    z = torch.randn(steps - 1, sb.N, 40, device=dev, generator=g) if noise == "tensor" else None
    outs = []
    for cw in (16, 2):
        sm = Sampler(w, hp, chain_waves=cw)
        o = sm.sampling_region(sb, steps, x_T, z, rect_head=False, multi_cands=1, seed=99, full_list=True)
        assert not w.chain_overflowed()
        outs.append(o["controls_list"].clone())
    assert outs[0].shape == outs[1].shape and torch.isfinite(outs[1]).all()
    d = (outs[0] - outs[1]).abs().max().item()
    assert 0.0 < d <= 2e-5, d      # (> 0: the second run did take the other kernel)


def test_chain2_against_oracle_on_fresh_scenes(dev):
    """No fixture: seeded scenes, weights and noise; CPU oracle vs HIP with k_chain2 forced, e7 + guidance (the guided,
    single-step launches stay on k_chain: the segments between them are what k_chain2 runs)."""
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import Sampler, acc_from_counts
    bs, S, K, steps = 5, 16, 4, 14
    hp, scene, w, sb = _setup(dev, bs, S, K, seed=4242)
    sd = golden_weights()
    g = torch.Generator().manual_seed(7)
    N = bs * S * 3
    x_T = torch.randn(N, 40, generator=g)
    z = torch.randn(steps - 1, N, 40, generator=g)
    guid = dict(enabled=True, before=3, niters=2, lr=0.01)
    ref = orc.sampling_region(sd, {k: v.numpy() for k, v in scene.items()}, S, steps, hp, x_T, z, rect_head=True,
                              multi_cands=5, guidance=guid, n_rolls=1)
    sm = Sampler(w, hp, chain_waves=2)
    out = sm.sampling_region(sb, steps, x_T.to(dev), z.to(dev), rect_head=True, multi_cands=5, guidance=guid, n_rolls=1,
                             full_list=True)
    cl = out["controls_list"].reshape(steps, N, 20, 2).cpu()
    assert (cl - ref["controls_list"]).abs().max().item() <= 1e-4
    assert (out["final_controls"].reshape(N, 20, 2).cpu() - ref["final_controls"]).abs().max().item() <= 1e-4
    acc, sacc = acc_from_counts(out["counts"])
    assert abs(acc - float(ref["final_acc"])) <= 0.005 and abs(sacc - float(ref["final_scene_acc"])) <= 0.005


def test_chain2_in_kernel_noise_is_the_fill_normal_stream(dev):
    """k_chain2 draws its noise in the shadow of layer 2's MFMAs, as single-instruction steps of Philox4x32-7 + Box-Muller:
    the rollout must equal, bit for bit, the one fed with pstl_fill_normal's tensors (which the epilogue adds), and a block
    of scenes evaluated alone with the right row_offset must reproduce its rows of the full batch."""
    from pstl_diffusion_policy_amd.engine import Sampler, SceneBatch
    bs, S, K, steps, seed = 21, 64, 3, 9, 20240229
    hp, scene, w, sb = _setup(dev, bs, S, K, seed=8)
    sm = Sampler(w, hp, chain_waves=2)
    z = torch.stack([sm.fill_normal(sb, steps, i, seed) for i in range(steps - 1, 0, -1)])
    x_T = sm.fill_normal(sb, steps, steps, seed)
    guid = dict(enabled=True, before=3, niters=1, lr=0.01)
    a = sm.sampling_region(sb, steps, None, None, rect_head=True, multi_cands=4, guidance=guid, seed=seed, full_list=True)
    b = sm.sampling_region(sb, steps, x_T, z, rect_head=True, multi_cands=4, guidance=guid, full_list=True)
    for k in ["controls_list", "final_controls", "final_scores", "counts"]:
        assert torch.equal(a[k], b[k]), k
    lo, hi = 6, 17
    sub = {k: v[lo:hi].clone() for k, v in scene.items()}
    part = sm.sampling_region(SceneBatch(sub, S, hp, dev, row_offset=lo * S * 3, global_valid_sum=float(sb.valid.sum()),
                                         global_rows=sb.N), steps, None, None, rect_head=True, multi_cands=4,
                              guidance=guid, seed=seed)
    r0, r1 = lo * S * 3, hi * S * 3
    assert torch.equal(part["final_controls"], a["final_controls"][r0:r1])
    assert torch.equal(part["final_scores"], a["final_scores"][r0:r1])


def test_chain2_run_to_run_and_long_segments(dev):
    """Bitwise repeatable; a 150-step rollout (two launches: a launch covers at most 128 reverse steps) agrees with k_chain."""
    from pstl_diffusion_policy_amd.engine import Sampler
    hp, scene, w, sb = _setup(dev, 4, 64, 2)
    steps = 150
    outs = {}
    for cw in (2, 2, 16):
        sm = Sampler(w, hp, chain_waves=cw)
        o = sm.sampling_region(sb, steps, None, None, rect_head=False, multi_cands=3, seed=5)
        outs.setdefault(cw, []).append(o["final_controls"].clone())
    assert torch.equal(outs[2][0], outs[2][1])
    assert (outs[2][0] - outs[16][0]).abs().max().item() <= 5e-5


def test_chain2_domain_guard(dev):
    """A layer input outside the split-f16 domain (|x| >= 4094) sets the sticky word of the packed buffer's status block in
    k_chain2 as in k_chain: one hidden unit with bias 5000 and zero outgoing weights (exact for the fp32 kernels and the
    oracle; 5000 x 2^4 overflows a half piece, 0 x inf follows)."""
    import warnings
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    sd = {k: v.copy() for k, v in golden_weights().items()}
    sd["policy_net.0.bias"][17] = 5000.0
    sd["policy_net.2.weight"][:, 17] = 0.0
    w = PackedWeights(sd, dev)
    assert w.split_f16_ok
    scene = make_scene_batch(4, K=2, S=64, seed=1, stlp_mode="wide")
    sb = SceneBatch(scene, 64, hp, dev)
    sm = Sampler(w, hp, chain_waves=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = torch.randn(sb.N, 40, device=dev)
        _, base_p, _ = sm.encode(sb, need_rect=False)
        sm.rollout(sb, base_p, x, None, 8, n_emit=0, seed=3)
        assert w.chain_overflowed(clear=True)


# 134 400 and 200 000+ rows: more 256-row tiles than CUs, so that the single-step form's workgroups walk 2-4 tiles each (and
# the last workgroups one fewer); the second size is not a multiple of 256 and its scenes have 48 rows (7 scenes per tile)
@pytest.mark.parametrize("bs,S", [(700, 64), (4201, 16)])
def test_chain2_single_step_form_walks_tiles(dev, bs, S):
    """The guided phase's launches (mu_only = 1, one reverse step): k_chain2's form whose workgroups walk several tiles --
    next state by LDS-DMA into the wave's image, scene rows re-read per tile -- against k_chain's single-step layout on the
    same state: the plain launch, and the whole guided sampling region around it."""
    from pstl_diffusion_policy_amd.engine import Sampler, SceneBatch
    hp, scene, w, sb = _setup(dev, bs, S, 2, seed=77)
    assert sb.N > 2 * 256 * torch.cuda.get_device_properties(0).multi_processor_count
    g = torch.Generator(device=dev).manual_seed(11)
    x0 = torch.randn(sb.N, 40, device=dev, generator=g)
    import ctypes
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.engine import diffusion_coeffs
    steps, i = 10, 4
    beta, alpha, alpha_hat = diffusion_coeffs(steps, dev)
    mus = []
    for cw in (16, 2):
        sm = Sampler(w, hp, chain_waves=cw)
        _, base_p, _ = sm.encode(sb, need_rect=False)
        x = x0.clone()
        cfg = sb.cfg(steps, 0, cw, 0)
        ffi.check(sm.L.pstl_rollout(ctypes.byref(cfg), ffi.ptr(w.packed), ffi.ptr(base_p), ffi.ptr(w.tbias(steps)),
                                    ffi.ptr(sb.stlp), ffi.ptr(sb.hl), ffi.ptr(beta), ffi.ptr(alpha), ffi.ptr(alpha_hat),
                                    ffi.ptr(None), i, i, 1, ffi.ptr(x), ffi.ptr(sm.debug_buf), 0, ffi.stream()), "rollout")
        torch.cuda.synchronize()
        assert not w.chain_overflowed()
        mus.append(x)
    d = (mus[0] - mus[1]).abs().max().item()
    assert 0.0 < d <= 2e-6, d
    # a block of scenes launched alone (fewer tiles than CUs: every workgroup has ONE tile) gives the rows of the walk bit for bit
    lo, hi = bs // 3, bs // 3 + 40
    r0, r1 = lo * S * 3, hi * S * 3
    sub = SceneBatch({k: v[lo:hi].clone() for k, v in scene.items()}, S, hp, dev, row_offset=r0,
                     global_valid_sum=float(sb.valid.sum()), global_rows=sb.N)
    sm = Sampler(w, hp, chain_waves=2)
    _, base_s, _ = sm.encode(sub, need_rect=False)
    xs = x0[r0:r1].clone()
    cfg = sub.cfg(steps, 0, 2, 0)
    ffi.check(sm.L.pstl_rollout(ctypes.byref(cfg), ffi.ptr(w.packed), ffi.ptr(base_s), ffi.ptr(w.tbias(steps)),
                                ffi.ptr(sub.stlp), ffi.ptr(sub.hl), ffi.ptr(beta), ffi.ptr(alpha), ffi.ptr(alpha_hat),
                                ffi.ptr(None), i, i, 1, ffi.ptr(xs), ffi.ptr(sm.debug_buf), 0, ffi.stream()), "rollout")
    assert torch.equal(xs, mus[1][r0:r1])
    guid = dict(enabled=True, before=4, niters=1, lr=0.01)
    outs = []
    for cw in (16, 2):
        sm = Sampler(w, hp, chain_waves=cw)
        o = sm.sampling_region(sb, 9, None, None, rect_head=False, multi_cands=2, guidance=guid, seed=5)
        outs.append(o["final_controls"].clone())
    assert torch.isfinite(outs[1]).all()
    # the guided steps take a signed Adam step of lr = 0.01 per control: where a gradient component is zero to rounding, the last
    # bits of mu decide its sign, so a row in a thousand legitimately differs by up to 2 lr -- and no row by more
    dr = (outs[0] - outs[1]).abs().reshape(sb.N, -1).max(dim=1).values
    assert (dr > 5e-5).sum().item() <= max(1, sb.N // 400), (dr > 5e-5).sum().item()
    assert dr.max().item() <= 0.2      # (four guided steps, each may turn a sign: 4 x 2 lr, and what the denoiser makes of it)


# 48 rows (the smallest batch the kernel takes: one scene of 16 samples x 3 modes, a quarter of a wave's rows), 96, 432; steps = 3
# is a two-step launch, the shortest k_chain2 runs
@pytest.mark.parametrize("bs,S,steps", [(1, 16, 3), (2, 16, 6), (3, 48, 4), (1, 16, 150)])
def test_chain2_minimal_shapes_against_oracle(dev, bs, S, steps):
    """Workgroups that are mostly empty: rows behind the end repeat the last row and are never stored; with guidance on the
    last two steps the single-step form runs with one (partial) tile."""
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.engine import Sampler, acc_from_counts
    hp, scene, w, sb = _setup(dev, bs, S, 2, seed=31 + bs)
    N = sb.N
    g = torch.Generator().manual_seed(steps)
    x_T = torch.randn(N, 40, generator=g)
    z = torch.randn(steps - 1, N, 40, generator=g)
    guid = dict(enabled=True, before=2, niters=1, lr=0.01) if steps < 100 else None
    ref = orc.sampling_region(golden_weights(), {k: v.numpy() for k, v in scene.items()}, S, steps, hp, x_T, z, rect_head=True,
                              multi_cands=2, guidance=guid)
    out = Sampler(w, hp, chain_waves=2).sampling_region(sb, steps, x_T.to(dev), z.to(dev), rect_head=True, multi_cands=2, guidance=guid)
    assert not w.chain_overflowed()
    assert (out["final_controls"].reshape(N, 20, 2).cpu() - ref["final_controls"]).abs().max().item() <= 1e-4
    acc, sacc = acc_from_counts(out["counts"])
    assert abs(acc - float(ref["final_acc"])) <= 0.005 and abs(sacc - float(ref["final_scene_acc"])) <= 0.005


def test_chain2_under_hip_graph_replay(dev):
    """A captured sampling region whose denoiser launches are ALL k_chain2 (196 608 rows = three full rounds of its workgroups for
    the multi-step launch; the guided phase's single-step launches walk three tiles per CU), the seed read from device memory
    (cfg.dyn): two replays with different seeds equal the eager runs bit for bit, and the library says which kernel ran."""
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.engine import DynBlock, GraphCapture, PackedWeights, Sampler, SceneBatch
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    hp = _hp()
    bs, S, steps = 1024, 64, 8
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)      # chain_waves = 0: the dispatch rule chooses
    scene = {k: v.to(dev) for k, v in make_scene_batch(bs, K=2, S=S, seed=12, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("params", "pre_stlp", "tj_scores_prior")}
    N = bs * S * 3
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if N < 3 * 256 * cus or N % (256 * cus) != 0:
        pytest.skip("sized for 256 CUs")
    cfg = SceneBatch(scene, S, hp, dev).cfg(steps, ffi.PSTL_FLAG_RNG, 0, 0)
    assert ffi.rollout_layout(cfg, True)[0] == 2 and ffi.rollout_layout(cfg, False)[0] == 2
    vsum = float(sum(scene[k].sum().item() for k in ("curr_id", "left_id", "right_id"))) * S
    guid = dict(enabled=True, before=3, niters=1, lr=0.01)
    kw = dict(rect_head=True, multi_cands=2, guidance=guid, want_scores3=False)
    dyn = DynBlock(dev)
    dyn.set(0, SceneBatch.loss_scale(vsum, N))

    def body():
        sb = SceneBatch(scene, S, hp, dev, global_valid_sum=vsum, global_rows=N, dyn=dyn.dev, scale_in_dyn=True)
        o = sm.sampling_region(sb, steps, None, None, seed=0, **kw)
        return o["counts"], o["final_controls"], o["final_scores"]

    g = GraphCapture(body)
    for seed in (41, 42):
        dyn.set(seed)
        got = [t.clone() for t in g.replay()]
        ref = sm.sampling_region(SceneBatch(scene, S, hp, dev, global_valid_sum=vsum, global_rows=N), steps, None, None, seed=seed, **kw)
        for a, k in zip(got, ("counts", "final_controls", "final_scores")):
            assert torch.equal(a, ref[k]), (seed, k)
    assert not sm.w.chain_overflowed()


def _single_mu(dev, sm, w, sb, x0, cw, steps=10, i=4):
    import ctypes
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.engine import diffusion_coeffs
    beta, alpha, alpha_hat = diffusion_coeffs(steps, dev)
    _, base_p, _ = sm.encode(sb, need_rect=False)
    x = x0.clone()
    cfg = sb.cfg(steps, 0, cw, 0)
    ffi.check(sm.L.pstl_rollout(ctypes.byref(cfg), ffi.ptr(w.packed), ffi.ptr(base_p), ffi.ptr(w.tbias(steps)), ffi.ptr(sb.stlp),
                                ffi.ptr(sb.hl), ffi.ptr(beta), ffi.ptr(alpha), ffi.ptr(alpha_hat), ffi.ptr(None), i, i, 1,
                                ffi.ptr(x), ffi.ptr(sm.debug_buf), 0, ffi.stream()), "rollout")
    torch.cuda.synchronize()
    return x, ffi.rollout_layout(cfg, False)


def test_chain2_single_step_row_tiles_per_wave(dev):
    """The single-step form picks its rows per workgroup by what fills the CUs: 196 608 rows walk three 256-row tiles per CU,
    98 304 rows two 192-row tiles, 49 152 rows one 192-row tile each, 24 576 rows one 128-row tile each.  Against k_chain
    (<= 2e-6), and every smaller batch -- a shard of the larger one -- bit for bit against it."""
    from pstl_diffusion_policy_amd.engine import Sampler, SceneBatch
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("sized for 256 CUs")
    hp, scene, w, sb = _setup(dev, 1024, 64, 2, seed=91)
    g = torch.Generator(device=dev).manual_seed(5)
    x0 = torch.randn(sb.N, 40, device=dev, generator=g)
    sm2, sm16 = Sampler(w, hp, chain_waves=2), Sampler(w, hp, chain_waves=16)
    got, lay = _single_mu(dev, sm2, w, sb, x0, 2)
    assert lay == (2, 16, 3), lay                              # k_chain2, 16 sixteen-row tiles per workgroup, three walked
    ref, _ = _single_mu(dev, sm16, w, sb, x0, 16)
    assert 0.0 < (got - ref).abs().max().item() <= 2e-6
    vs = float(sb.valid.sum())
    prev = got
    for n_scn, want in ((512, (2, 12, 2)), (256, (2, 12, 1)), (128, (2, 8, 1))):
        part = SceneBatch({k: v[:n_scn].clone() for k, v in scene.items()}, 64, hp, dev, global_valid_sum=vs, global_rows=sb.N)
        gp, layp = _single_mu(dev, sm2, w, part, x0[:part.N], 2)
        assert layp == want, (n_scn, layp)
        assert torch.equal(gp, prev[:part.N]), n_scn
        prev = gp
    assert not w.chain_overflowed()


def test_chain2_192_row_workgroups(dev):
    """Multi-step launches take 192-row workgroups (three row tiles per wave) where rounds of them cost less than rounds of 256-row
    ones: 98 304 rows = 512 workgroups of 192 = two full rounds on 256 CUs.  Against k_chain on the same in-kernel noise (<= 2e-5),
    and bit for bit against the 256-row form that the whole batch of 196 608 rows gets, of which these rows are a shard."""
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.engine import Sampler, SceneBatch
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("sized for 256 CUs")
    hp, scene, w, sb = _setup(dev, 1024, 64, 2, seed=17)
    steps, seed = 9, 77
    vs, n_all = float(sb.valid.sum()), sb.N
    half = SceneBatch({k: v[:512].clone() for k, v in scene.items()}, 64, hp, dev, global_valid_sum=vs, global_rows=n_all)
    assert ffi.rollout_layout(sb.cfg(steps, ffi.PSTL_FLAG_RNG, 0, 0), True)[:2] == (2, 16)
    assert ffi.rollout_layout(half.cfg(steps, ffi.PSTL_FLAG_RNG, 0, 0), True)[:2] == (2, 12)
    outs = {}
    for name, b, cw in (("full", sb, 0), ("half", half, 0), ("half_k_chain", half, 16)):
        sm = Sampler(w, hp, chain_waves=cw)
        _, base_p, _ = sm.encode(b, need_rect=False)
        x = sm.fill_normal(b, steps, steps, seed)
        emit = sm.rollout(b, base_p, x, None, steps, n_emit=2, seed=seed)
        outs[name] = (x.clone(), emit.clone())
        assert not w.chain_overflowed()
    assert torch.equal(outs["half"][0], outs["full"][0][:half.N]) and torch.equal(outs["half"][1], outs["full"][1][:, :half.N])
    d = (outs["half"][0] - outs["half_k_chain"][0]).abs().max().item()
    assert 0.0 < d <= 2e-5, d


@pytest.mark.parametrize("diverse,clip_rect", [(True, False), (False, True), (True, True)])
def test_chain2_refine_form_against_k_chain(dev, diverse, clip_rect):
    """RefineNet's inference pass on k_chain2 (tile-walking form, rect_net's weights, init + pooled as input, the tanh interval
    head): against k_chain's on the same inputs (two summation orders of the same products: <= 1e-5 in physical controls of up to 5;
    observed 3e-6), rows the
    scores call satisfied are returned untouched in both, and a shard evaluated alone (one tile per workgroup, 128-row workgroups)
    reproduces the rows of the whole batch (three 256-row tiles walked per CU) bit for bit -- the pooled maxima are per scene, so a
    scene-aligned shard sees the same ones."""
    from pstl_diffusion_policy_amd.engine import Sampler, SceneBatch
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("sized for 256 CUs")
    hp, scene, w, sb = _setup(dev, 1024, 64, 2, seed=23)
    g = torch.Generator(device=dev).manual_seed(9)
    sc4 = torch.tensor([hp["mul_w_max"], hp["mul_a_max"]], device=dev).repeat(20)
    init = (torch.rand(sb.N, 40, device=dev, generator=g) * 2 - 1) * sc4 * 0.9
    scores = torch.randn(sb.N, device=dev, generator=g)
    outs = []
    for cw in (16, 2):
        sm = Sampler(w, hp, chain_waves=cw)
        _, _, base_r = sm.encode(sb, need_rect=True)
        outs.append(sm.refine(sb, base_r, init, scores, diverse=diverse, clip_rect=clip_rect))
        assert not w.chain_overflowed()
    d = (outs[0] - outs[1]).abs().max().item()
    assert 0.0 < d <= 1e-5, d
    keep = scores >= 0
    assert torch.equal(outs[1][keep], init[keep]) or clip_rect
    sm = Sampler(w, hp, chain_waves=2)
    for lo, hi in ((300, 428), (500, 756)):      # 24 576 rows: 128-row workgroups; 49 152 rows: 192-row workgroups
        sub = SceneBatch({k: v[lo:hi].clone() for k, v in scene.items()}, 64, hp, dev)
        _, _, base_s = sm.encode(sub, need_rect=True)
        r0, r1 = lo * 192, hi * 192
        part = sm.refine(sub, base_s, init[r0:r1].contiguous(), scores[r0:r1].contiguous(), diverse=diverse, clip_rect=clip_rect)
        assert torch.equal(part, outs[1][r0:r1]), (lo, hi)


def test_default_arithmetic_is_shard_invariant_under_a_job_wide_plan(dev):
    """with chain_waves = 0 the library picks k_chain or k_chain2 by batch size -- the same products in two
    summation orders -- so a shard evaluated alone could get other last bits than the same rows inside the whole batch.  A job
    names ONE row count for that choice (pstl_cfg.plan_rows = SceneBatch(plan_rows=...), shard.plan_rows: its largest shard):
    every shard then runs the kernel the others run, whatever its own size, and reproduces the batch's rows bit for bit in the
    DEFAULT mode.  256 scenes x 192 rows = 49 152 rows take k_chain2's multi-step launch; a 24-scene shard alone would take
    k_chain's latency layout -- without the plan it differs in the last bits (asserted too, so that the test cannot pass vacuously)."""
    from pstl_diffusion_policy_amd import ffi
    from pstl_diffusion_policy_amd.engine import Sampler, SceneBatch
    from pstl_diffusion_policy_amd.shard import plan_rows
    bs, S, K, steps, seed = 256, 64, 2, 9, 777
    hp, scene, w, sb = _setup(dev, bs, S, K, seed=12)
    sm = Sampler(w, hp, chain_waves=0)
    guid = dict(enabled=True, before=3, niters=1, lr=0.01)
    assert ffi.rollout_layout(sb.cfg(steps, ffi.PSTL_FLAG_RNG, 0))[0] == 2, "the whole batch is expected on k_chain2"
    full = sm.sampling_region(sb, steps, None, None, rect_head=True, multi_cands=4, guidance=guid, seed=seed)
    lo, hi = 100, 124
    sub = {k: v[lo:hi].clone() for k, v in scene.items()}
    r0, r1 = lo * S * 3, hi * S * 3
    kw = dict(row_offset=r0, global_valid_sum=float(sb.valid.sum()), global_rows=sb.N)
    alone = SceneBatch(sub, S, hp, dev, **kw)
    assert ffi.rollout_layout(alone.cfg(steps, ffi.PSTL_FLAG_RNG, 0))[0] != 2, "the shard alone is expected on k_chain"
    planned = SceneBatch(sub, S, hp, dev, plan_rows=sb.N, **kw)
    assert ffi.rollout_layout(planned.cfg(steps, ffi.PSTL_FLAG_RNG, 0))[0] == 2
    run = lambda b: sm.sampling_region(b, steps, None, None, rect_head=True, multi_cands=4, guidance=guid, seed=seed)
    a, p = run(alone), run(planned)
    for k in ("final_controls", "final_scores"):
        assert torch.equal(p[k], full[k][r0:r1]), k
    assert not torch.equal(a["final_controls"], full["final_controls"][r0:r1])
    assert (a["final_controls"] - full["final_controls"][r0:r1]).abs().max().item() <= 2e-4
    # the job-side helper: eight ranks over 256 scenes plan for 32 scenes' rows; seven scenes over eight ranks for one scene's
    assert plan_rows(256, 8, 3 * S) == 32 * 3 * S and plan_rows(7, 8, 3 * S) == 3 * S
