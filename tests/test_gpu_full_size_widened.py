"""BASELINE.json's single-GPU size (4096 scenes x 64 samples x 3 modes = 786432 rows) for the widened rows, checked
through size-independent properties (the oracle cannot run at this size): shard additivity and bitwise reproducibility of
the diversity totals; the traj-opt loop split over calls and over scene shards; a generic formula tree against the fused
STL kernel on the same rows; the DPP loss invariant under a permutation of the groups."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import golden_weights

pytestmark = pytest.mark.gpu
BS, S, K = 4096, 64, 2


@pytest.fixture(scope="module")
def ctx():
    assert torch.cuda.is_available(), "the gpu-marked tests need a GPU"
    from pstl_diffusion_policy_amd.engine import PackedWeights, SceneBatch, Sampler
    from pstl_diffusion_policy_amd.synthetic import default_hparams, make_scene_batch
    dev = torch.device("cuda:0")
    hp = default_hparams()
    scene = {k: v.to(dev) for k, v in make_scene_batch(BS, K=K, S=S, seed=3, invalid_lane_frac=0.2, stlp_mode="wide").items()
             if k not in ("pre_stlp", "tj_scores_prior")}
    sm = Sampler(PackedWeights(golden_weights(), dev), hp)
    sb = SceneBatch(scene, S, hp, dev)
    out = sm.sampling_region(sb, 12, None, None, rect_head=True, multi_cands=3, seed=5, want_scores3=True, diversity=True)
    torch.cuda.synchronize()
    return dict(dev=dev, hp=hp, scene=scene, sm=sm, sb=sb, out=out)


def _sub(ctx, lo, hi, **kw):
    from pstl_diffusion_policy_amd.engine import SceneBatch
    sub = {k: v[lo:hi].contiguous() for k, v in ctx["scene"].items()}
    return SceneBatch(sub, S, ctx["hp"], ctx["dev"], **kw)


def test_diversity_totals_full_size(ctx):
    from pstl_diffusion_policy_amd.engine import diversity_from_totals
    sm, sb, out = ctx["sm"], ctx["sb"], ctx["out"]
    N = sb.N
    assert N == 786432
    pm, ps, tot = out["div_per_mode"], out["div_per_scene"], out["div_totals"]
    pm2, ps2, tot2 = sm.diversity(sb, out["final_controls"], out["final_scores"])
    assert torch.equal(pm, pm2) and torch.equal(ps, ps2) and torch.equal(tot, tot2)          # bitwise reproducible
    assert torch.isfinite(pm).all() and torch.isfinite(ps).all()
    acc = torch.zeros(12, dtype=torch.float64, device=ctx["dev"])
    for q in range(4):                                                                        # four shards, as four GPUs
        lo, hi = q * BS // 4, (q + 1) * BS // 4
        r0, r1 = lo * S * 3, hi * S * 3
        p, s_, t = sm.diversity(_sub(ctx, lo, hi), out["final_controls"][r0:r1].contiguous(), out["final_scores"][r0:r1].contiguous())
        assert torch.equal(p, pm[lo:hi]) and torch.equal(s_, ps[lo:hi])
        acc += t
    np.testing.assert_allclose(acc.cpu().numpy(), tot.cpu().numpy(), rtol=1e-12)
    d = diversity_from_totals(tot)
    sat = (out["final_scores"] > 0).reshape(BS, S, 3).sum(dim=1)
    assert torch.equal(pm[:, :, 6].long(), sat)                                              # satisfied-sample counts
    assert torch.equal(pm[:, :, 7], sb.valid.reshape(BS, S, 3)[:, 0].double())
    assert (pm[:, :, 1][pm[:, :, 6] < 3] == 0).all()                                         # no hull below 3 points
    assert 0 <= d["ent_s"] <= np.log2(10) + 1e-6 and 0 <= d["ent_w"] <= np.log2(10) + 1e-6
    assert d["std"] > 0 and d["vol"] > 0 and d["area"] > 0 and d["fde"] >= 0


def test_generic_formulas_agree_with_the_fused_kernel_full_size(ctx):
    """build_stl_cache's formula objects (generic program kernel) on signals from pstl_stl_signals reproduce the three
    scores of the fused kernel on all 786432 rows."""
    from pstl_diffusion_policy_amd import nusc_train as nt
    sm, sb, out = ctx["sm"], ctx["sb"], ctx["out"]
    args = nt.generate_parser(["--diffusion", "--load_stlp"])
    traj = sm.trajs(sb, out["final_controls"])[:, :-1].contiguous()
    x = nt.prep_stl_cache({"ego_traj": traj, "_pstl": sb}, args)
    stls = nt.build_stl_cache(args)
    for m in range(3):
        got = stls[m](x, args.smoothing_factor)[:, 0]
        want = out["final_scores3"][m]
        err = (got - want).abs()
        assert err.max().item() <= 5e-4 + 1e-4 * want.abs().max().item(), (m, err.max().item())
        assert ((got > 0) == (want > 0))[want.abs() > 1e-3].all()


def test_satisfaction_masks_full_size_two_evaluators(ctx):
    """nothing pinned the satisfaction masks at 786 432 rows.  The mode-selected score of every row
    through the fused kernel (k_stl_forward: closed-form sweeps) and through the generic formula-tree evaluator
    (k_stl_program on pstl_stl_signals' signals: the stl_d_lib composition) -- two independent implementations of
    compute_stl_dense -- must give the SAME mask `score > 0` on every row whose score is not within 2e-5 of zero, and the rows
    inside that band are counted (they are where float32 rounding decides, in the reference too)."""
    from pstl_diffusion_policy_amd import nusc_train as nt
    sm, sb, out = ctx["sm"], ctx["sb"], ctx["out"]
    N = sb.N
    args = nt.generate_parser(["--diffusion", "--load_stlp"])
    traj = sm.trajs(sb, out["final_controls"])[:, :-1].contiguous()
    x = nt.prep_stl_cache({"ego_traj": traj, "_pstl": sb}, args)
    stls = nt.build_stl_cache(args)
    generic3 = torch.stack([stls[m](x, args.smoothing_factor)[:, 0] for m in range(3)])          # (3, N)
    mode = sb.hl.long().clamp(0, 2)
    generic = generic3.gather(0, mode.reshape(1, N))[0]
    fused = out["final_scores"]
    assert N == 786432 and fused.shape == generic.shape
    differ = (fused > 0) != (generic > 0)
    band = fused.abs() < 2e-5
    assert not differ[~band].any(), "masks differ outside the rounding band: %d rows" % int(differ[~band].sum())
    assert int(band.sum()) <= 64 and int(differ.sum()) <= int(band.sum()), (int(band.sum()), int(differ.sum()))
    # both masks give the same satisfaction counters unless a row sits in the band
    if int(differ.sum()) == 0:
        c1 = sm.metrics(sb, fused)[0]
        c2 = sm.metrics(sb, generic.contiguous())[0]
        assert torch.equal(c1, c2)


def test_trajopt_full_size_split_and_sharded(ctx):
    sm, sb = ctx["sm"], ctx["sb"]
    N = sb.N
    p0 = ctx["scene"]["params"].reshape(N, 40).contiguous()
    vsum = float(sb.valid.sum())
    a = p0.clone()
    sc_a, _ = sm.trajopt(sb, a, 6, 0.01)
    b = p0.clone()                                                      # same run split over two calls
    _, work = sm.trajopt(sb, b, 2, 0.01)
    sc_b, _ = sm.trajopt(sb, b, 4, 0.01, work=work, first_iter=2)
    assert torch.equal(a, b) and torch.equal(sc_a, sc_b)
    for q in (0, 3):                                                    # a quarter of the scenes alone, global constants
        lo, hi = q * BS // 4, (q + 1) * BS // 4
        r0, r1 = lo * S * 3, hi * S * 3
        c = p0[r0:r1].clone()
        sc_c, _ = sm.trajopt(_sub(ctx, lo, hi), c, 6, 0.01, global_valid_sum=vsum, global_rows=N)
        assert torch.equal(c, a[r0:r1]) and torch.equal(sc_c, sc_a[r0:r1])
    # Adam's bias-corrected step is bounded by lr * (1 - beta1) / sqrt(1 - beta2) ~ 3.2 lr (typically ~lr)
    assert torch.isfinite(a).all() and (a - p0).abs().max().item() <= 6 * 0.01 * 3.2


def test_refinement_full_size_sharded(ctx):
    """--refinement's mixing loop on all 786 432 rows (by-mode wavefronts regrouped per XCD, the rows to mix packed per scene):
    bitwise reproducible, untouched where the reference leaves a row alone, and a quarter of the scenes evaluated alone -- with
    the batch's loss scale -- reproduces its rows bit for bit, gradients included (the regrouped blocks read the tables and
    the list of the scene they stand for)."""
    sm, sb, dev = ctx["sm"], ctx["sb"], ctx["dev"]
    N, iters = sb.N, 3
    g = torch.Generator(device=dev).manual_seed(29)
    clist = torch.rand(100, N, 40, device=dev, generator=g) * 2 - 1            # 12.6 GB
    cin = torch.rand(N, 40, device=dev, generator=g) * 2 - 1
    out, tr = sm.refinement(sb, cin, clist, iters=iters, trace=True)
    out2 = sm.refinement(sb, cin, clist, iters=iters)
    assert torch.equal(out, out2) and torch.isfinite(out).all()
    sc0 = sm.score(sb, cin.reshape(1, N, 40))["scores"][0]
    keep = ~((sc0 <= 0) & (sb.valid > 0))                                       # nusc_train.py:1045-1046: these rows are copied
    assert torch.equal(out[keep], cin[keep]) and 0.05 * N < int((~keep).sum()) < N
    assert (tr[:, keep] == 0).all()
    for q in (1, 3):
        lo, hi = q * BS // 4, (q + 1) * BS // 4
        r0, r1 = lo * S * 3, hi * S * 3
        sub = _sub(ctx, lo, hi)
        sub.grad_scale = sb.grad_scale
        o, t = sm.refinement(sub, cin[r0:r1].contiguous(), clist[:, r0:r1].contiguous(), iters=iters, trace=True)
        assert torch.equal(o, out[r0:r1]) and torch.equal(t, tr[:, r0:r1]), q
    del clist


def test_dpp_loss_full_size_is_invariant_to_scene_order(ctx):
    """Reversing the order of the scenes permutes the groups: group diversities and gradients permute with them."""
    from pstl_diffusion_policy_amd import ffi
    sm, sb, out = ctx["sm"], ctx["sb"], ctx["out"]
    dev, N = ctx["dev"], sb.N
    cfg = sb.cfg(2)
    G = BS * 3 * cfg.n_shards
    rect, scores = out["final_controls"], out["final_scores"]

    def run(rc, sc):
        o = dict(div=torch.empty(G, device=dev), dc=torch.empty(N, 40, device=dev), ds=torch.empty(N, device=dev))
        ffi.check(ffi.lib().pstl_diversity_loss(ctypes.byref(cfg), ffi.ptr(rc), ffi.ptr(None), ffi.ptr(sc), ctypes.c_float(1.0),
                                                ctypes.c_float(1.0), 0, ctypes.c_float(0.0), ffi.ptr(o["div"]), ffi.ptr(None),
                                                ffi.ptr(None, torch.float64), ffi.ptr(o["dc"]), ffi.ptr(o["ds"]), ffi.stream()),
                  "diversity_loss")
        return o
    a = run(rect, scores)
    rrev = rect.reshape(BS, S * 3, 40).flip(0).reshape(N, 40).contiguous()
    srev = scores.reshape(BS, S * 3).flip(0).reshape(N).contiguous()
    b = run(rrev, srev)
    torch.cuda.synchronize()
    assert torch.equal(a["div"].reshape(BS, -1).flip(0), b["div"].reshape(BS, -1))
    assert torch.equal(a["dc"].reshape(BS, -1).flip(0), b["dc"].reshape(BS, -1))
    assert torch.equal(a["ds"].reshape(BS, -1).flip(0), b["ds"].reshape(BS, -1))
    n = S // cfg.n_shards
    assert torch.isfinite(a["div"]).all() and (a["div"] >= -1e-5).all() and (a["div"] <= n + 1e-4).all()


def test_split_bf16_chain_tracks_the_fp32_chain_full_size(ctx):
    """The default MLP-chain arithmetic (both networks: every fp32 operand as two IEEE-half pieces, three f16 MFMA products
    per fp32 product, fp32 accumulate; at this size the multi-step launches run on the row-stationary kernel k_chain2)
    against the exact fp32-MFMA variant, 786 432 rows through 99 chained reverse steps on the same in-kernel noise: every
    one of the 31 million final controls stays within 6e-5 (the typical element is at 1e-6), inside the 1e-4 parity gate (no
    guidance: the comparison is of the chain alone), and RefineNet's output likewise.  (The name is historical: round 1's
    default had bfloat16 pieces.)"""
    from pstl_diffusion_policy_amd.engine import Sampler
    sb, dev = ctx["sb"], ctx["dev"]
    outs = {}
    for cw in (8, 0):   # 0 = the default: both networks on split-f16 products
        sm = Sampler(ctx["sm"].w, ctx["hp"], chain_waves=cw)
        o = sm.sampling_region(sb, 100, None, None, rect_head=True, multi_cands=3, seed=77, want_scores3=False)
        outs[cw] = {k: o[k].clone() for k in ("final_controls", "sel_controls", "sel_idx", "sel_scores")}
    torch.cuda.synchronize()
    same = outs[8]["sel_idx"] == outs[0]["sel_idx"]
    assert same.float().mean().item() > 0.999          # a candidate pair within rounding of each other may swap
    d_sel = (outs[8]["sel_controls"] - outs[0]["sel_controls"]).abs().reshape(sb.N, -1).amax(dim=1)[same]
    assert d_sel.max().item() < 6e-5, "sampled controls, split-bf16 vs fp32 chain: %.3e" % d_sel.max().item()
    # RefineNet rewrites a row only if its STL score is negative: rows within rounding of 0 may take either branch
    # (and merge_net max-pools over the samples of a scene: a swapped candidate anywhere in the scene moves all its rows)
    scene_same = same.reshape(BS, 3 * S).all(dim=1).repeat_interleave(3 * S)
    clear = scene_same & (outs[8]["sel_scores"].abs() > 1e-3) & (outs[0]["sel_scores"].abs() > 1e-3)
    assert clear.float().mean().item() > 0.9
    d_fin = (outs[8]["final_controls"] - outs[0]["final_controls"]).abs().reshape(sb.N, -1).amax(dim=1)[clear]
    assert d_fin.max().item() < 6e-5, "refined controls, split-bf16 vs fp32 chain: %.3e" % d_fin.max().item()


def test_config5_training_step_full_size(ctx):
    """Config 5 (e8_ours_ablation: RefineNet under the STL loss) at BASELINE's single-GPU size, 786 432 rows, through
    size-independent properties -- the oracle cannot run here:
      * the loss of a batch is the mean over its rows, so the gradients of the two half batches (each scaled with the
        GLOBAL valid-row statistics, as a rank of a two-GPU run does) add up to the full batch's, to summation order;
      * the same for the loss itself; every gradient is finite and non-trivial;
      * three optimisation steps on frozen noise lower the loss."""
    from pstl_diffusion_policy_amd.engine import PackedWeights, RectTrainer, Sampler
    dev, hp, sb = ctx["dev"], ctx["hp"], ctx["sb"]
    sd = {k: torch.from_numpy(v).to(dev) for k, v in golden_weights().items()}
    params = {k: sd[k].clone().requires_grad_() for k in RectTrainer.NAMES}
    steps, mc = 10, 5

    def grads_of(sbx, seed_rows):
        sm = Sampler(PackedWeights(dict(sd, **{k: v.detach() for k, v in params.items()}), dev), hp)
        tr = RectTrainer(sm)
        feature, base_p, base_r = sm.encode(sbx, need_rect=True)
        x = sm.fill_normal(sbx, steps, steps, 21)
        emit = sm.rollout(sbx, base_p, x, None, steps, n_emit=mc, clip=True, seed=21)
        r = sm.score(sbx, emit[-mc:].contiguous(), select=True)
        loss, rect, scores, g = tr.loss_and_grads(sbx, feature, base_r, params["rect_net.2.weight"],
                                                  params["rect_net.4.weight"], r["sel_controls"], r["sel_scores"])
        return loss, g, scores

    loss_full, g_full, scores = grads_of(sb, 0)
    assert sb.N == 786432 and torch.isfinite(loss_full)
    vsum = float(sb.valid.sum().item())
    half = BS // 2
    halves = [_sub(ctx, 0, half, global_valid_sum=vsum, global_rows=sb.N, row_offset=0),
              _sub(ctx, half, BS, global_valid_sum=vsum, global_rows=sb.N, row_offset=half * S * 3)]
    parts = [grads_of(h, 0) for h in halves]
    np.testing.assert_allclose(float(parts[0][0] + parts[1][0]), float(loss_full), rtol=2e-5)
    for k in RectTrainer.NAMES:
        full = g_full[k].cpu().numpy()
        two = (parts[0][1][k] + parts[1][1][k]).cpu().numpy()
        assert np.isfinite(full).all() and np.abs(full).max() > 0, k
        np.testing.assert_allclose(two, full, rtol=2e-3, atol=2e-5 * np.abs(full).max(), err_msg=k)
    # the rows of the two halves are the rows of the full batch (in-kernel noise is keyed by the global row)
    assert torch.equal(torch.cat([parts[0][2], parts[1][2]]), scores)
    opt = torch.optim.Adam([params[k] for k in RectTrainer.NAMES], lr=3e-4)
    losses = []
    for it in range(3):
        sm = Sampler(PackedWeights(dict(sd, **{k: v.detach() for k, v in params.items()}), dev), hp)
        loss, _ = RectTrainer(sm).train_step(sb, params, opt, steps, seed=21, multi_cands=mc)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    print("config 5 at 786432 rows: loss over three Adam steps", losses)
