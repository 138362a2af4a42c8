"""Host-side mirror of the reference's sampling surface (reference: nusc_train.py), calling libpstl_hip.so.

Same function names, argument meaning and return structure as the reference for the hot path:
  generate_parser (:1635) . get_diffusion_coeffs (:528) . dup (:20) . mask_mean (:23) . generate_trajs (:39) .
  build_stl_cache (:95) . augment_batch_data (:724) . pre_prepare_stl_cache (:258) . compute_stl_dense (:318) .
  diffusion_rollout (:557) . normalize_diff (:647) . run_sampling_test (:890) . main (:1185)
and the same CLI flags (--diffusion --guidance* --run_sampling_test --multi_cands --rect_head --no_refinenet
--diffusion_steps --sampling_size --n_rolls --time_profile ...), including the reference's post-parse overrides.

Differences that are deliberate:
  * batches carry a scene-indexed side channel (`batch["_pstl"]`, an engine.SceneBatch) so that nothing is replicated
    per row; the reference's dense keys (`neighbors_dense`, ...) are only materialised with `dense=True`.
    `compute_stl_dense` also accepts a purely dense `stl_input` (rows_per_scene = 1) as the reference passes it.
  * data comes from the seeded synthetic scene generator (synthetic.py): the nuScenes cache / devkit are out of scope.
  * the metrics the reference computes on the CPU after its timer stops (diversity std / hull volume / entropies /
    occupancy area, ADE/FDE; nusc_train.py:1107-1140) come from one HIP kernel (pstl_diversity), for the sampled
    trajectories and for the dataset's stored traj-opt solutions (the "TJ" columns, :918-957).
"""
import argparse
import ctypes
import time

import numpy as np
import torch

from . import ffi
from .engine import Sampler, SceneBatch, acc_from_counts, diffusion_coeffs, diversity_from_totals
from .nusc_model import Net
from .stl_d_lib import AP, Always, And, Eventually, ListAnd
from .synthetic import make_scene_batch


# ---------------------------------------------------------------------------------------------------------------
# small helpers with the reference's names
# ---------------------------------------------------------------------------------------------------------------
def dup(x, m):
    """(N, ...) -> (N*m, ...), each row repeated m times contiguously."""
    return x.unsqueeze(1).repeat((1, m) + tuple(1 for _ in x.shape[1:])).reshape((-1,) + x.shape[1:])


def mask_mean(loss, mask, dim=None):
    if dim is not None:
        return torch.mean(loss * mask, dim=dim) / torch.clip(torch.mean(mask, dim=dim), 1e-2)
    return torch.mean(loss * mask) / torch.clip(torch.mean(mask), 1e-2)


def get_diffusion_coeffs(args):
    """(beta, alpha, alpha_hat) on the GPU; cosine schedule (the reference forces --cos on, nusc_train.py:1782)."""
    if not getattr(args, "cos", True):
        raise NotImplementedError("the reference forces the cosine schedule (nusc_train.py:1782)")
    return diffusion_coeffs(args.diffusion_steps, torch.device("cuda"))


def normalize_diff(x, n, nt, w_max, a_max, clip):
    x = x.reshape(n, nt, 2)
    w = x[..., 0] * w_max
    a = x[..., 1] * a_max
    if clip:
        w = torch.clip(w, -w_max, w_max)
        a = torch.clip(a, -a_max, a_max)
    return torch.stack([w, a], dim=-1)


def _hp(args):
    return dict(nt=args.nt, dt=args.dt, mul_w_max=args.mul_w_max, mul_a_max=args.mul_a_max,
                smoothing_factor=args.smoothing_factor, stl_nn_thres=args.stl_nn_thres, ego_L=args.ego_L,
                ego_W=args.ego_W, refined_nL=args.refined_nL, refined_nW=args.refined_nW, n_segs=args.n_segs,
                n_shards=args.n_shards, norm_stl=bool(getattr(args, "norm_stl", False)))


def generate_trajs(s, us, dt):
    """(..., 4) x (..., T, 2) -> (..., T+1, 4): unicycle Euler rollout on the GPU (one row per lane)."""
    assert s.shape[-1] == 4 and us.shape[-1] == 2 and us.shape[:-2] == s.shape[:-1]
    assert us.shape[-2] == ffi.T, "libpstl_hip is specialised for nt = 20"
    lead = us.shape[:-2]
    dev = us.device
    s0 = ffi.f32(s.reshape(-1, 4), dev)
    u = ffi.f32(us.reshape(-1, ffi.CTRL), dev)
    R = u.shape[0]
    out = torch.empty(R, ffi.T + 1, 4, dtype=torch.float32, device=dev)
    hp = dict(smoothing_factor=1.0, stl_nn_thres=0.0, mul_w_max=1.0, mul_a_max=1.0, dt=float(dt), ego_L=1.0, ego_W=1.0)
    cfg = ffi.make_cfg(R, 1, 1, 1, 2, hp)
    ffi.check(ffi.lib().pstl_generate_trajs(ctypes.byref(cfg), ffi.ptr(s0), ffi.ptr(u), ffi.ptr(out), ffi.stream()),
              "generate_trajs")
    return out.reshape(lead + (ffi.T + 1, 4))


# indices into stlp (reference nusc_train.py:52-57) and the state vector
I_VMIN, I_VMAX, I_DMIN, I_DMAX, I_DSAFE, I_THMAX = range(6)
I_V = 3


def build_stl_cache(args):
    """The three lane-mode formulas [keep lane, change left, change right] of the path (reference nusc_train.py:95-140)
    as callable stl_d_lib objects: `stls[m](x, tau)[:, 0]` is the robustness of rows under formula m, with x the signal
    dict prep_stl_cache fills (ego_traj, stlp, x2{curr,left,right}_{d,th}, min_nei_d [, *_factor with --norm_stl]).
    compute_stl_dense itself evaluates them through the fused kernel (pstl_stl_forward), which never materialises x."""
    nt = args.nt
    norm = bool(getattr(args, "norm_stl", False))

    def over(expr, factor):      # --norm_stl divides every metric predicate by its factor (nusc_train.py:97-113)
        return (lambda x: expr(x) / x[factor]) if norm else expr

    def at_least(sig, idx, factor):
        return AP(over(lambda x: sig(x) - x["stlp"][..., idx], factor))

    def at_most(sig, idx, factor):
        return AP(over(lambda x: -sig(x) + x["stlp"][..., idx], factor))

    def heading_ok(key):
        return AP(lambda x: (x["stlp"][..., I_THMAX] - x[key]) / x["stlp"][..., I_THMAX])

    speed = lambda x: x["ego_traj"][..., I_V]
    field = lambda key: (lambda x: x[key])
    keep_speed = [Always(0, nt, at_least(speed, I_VMIN, "v_factor")), Always(0, nt, at_most(speed, I_VMAX, "v_factor"))]
    keep_lane = [Always(0, nt, at_least(field("x2curr_d"), I_DMIN, "d_factor")),
                 Always(0, nt, at_most(field("x2curr_d"), I_DMAX, "d_factor")), Always(0, nt, heading_ok("x2curr_th"))]
    safe = [Always(0, nt, at_least(field("min_nei_d"), I_DSAFE, "safe_factor"))]

    def reach(side):
        band = And(at_least(field("x2%s_d" % side), I_DMIN, "d_factor"), at_most(field("x2%s_d" % side), I_DMAX, "d_factor"))
        return [Eventually(0, nt // 2, Always(0, nt, band)), Eventually(0, nt // 2, Always(0, nt, heading_ok("x2%s_th" % side)))]

    return [ListAnd(keep_speed + keep_lane + safe), ListAnd(keep_speed + reach("left") + safe),
            ListAnd(keep_speed + reach("right") + safe)]


def get_stl_scores(scores_list, stl_i):
    """Mode select (reference nusc_train.py:150-151)."""
    return sum(scores_list[m] * (stl_i == m).float() for m in range(4))


# ---------------------------------------------------------------------------------------------------------------
# batch construction
# ---------------------------------------------------------------------------------------------------------------
def get_dense_stlp(batch_cuda, the_stlp, args, n_randoms=None):
    """Per-row STL parameters of the traj-opt / data-generation pass (reference nusc_train.py:657-722): the mode that
    matches the scene's ground-truth label keeps the GT parameters `the_stlp`; the other modes get the fixed prior
    (0, 20, -2.5, 2.5, 0.1, 0.5), or with --flex parameters drawn around the GT ones (torch's CPU generator, in the
    reference's order of draws, so a seeded run reproduces the reference's parameters).  -> (bs*n_randoms*3, 1, 6)."""
    bs = the_stlp.shape[0]
    if n_randoms is None:
        n_randoms = args.n_randoms
    dev = the_stlp.device
    hl = batch_cuda["gt_high_level"].reshape(bs, 1, 1)
    mid = the_stlp.unsqueeze(1).repeat(1, n_randoms, 1)                       # (bs, n_randoms, 6)
    U = lambda a, b: (torch.rand(bs, 1) * (b - a) + a).repeat(1, n_randoms).to(dev)

    def flex_params(level):
        vd0, vd1 = U(1.3, 3), U(1.3, 3)
        vmin = torch.clip(mid[:, :, 0] - vd0, -0.3)
        vmax = torch.clip(mid[:, :, 1] + vd1, -0.3)
        if level == 0:
            l0, l1 = U(0, 1), U(0, 1)
            dmin = l0 * mid[:, :, 2] + (1 - l0) * (mid[:, :, 2] - 2.5)
            dmax = l1 * mid[:, :, 2] + (1 - l1) * (mid[:, :, 2] + 2.5)          # (the reference blends dmin here too)
        else:
            dmin, dmax = U(-2.5, -0.5), U(0.5, 2.5)
        l2 = U(0, 1)
        dsafe = torch.clip(l2 * mid[:, :, 4] + (1 - l2) * (mid[:, :, 4] - 1.5), 0)
        l3 = U(0, 1)
        thmax = l3 * mid[:, :, 5] + (1 - l3) * (mid[:, :, 5] + 0.3)
        return torch.stack([vmin, vmax, dmin, dmax, dsafe, thmax], dim=-1)

    if args.flex:
        d0, d1, d2 = flex_params(0), flex_params(1), flex_params(2)
        keep0 = (hl * (3 - hl) == 0).float()                                   # labels 0 and 3 keep the GT set in mode 0
        per_mode = [keep0 * mid + (1 - keep0) * d0, (hl == 1).float() * mid + (hl != 1).float() * d1,
                    (hl == 2).float() * mid + (hl != 2).float() * d2]
    else:
        prior = torch.tensor([0.0, 20.0, -2.5, 2.5, 0.1, 0.5], device=dev).reshape(1, 1, 6).repeat(bs, n_randoms, 1)
        per_mode = [(hl == m).float() * mid + (hl != m).float() * prior for m in range(3)]
    return torch.stack(per_mode, dim=-2).reshape(bs * n_randoms * 3, 1, 6)


def augment_batch_data(batch, the_stlp, args, n_randoms=None, stlp_dense=None, dense=False):
    """Adds the per-row constants of the sampling harness (reference nusc_train.py:724-754).  `batch["_pstl"]` holds
    the scene-indexed device tensors the kernels read; dense replicas are only built with dense=True."""
    if n_randoms is None:
        new_sample = False
        n_randoms = args.n_randoms
    else:
        new_sample = True
    m = n_randoms * 3
    bs = batch["currlane_wpts"].shape[0]
    dev = batch["currlane_wpts"].device
    batch["stlp"] = the_stlp.unsqueeze(-2)
    if stlp_dense is not None:
        batch["stlp_dense"] = stlp_dense
    elif args.load_stlp:
        if new_sample:
            batch["stlp_dense"] = batch["pre_stlp"].reshape(bs, -1, 3, 6)[:, 0:1].repeat(1, n_randoms, 1, 1).reshape(bs * m, 1, 6)
        else:
            batch["stlp_dense"] = batch["pre_stlp"].reshape(bs * m, 1, 6)
    else:
        batch["stlp_dense"] = get_dense_stlp(batch, the_stlp, args, n_randoms=n_randoms)
    valids = torch.cat([batch["curr_id"], batch["left_id"], batch["right_id"]], dim=-1)
    batch["valids_dense"] = dup(valids, n_randoms).reshape(bs * n_randoms, 3)
    batch["highlevel_dense"] = torch.tensor([0, 1.0, 2.0], device=dev).reshape(1, 3, 1).repeat(bs * n_randoms, 1, 1).reshape(bs * m, 1)
    scene = {k: batch[k] for k in ("ego_traj", "neighbors", "currlane_wpts", "leftlane_wpts", "rightlane_wpts",
                                   "curr_id", "left_id", "right_id")}
    scene["neighbors_traj"] = batch["neighbor_trajs_aug"]
    scene["stlp_rows"] = batch["stlp_dense"][:, 0]
    batch["_pstl"] = SceneBatch(scene, n_randoms, _hp(args), dev)
    if dense:
        batch["neighbors_dense"] = dup(batch["neighbor_trajs_aug"], m)
        for k in ("curr", "left", "right"):
            batch["%slane_wpts_dense" % k] = dup(batch["%slane_wpts" % k], m)
    return batch


def pre_prepare_stl_cache(batch_cuda, dense_trajs=None, detach=False, repeat_n=None, mono=False, mono_n=None, gt_stlp=None):
    if mono:
        raise NotImplementedError("mono (gt_data_training) rows are outside the hot path")
    stl_input = {"stlp": batch_cuda["stlp_dense"], "dense_valids": batch_cuda["valids_dense"],
                 "gt_high_level": batch_cuda["gt_high_level"]}
    if repeat_n is not None:
        stl_input = {k: v.repeat(repeat_n, *[1] * (v.dim() - 1)) for k, v in stl_input.items()}
    for k in ("neighbors_dense", "currlane_wpts_dense", "leftlane_wpts_dense", "rightlane_wpts_dense"):
        if k in batch_cuda:
            v = batch_cuda[k]
            stl_input[k.replace("_dense", "")] = v if repeat_n is None else v.repeat(repeat_n, *[1] * (v.dim() - 1))
    stl_input["_pstl"] = batch_cuda.get("_pstl")
    stl_input["_pstl_reps"] = repeat_n or 1
    if dense_trajs is not None:
        stl_input["ego_traj"] = dense_trajs.detach() if detach else dense_trajs
    return stl_input


def _scene_tables(stl_input, R, dev, hp):
    """The prepared scene tables the STL kernels read, from either layout a caller may hand over: the scene-indexed side
    channel `_pstl` (nothing replicated per row) or the reference's dense per-row tensors (rows_per_scene = 1).
    Returns (cfg, s0, nei_prep, lane_prep, stlp, reps, n_per)."""
    sb = stl_input.get("_pstl")
    reps = stl_input.get("_pstl_reps", 1)
    if sb is not None and sb.N * reps == R:
        # the STL parameters are the ones the caller hands over NOW (stl_input["stlp"], reference nusc_train.py:266), not the
        # ones captured when the batch was augmented: the closed-loop caller overwrites them in between (nusc_sim.py:442-472)
        stlp = sb.stlp
        if stl_input.get("stlp") is not None and stl_input["stlp"].shape[0] == R:
            stlp = ffi.f32(stl_input["stlp"].reshape(reps, sb.N, 6)[0], dev)
        return sb.cfg(2), sb.s0, sb.nei_prep, sb.lane_prep, stlp, reps, sb.N
    nei = ffi.f32(stl_input["neighbors"][..., :7], dev)
    lanes = [ffi.f32(stl_input["%slane_wpts" % k], dev) for k in ("curr", "left", "right")]
    K = nei.shape[1]
    cfg = ffi.make_cfg(R, 1, 1, K, 2, hp)
    nei_prep = torch.empty(R, K, ffi.T, ffi.NEI_PREP, dtype=torch.float32, device=dev)
    lane_prep = torch.empty(R, 3, ffi.NSEG, 4, dtype=torch.float32, device=dev)
    ffi.check(ffi.lib().pstl_prepare_scene(ctypes.byref(cfg), ffi.ptr(nei), ffi.ptr(lanes[0]), ffi.ptr(lanes[1]),
                                           ffi.ptr(lanes[2]), ffi.ptr(nei_prep), ffi.ptr(lane_prep), ffi.stream()),
              "prepare_scene")
    return cfg, None, nei_prep, lane_prep, ffi.f32(stl_input["stlp"].reshape(R, 6), dev), 1, R


SIGNAL_KEYS = ("x2curr_d", "x2curr_th", "x2left_d", "x2left_th", "x2right_d", "x2right_th", "min_nei_d")


def prep_stl_cache(x, args):
    """Adds the seven signals of the formulas to x IN PLACE, as the reference does (nusc_train.py:74-93): x2*_d / x2*_th
    (signed lateral distance and heading error to the three lanes) and min_nei_d (clearance to the closest neighbour),
    each (R,T) for x["ego_traj"] (R,T,>=4); with --norm_stl also the three normalisation factors (:88-91)."""
    traj = x["ego_traj"]
    dev = traj.device
    R = traj.shape[0]
    cfg, _, nei_prep, lane_prep, stlp, reps, n_per = _scene_tables(x, R, dev, _hp(args))
    states = ffi.f32(traj[..., :4], dev).reshape(R, ffi.T, 4)
    sig = torch.empty(7, reps, n_per, ffi.T, dtype=torch.float32, device=dev)
    for r in range(reps):   # the scene tables repeat per candidate block (repeat_n of pre_prepare_stl_cache)
        out_r = torch.empty(7, n_per, ffi.T, dtype=torch.float32, device=dev)
        st_r = states.reshape(reps, n_per, ffi.T, 4)[r].contiguous()
        ffi.check(ffi.lib().pstl_stl_signals(ctypes.byref(cfg), ffi.ptr(None), ffi.ptr(None), ffi.ptr(st_r), ffi.ptr(nei_prep),
                                             ffi.ptr(lane_prep), ffi.ptr(out_r), ffi.stream()), "stl_signals")
        sig[:, r] = out_r
    for i, k in enumerate(SIGNAL_KEYS):
        x[k] = sig[i].reshape(R, ffi.T)
    if "stlp" not in x or x["stlp"].shape[0] != R:
        x["stlp"] = stlp.reshape(n_per, 1, 6).repeat(reps, 1, 1)
    if getattr(args, "norm_stl", False):
        p = x["stlp"]
        x["v_factor"] = torch.clip(p[..., I_VMAX] - p[..., I_VMIN], 0.3)
        x["d_factor"] = torch.clip((p[..., I_DMAX] - p[..., I_DMIN]) * 5, 0.3)
        x["safe_factor"] = torch.clip(p[..., I_DSAFE], 0.3)
    return x


def infer_gt_stlp(batch_cuda, gt_trajs, args, data_loader=None):
    """STL parameters (vmin, vmax, dmin, dmax, dsafe, thmax) that the ground-truth future itself satisfies (reference
    nusc_train.py:210-251): extrema of the GT speed, of its lane distance / heading error w.r.t. the lane of its
    high-level label (side lanes: from step nt/2 - 1 on) and of its clearance, widened by the --flex margins.  The signals
    come from pstl_stl_signals (dense layout: one row per scene)."""
    bs = gt_trajs.shape[0]
    x = {"ego_traj": gt_trajs[..., :4], "neighbors": batch_cuda["neighbor_trajs_aug"],
         "currlane_wpts": batch_cuda["currlane_wpts"], "leftlane_wpts": batch_cuda["leftlane_wpts"],
         "rightlane_wpts": batch_cuda["rightlane_wpts"], "stlp": torch.zeros(bs, 1, 6, device=gt_trajs.device)}
    x = prep_stl_cache(x, argparse.Namespace(**dict(vars(args), norm_stl=False)))
    v = gt_trajs[..., 3]
    hl = batch_cuda["gt_high_level"][:, 0]
    h = args.nt // 2 - 1
    sel = lambda a0, a1, a2, dflt: (a0 * (hl == 0).float() + a1 * (hl == 1).float() + a2 * (hl == 2).float()
                                    + dflt * (hl == 3).float())
    mn = lambda t: torch.min(t, dim=-1)[0]
    mx = lambda t: torch.max(t, dim=-1)[0]
    dmin = sel(mn(x["x2curr_d"]), mn(x["x2left_d"][:, h:]), mn(x["x2right_d"][:, h:]), -5)
    dmax = sel(mx(x["x2curr_d"]), mx(x["x2left_d"][:, h:]), mx(x["x2right_d"][:, h:]), 5)
    thmax = sel(mx(x["x2curr_th"]), mx(x["x2left_th"][:, h:]), mx(x["x2right_th"][:, h:]), 0.5)
    dsafe = mn(x["min_nei_d"])
    if args.flex:
        return torch.stack([torch.clip(mn(v) - 1, -0.3), mx(v) + 1, dmin - 0.3, dmax + 0.3, torch.clip(dsafe - 0.1, 0),
                            thmax + 0.1], dim=-1)
    return torch.stack([mn(v) - 0.1, mx(v) + 0.1, dmin - 0.1, dmax + 0.1, dsafe - 0.1, thmax + 0.05], dim=-1)


def compute_stl_dense(stl_input, stls_cac, stl_idx, mask, args, debug=False, tj_scores=None, scene=False):
    """Scores trajectories stl_input["ego_traj"] (R,T,4) (reference nusc_train.py:318-345).
    Returns (scores_list [curr, left, right, ones], scores (R,), acc[, scene_acc]).
    The fused kernel (pstl_stl_forward) evaluates the three fixed formulas without materialising any signal, --norm_stl
    included (PSTL_FLAG_NORM_STL).  stls_cac == "generic" (or objects other than build_stl_cache's): the signals are prepared
    (prep_stl_cache) and the formula objects are evaluated by the generic program kernel, as the reference's code path
    reads (:319-323)."""
    traj = stl_input["ego_traj"]
    dev = traj.device
    R = traj.shape[0]
    hp = _hp(args)
    if getattr(args, "generic_stl", False):
        x = prep_stl_cache(stl_input, args)
        scores_list = [stl(x, args.smoothing_factor)[:, 0] for stl in stls_cac]
        scores_list.append(torch.ones_like(scores_list[0]))
        scores = get_stl_scores(scores_list, stl_idx.reshape(R).to(dev))
        return _stl_metrics_tail(stl_input, scores_list, scores, mask, args, debug, tj_scores, scene)
    states = ffi.f32(traj[..., :4], dev).reshape(R, ffi.T, 4)
    cfg, s0, nei_prep, lane_prep, stlp, reps, n_per = _scene_tables(stl_input, R, dev, hp)
    hl = ffi.f32(stl_idx.reshape(reps, n_per)[0], dev)
    states = states.reshape(reps, n_per, ffi.T, 4)
    scores = torch.empty(reps, n_per, dtype=torch.float32, device=dev)
    s3 = torch.empty(3, reps, n_per, dtype=torch.float32, device=dev)
    ffi.check(ffi.lib().pstl_stl_forward(ctypes.byref(cfg), ffi.ptr(s0), ffi.ptr(None), ffi.ptr(states), int(reps),
                                         ffi.ptr(nei_prep), ffi.ptr(lane_prep), ffi.ptr(stlp), ffi.ptr(hl),
                                         ffi.ptr(scores), ffi.ptr(s3), ffi.ptr(None), ffi.ptr(None),
                                         ffi.ptr(None, torch.int32), ffi.stream()), "stl_forward")
    scores = scores.reshape(R)
    scores_list = [s3[0].reshape(R), s3[1].reshape(R), s3[2].reshape(R)]
    scores_list.append(torch.ones_like(scores))
    return _stl_metrics_tail(stl_input, scores_list, scores, mask, args, debug, tj_scores, scene)


def _stl_metrics_tail(stl_input, scores_list, scores, mask, args, debug, tj_scores, scene):
    mask_flat = mask.reshape(-1).to(scores.dtype)
    if getattr(args, "oracle_filter", False) and tj_scores is not None:
        cube = torch.max(tj_scores.reshape(-1, args.n_randoms, 3), dim=1, keepdim=True)[0]
        mask_flat = mask_flat * (cube > 0).float().repeat(1, args.n_randoms, 1).reshape(-1)
    acc = mask_mean((scores > 0).float(), mask_flat)
    if debug:
        return scores_list, scores, acc, stl_input
    if scene:
        cube = scores.reshape(-1, args.n_randoms, 3)
        mcube = mask.reshape(-1, args.n_randoms, 3)
        scene_acc = mask_mean((torch.max(cube, dim=1)[0] > 0).float(), mcube[:, 0, :])
        return scores_list, scores, acc, scene_acc
    return scores_list, scores, acc


# ---------------------------------------------------------------------------------------------------------------
# reverse diffusion
# ---------------------------------------------------------------------------------------------------------------
def _scene_batch_of(batch_cuda, n, n_randoms, args, dev, guidance_extras=None):
    """The scene-indexed tables of a batch for the rollout / guidance kernels.  `batch_cuda["_pstl"]` (written by this
    package's augment_batch_data) when it is there and fits; otherwise built from what the reference's own callers hold in
    the dict: the scene-level tensors Net.forward reads (ego_traj, neighbors, *lane_wpts, *_id) and the neighbour futures
    (`neighbor_trajs_aug`, or one row per scene of the row-replicated `neighbors_dense`)."""
    sb = batch_cuda.get("_pstl")
    if sb is not None and sb.N == n:
        return sb
    m = n_randoms * 3
    src = batch_cuda
    if "neighbor_trajs_aug" not in src and "neighbors_dense" not in src and guidance_extras is not None:
        src = dict(batch_cuda, **{k: v for k, v in guidance_extras[0].items() if k in ("neighbor_trajs_aug", "neighbors_dense")})
    scene = {k: src[k] for k in ("ego_traj", "neighbors", "currlane_wpts", "leftlane_wpts", "rightlane_wpts",
                                 "curr_id", "left_id", "right_id")}
    if "neighbor_trajs_aug" in src:
        scene["neighbors_traj"] = src["neighbor_trajs_aug"]
    elif "neighbors_dense" in src:
        scene["neighbors_traj"] = src["neighbors_dense"][::m]
    else:
        raise KeyError("diffusion_rollout needs the neighbours' futures: batch['neighbor_trajs_aug'] (bs,K,nt,7) or the "
                       "row-replicated batch['neighbors_dense']")
    scene["stlp_rows"] = batch_cuda["stlp_dense"].reshape(n, 6)
    sb = SceneBatch(scene, n_randoms, _hp(args), dev)
    if sb.N != n:
        raise ValueError("batch holds %d scenes x %d samples x 3 modes = %d rows, the noise has %d" % (sb.bs, n_randoms, sb.N, n))
    if guidance_extras is not None:     # the states the guidance block rolls out from (reference :612): one row per scene
        sb.s0 = ffi.f32(guidance_extras[1].reshape(n, -1)[::m, :4], dev)
    return sb


def diffusion_rollout(noise, net, batch_cuda, highlevel_dense, feature, args, coeffs=None, fastforward=False,
                      n_randoms=None, return_feature=False, mono=False, tmp_stlp=None, guidance_extras=None,
                      maximize=False):
    """Reference nusc_train.py:557-645.  `noise` only provides the shape (as in the reference); x_T and the per-step
    noise are drawn from torch's global generator in the reference's order."""
    if mono:
        raise NotImplementedError("mono (gt_data_training) rows are outside the hot path")
    n = noise.shape[0]
    net.eval()
    steps = args.diffusion_steps
    dev = noise.device
    if n_randoms is None:
        n_randoms = args.n_randoms
    if feature is None:
        feature = net.scene_feature(batch_cuda, n_randoms * 3)
    sb = _scene_batch_of(batch_cuda, n, n_randoms, args, dev, guidance_extras)
    # Per-row constants as they are at CALL time, as the reference reads them (ext["stlp"] = batch_cuda["stlp_dense"], :574;
    # highlevel_dense is an argument): its closed-loop caller overwrites stlp_dense between augment_batch_data and this call
    # (nusc_sim.py:442-472) -- in place or by assigning a new tensor, both must count.
    sb.stlp = ffi.f32(batch_cuda["stlp_dense"].reshape(n, 6), dev)
    sb.hl = ffi.f32(highlevel_dense.reshape(n), dev)
    with torch.no_grad():
        kernel_noise = bool(getattr(args, "kernel_noise", False)) and not fastforward
        seed = None
        if kernel_noise:     # one draw from torch's generator keys the in-kernel streams (x_T and every step's noise)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            x = Sampler(net.packed(), net.hparams()).fill_normal(sb, steps, steps, seed)
            zs = None
        else:
            x = torch.randn_like(noise).float().contiguous()
            zs = torch.empty(max(steps - 1, 1), n, ffi.CTRL, dtype=torch.float32, device=dev)
            if not fastforward:
                for k, i in enumerate(reversed(range(1, steps))):
                    if i > 1:
                        zs[k] = torch.randn_like(x)
        guidance = None
        if args.guidance and guidance_extras is not None:
            guidance = dict(enabled=True, before=args.guidance_before, niters=args.guidance_niters,
                            lr=args.guidance_lr, reverse=args.guidance_reverse, sets=args.guidance_sets,
                            freq=args.guidance_freq, maximize=maximize)
        n_emit = steps if args.diff_full else 1
        if fastforward:
            emit = normalize_diff(x, n, args.nt, args.mul_w_max, args.mul_a_max, args.diffusion_clip).reshape(1, n, -1)
        else:
            # domain guard of the split-f16 chain (Net.domain_check): in eager mode x_T is kept, so that a rollout that set the
            # flag is repeated on the exact-fp32 kernels from the same x_T and the same noise (supplied, or the same Philox seed)
            eager = net.domain_check == "eager" and net.chain_arith() in (0, 16, 2)
            x_T = x.clone() if eager else None
            sm = Sampler(net.packed(), net.hparams(), chain_waves=net.chain_arith())
            emit = sm.rollout(sb, feature.pstl["base_policy"], x, zs, steps, n_emit=n_emit, clip=args.diffusion_clip,
                              guidance=guidance, coeffs=coeffs, seed=seed)
            if eager and net.check_domain():
                sm = Sampler(net.packed(), net.hparams(), chain_waves=net.chain_arith())
                emit = sm.rollout(sb, feature.pstl["base_policy"], x_T, zs, steps, n_emit=n_emit, clip=args.diffusion_clip,
                                  guidance=guidance, coeffs=coeffs, seed=seed)
    diffused_result = emit[-1].reshape(n, args.nt, 2)
    if args.diff_full:
        final_list = [e.reshape(n, args.nt, 2) for e in emit]
        return (diffused_result, feature, final_list) if return_feature else (diffused_result, final_list)
    return (diffused_result, feature) if return_feature else diffused_result


# ---------------------------------------------------------------------------------------------------------------
# the sampling harness
# ---------------------------------------------------------------------------------------------------------------
class MyTimer:
    """Named wall-clock marks (reference utils.py:112-147), with a device sync so that GPU time is attributed."""
    def __init__(self):
        self.marks, self.acc, self.count = [], {}, 0

    def add(self, key):
        torch.cuda.synchronize()
        now = time.time()
        if self.marks:
            name = "%s-%s" % (key, self.marks[-1][0])
            self.acc[name] = self.acc.get(name, 0.0) + now - self.marks[-1][1]
        self.marks.append((key, now))

    def next_batch(self):
        self.marks = []
        self.count += 1

    def print_profile(self):
        print(" ".join("%s:%.3f" % (k, v / max(self.count, 1)) for k, v in self.acc.items()))


class MeterDict:
    def __init__(self):
        self.sum, self.cnt, self.last, self.hist = {}, {}, {}, {}

    def update(self, k, v):
        self.sum[k] = self.sum.get(k, 0.0) + v
        self.cnt[k] = self.cnt.get(k, 0) + 1
        self.last[k] = v
        self.hist.setdefault(k, []).append(v)      # every value, in order (tools/paper_metric.py reads md.hist["time"])

    def __getitem__(self, k):
        return self.last.get(k, float("nan"))

    def __call__(self, k):
        return self.sum[k] / self.cnt[k] if k in self.sum else float("nan")


class SyntheticLoader:
    """Stand-in for the nuScenes DataLoader: seeded synthetic scenes with the dataset's output schema."""
    def __init__(self, args, n_batches=None):
        self.args = args
        # the reference walks its whole validation loader (--n_trials 100 batches by default); the synthetic stand-in stops
        # after 3 batches unless --n_trials asks for a specific number
        self.n_batches = n_batches if n_batches is not None else (min(args.n_trials + 1, 3) if args.n_trials >= 100
                                                                  else args.n_trials + 1)

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        a = self.args
        for bi in range(self.n_batches):
            yield make_scene_batch(a.batch_size, K=a.n_neighbors, nt=a.nt, n_segs=a.n_segs, S=a.n_randoms,
                                   seed=a.seed + bi, dt=a.dt, invalid_lane_frac=0.2, stlp_mode="wide")


def dict_to_cuda(batch):
    return {k: (v.cuda() if hasattr(v, "device") else v) for k, v in batch.items()}


class _GraphRegion:
    """The timed region of run_sampling_test as ONE HIP-graph replay per batch (--kernel_noise): at the
    reference's own batch size -- 128 scenes = 24 576 rows -- a third of the eager region's wall time is Python between ~75
    launches of 5-400 us.  The scene tensors of a batch are copied into static buffers, the noise seed and the guidance-loss
    scale travel through a pstl_dyn block (engine.DynBlock), and the captured launch sequence -- scene tables, scene encoder,
    rollout with guidance, candidate scoring, RefineNet, final scoring, counters -- is replayed.  Same kernels, same arguments,
    same order as the eager calls above: identical results (tests/test_gpu_reference_surface.py)."""
    KEYS = ("ego_traj", "neighbors", "neighbors_traj", "currlane_wpts", "leftlane_wpts", "rightlane_wpts", "curr_id", "left_id",
            "right_id", "stlp_rows")

    def __init__(self, net, args, coeffs, scene, S):
        from .engine import DynBlock, GraphCapture
        dev = scene["ego_traj"].device
        self.static = {k: scene[k].to(torch.float32).contiguous().clone() for k in self.KEYS}
        self.dyn = DynBlock(dev)
        self.dyn.set(0, 1.0)
        self.sm = Sampler(net.packed(), net.hparams(), chain_waves=net.chain_arith())
        hp, steps = _hp(args), args.diffusion_steps
        guidance = None
        if args.guidance:
            guidance = dict(enabled=True, before=args.guidance_before, niters=args.guidance_niters, lr=args.guidance_lr,
                            reverse=args.guidance_reverse, sets=args.guidance_sets, freq=args.guidance_freq)
        kw = dict(rect_head=bool(args.rect_head), multi_cands=args.multi_cands if args.rect_head else None, guidance=guidance,
                  n_rolls=args.n_rolls, refinenet=not args.no_refinenet, diverse=bool(args.diverse_loss and not args.no_arch),
                  clip_rect=bool(args.clip_rect), use_rect=not args.not_use_rect, coeffs=coeffs, want_scores3=False)

        def body():
            sb = SceneBatch(self.static, S, hp, dev, global_valid_sum=1.0, dyn=self.dyn.dev, scale_in_dyn=True)
            o = self.sm.sampling_region(sb, steps, None, None, seed=0, **kw)
            return o["final_controls"], o["final_scores"], o["counts"]

        self.graph = GraphCapture(body)

    @staticmethod
    def supported(args):
        return (bool(getattr(args, "kernel_noise", False)) and not args.refinement and not args.time_profile
                and not getattr(args, "no_graph", False))

    def run(self, scene, seed, S, ids_sum):
        """ids_sum: sum of the three lane-id columns of the batch, taken from its HOST copy (no device synchronisation here)."""
        torch._foreach_copy_([self.static[k] for k in self.KEYS], [scene[k].to(torch.float32) for k in self.KEYS])   # one launch
        N = scene["ego_traj"].shape[0] * S * 3
        self.dyn.set(seed, SceneBatch.loss_scale(float(ids_sum) * S, N))
        return self.graph.replay()


def run_sampling_test(stls_cac, data_loader, net, coeffs, args, result_queue=None, thread_nusc=None):
    """Reference nusc_train.py:890-1183, the neural-sampling half (the traj-opt 'TJ' reference numbers need the
    dataset's traj-opt parameters and are printed as nan)."""
    md = MeterDict()
    myt = MyTimer() if args.time_profile else None
    for bi, batch in enumerate(data_loader):
        if bi > args.n_trials:
            continue
        batch_cuda = dict_to_cuda(batch)
        gt_trajs = batch_cuda["ego_traj"][..., :4]
        states = gt_trajs[..., 0, :4]
        bs = states.shape[0]
        batch_cuda["neighbor_trajs_aug"] = batch_cuda["neighbors_traj"][..., :7]
        gt_stlp = infer_gt_stlp(batch_cuda, gt_trajs, args)
        N = bs * args.sampling_size * 3
        # the dataset's traj-opt solutions ("TJ" columns of the printed line; reference nusc_train.py:918-957)
        if "params" in batch_cuda:
            tj_batch = augment_batch_data({k: batch_cuda[k] for k in batch_cuda if not k.startswith("_")}, gt_stlp, args)
            tsb = tj_batch["_pstl"]
            tj_controls = batch_cuda["params"].reshape(tsb.N, -1).float().contiguous()
            tsm = Sampler(net.packed(), net.hparams(), chain_waves=net.chain_arith())
            tj_scores = tsm.score(tsb, tj_controls.reshape(1, tsb.N, -1))["scores"][0]
            tcounts, _ = tsm.metrics(tsb, tj_scores)
            tacc, tsacc = acc_from_counts(tcounts)
            md.update("tj_acc", tacc)
            md.update("tj_scene_acc", tsacc)
            _, _, tj_tot = tsm.diversity(tsb, tj_controls, tj_scores)
            for k, v in diversity_from_totals(tj_tot).items():
                md.update("tj_" + k, v)
        def graph_region():
            """--kernel_noise: the same region as one HIP-graph replay (see _GraphRegion); the post-timer metrics get an eager
            SceneBatch of the batch."""
            torch.cuda.synchronize()
            tttt1 = time.time()
            S = args.sampling_size
            nb = {k: batch_cuda[k] for k in ("ego_traj", "neighbors", "currlane_wpts", "leftlane_wpts", "rightlane_wpts",
                                             "curr_id", "left_id", "right_id", "gt_high_level", "pre_stlp") if k in batch_cuda}
            nb["neighbor_trajs_aug"] = batch_cuda["neighbors_traj"][..., :7]
            scene = {k: nb[k] for k in ("ego_traj", "neighbors", "currlane_wpts", "leftlane_wpts", "rightlane_wpts", "curr_id",
                                        "left_id", "right_id")}
            scene["neighbors_traj"] = nb["neighbor_trajs_aug"]
            if args.load_stlp:
                scene["stlp_rows"] = nb["pre_stlp"].reshape(bs, -1, 3, 6)[:, 0:1].repeat(1, S, 1, 1).reshape(bs * S * 3, 6)
            else:
                scene["stlp_rows"] = get_dense_stlp(nb, gt_stlp, args, n_randoms=S)[:, 0]
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())     # (as the eager --kernel_noise path keys its streams)
            cache = net.__dict__.setdefault("_graph_regions", {})
            key = (bs, S, scene["neighbors"].shape[1], net.packed().packed.data_ptr(), net.chain_arith())
            if key not in cache:
                cache.clear()
                cache[key] = _GraphRegion(net, args, coeffs, scene, S)
                torch.cuda.synchronize()
                tttt1 = time.time()            # (the capture itself is not part of a batch's time)
            ids_sum = sum(float(torch.as_tensor(batch[k]).to(torch.float32).sum()) for k in ("curr_id", "left_id", "right_id"))
            ctrl, sc, counts = cache[key].run(scene, seed, S, ids_sum)
            torch.cuda.synchronize()
            tttt2 = time.time()
            sb = SceneBatch(scene, S, _hp(args), states.device)
            # the printed rates with the harness's own float32 formula (mask_mean of torch means, as compute_stl_dense above:
            # the graph's integer counters give the same numbers up to the last bit of that formula)
            acc = mask_mean((sc > 0).float(), sb.valid)
            sacc = mask_mean((torch.max(sc.reshape(-1, S, 3), dim=1)[0] > 0).float(), sb.valid.reshape(-1, S, 3)[:, 0, :])
            return (ctrl.reshape(N, args.nt, 2), sc, acc, sacc, cache[key].sm, sb, tttt2 - tttt1)

        def timed_region(first=True):
            if _GraphRegion.supported(args):
                return graph_region()
            torch.cuda.synchronize()
            tttt1 = time.time()
            if myt and first:
                myt.next_batch()
            new_batch = {k: batch_cuda[k] for k in ("ego_traj", "neighbors", "currlane_wpts", "leftlane_wpts",
                                                    "rightlane_wpts", "curr_id", "left_id", "right_id", "gt_high_level",
                                                    "pre_stlp") if k in batch_cuda}
            new_batch["neighbor_trajs_aug"] = batch_cuda["neighbors_traj"][..., :7]
            new_batch = augment_batch_data(new_batch, gt_stlp, args, n_randoms=args.sampling_size)
            highlevel_new = new_batch["highlevel_dense"]
            states_flat_new = states.unsqueeze(1).unsqueeze(1).repeat(1, args.sampling_size, 3, 1).reshape(N, 4)
            noise = torch.empty(N, args.nt * 2, device=states.device)
            guidance_extras = (new_batch, states_flat_new, stls_cac) if args.guidance else None
            if myt:
                myt.add("start_diffusion")
            res = diffusion_rollout(noise, net, new_batch, highlevel_new, None, args, coeffs, fastforward=False,
                                    n_randoms=args.sampling_size, return_feature=True, guidance_extras=guidance_extras)
            if myt:
                myt.add("end_diffusion")
            if args.diff_full:
                nn_controls, feature, nn_controls_list = res
            else:
                nn_controls, feature = res
                nn_controls_list = None
            sb = new_batch["_pstl"]
            sm = Sampler(net.packed(), net.hparams(), chain_waves=net.chain_arith())
            if args.rect_head and not args.not_use_rect:
                if args.multi_cands is not None:
                    cands = torch.stack(nn_controls_list[-args.multi_cands:], dim=0).reshape(args.multi_cands, N, -1).contiguous()
                    r = sm.score(sb, cands, select=True)
                    nn_controls, prev_scores = r["sel_controls"].reshape(N, args.nt, 2), r["sel_scores"]
                    if myt:
                        myt.add("selected_stl_max")
                else:
                    prev_scores = sm.score(sb, nn_controls.reshape(1, N, -1).contiguous())["scores"][0]
                if not args.no_refinenet:
                    nn_controls = net.rect_forward(feature, highlevel_new, new_batch["stlp_dense"][:, 0], nn_controls, prev_scores)
                if myt:
                    myt.add("rect_forward()")
                for _ in range(args.n_rolls or 0):
                    sc = sm.score(sb, nn_controls.reshape(1, N, -1).contiguous())["scores"][0]
                    nn_controls = net.rect_forward(feature, highlevel_new, new_batch["stlp_dense"][:, 0], nn_controls, sc)
                if args.refinement:      # "further gradient" (reference :1034-1071): K = 8, 50 iterations, lr 0.3, thres 5e-4
                    clist = torch.stack(nn_controls_list, dim=0).reshape(len(nn_controls_list), N, -1).contiguous()
                    nn_controls = sm.refinement(sb, nn_controls.reshape(N, -1).contiguous(), clist).reshape(N, args.nt, 2)
            nn_trajs = generate_trajs(states_flat_new, nn_controls, args.dt).reshape(N, args.nt + 1, 4)
            stl_input = pre_prepare_stl_cache(new_batch, dense_trajs=nn_trajs[:, :-1])
            scores_list, scores, acc, scene_acc = compute_stl_dense(stl_input, stls_cac, new_batch["highlevel_dense"],
                                                                    stl_input["dense_valids"], args, scene=True)
            torch.cuda.synchronize()
            tttt2 = time.time()
            return nn_controls, scores, acc, scene_acc, sm, sb, tttt2 - tttt1

        # Domain of the default (split-f16) chain arithmetic, checked ONCE per batch where the harness synchronises anyway
        # (the per-call checks of Net.forward / rect_forward / diffusion_rollout are deferred for the timed region): a layer
        # input beyond the half range leaves undefined results and sets the packed buffer's status word.  The batch is then
        # run again on the exact-fp32 kernels, and so is everything after it (net.check_domain warns and switches).
        mode, net.domain_check = net.domain_check, "deferred"
        try:
            nn_controls, scores, acc, scene_acc, sm, sb, elapsed = timed_region()
            if net.check_domain():
                nn_controls, scores, acc, scene_acc, sm, sb, elapsed = timed_region(first=False)
        finally:
            net.domain_check = mode
        md.update("acc", acc.item())
        md.update("scene_acc", scene_acc.item())
        md.update("time", elapsed)
        # after the timer, as in the reference (nusc_train.py:1107-1130): diversity + ADE/FDE of the final samples
        _, _, div_totals = sm.diversity(sb, nn_controls.reshape(N, -1).contiguous(), scores.contiguous())
        for k, v in diversity_from_totals(div_totals).items():
            md.update(k, v)
        nan = float("nan")
        print("###[%02d] TJ acc:%.3f scene_acc:%.3f ade:%.3f fde:%.3f std:%.3f vol:%.3f area:%.3f s:%.3f u:%.3f| "
              "NN acc:%.3f scene_acc:%.3f ade:%.3f fde:%.3f std:%.3f vol:%.3f area:%.3f s:%.3f u:%.3f ||| T:%.3f" % (
                  bi, md("tj_acc"), md("tj_scene_acc"), md("tj_ade"), md("tj_fde"), md("tj_std"), md("tj_vol"), md("tj_area"),
                  md("tj_ent_s"), md("tj_ent_wa"), md("acc"), md("scene_acc"), md("ade"), md("fde"),
                  md("std"), md("vol"), md("area"), md("ent_s"), md("ent_wa"), md("time")))
    if myt:
        myt.print_profile()
    return md


# ---------------------------------------------------------------------------------------------------------------
# trajectory optimisation = the data-augmentation pass (reference --trajopt_only, nusc_train.py:1302-1349)
# ---------------------------------------------------------------------------------------------------------------
def save_trajopt_params(params, iter_i, traj_i, ti, args, save_stlp=None):
    """On-disk format of the traj-opt solutions (reference nusc_train.py:775-797; read back by nusc_dataset.py:203-225):
    one .npy per scene -- params_%05d_%04d{_init,}.npy (n_randoms,3,nt,2), scores_%05d_%04d.npy (n_randoms,3),
    params_%05d_%04d_stlp.npy (n_randoms,3,1,6)."""
    import os
    if args.test or not args.model_dir:
        return
    os.makedirs(args.model_dir, exist_ok=True)
    params_np = params.detach().cpu().numpy()
    bs = params_np.shape[0]
    stlp_np = None
    if save_stlp is not None:
        stlp_np = save_stlp.detach().cpu().numpy().reshape(bs, args.n_randoms, 3, 1, save_stlp.shape[-1])
    for i in range(bs):
        key = (int(traj_i[i]), int(ti[i]))
        if iter_i == "scores":
            name = "scores_%05d_%04d.npy" % key
        elif iter_i == "init":
            name = "params_%05d_%04d_init.npy" % key
        elif iter_i == "final":
            name = "params_%05d_%04d.npy" % key
        else:
            name = "params_%05d_%04d_iter%05d.npy" % (key + (int(iter_i),))
        np.save(os.path.join(args.model_dir, name), params_np[i])
        if stlp_np is not None:
            np.save(os.path.join(args.model_dir, "params_%05d_%04d_stlp.npy" % key), stlp_np[i])


def run_trajopt(data_loader, args):
    """For every batch: args.traj_opt_iters Adam iterations on the dataset's `params` under the STL loss, all inside one
    kernel launch (engine.Sampler.trajopt), then the reference's three files per scene."""
    from . import ffi as _ffi
    md = MeterDict()
    sm = Sampler.__new__(Sampler)
    sm.L = _ffi.lib()
    for bi, batch in enumerate(data_loader):
        batch_cuda = dict_to_cuda(batch)
        bs = batch_cuda["ego_traj"].shape[0]
        S = args.n_randoms
        N = bs * S * 3
        new_batch = {k: batch_cuda[k] for k in ("ego_traj", "neighbors", "currlane_wpts", "leftlane_wpts",
                                                "rightlane_wpts", "curr_id", "left_id", "right_id", "gt_high_level",
                                                "pre_stlp") if k in batch_cuda}
        new_batch["neighbor_trajs_aug"] = batch_cuda["neighbors_traj"][..., :7]
        gt_stlp = infer_gt_stlp(new_batch, batch_cuda["ego_traj"][..., :4], args)   # reference :1279, then get_dense_stlp
        new_batch = augment_batch_data(new_batch, gt_stlp, args)
        sb = new_batch["_pstl"]
        traj_i = batch_cuda.get("traj_i", torch.full((bs,), bi))
        ti = batch_cuda.get("ti", torch.arange(bs))
        params = batch_cuda["params"].reshape(N, -1).float().contiguous().clone()
        save_trajopt_params(params.reshape(bs, S, 3, args.nt, 2), "init", traj_i, ti, args, save_stlp=new_batch["stlp_dense"])
        torch.cuda.synchronize()
        t0 = time.time()
        scores, _ = sm.trajopt(sb, params, args.traj_opt_iters, args.trajopt_lr, args.stl_trajopt_thres, args.reg_loss)
        torch.cuda.synchronize()
        dt = time.time() - t0
        acc = mask_mean((scores >= 0).float(), sb.valid)
        md.update("avg_acc", acc.item())
        md.update("time", dt)
        save_trajopt_params(params.reshape(bs, S, 3, args.nt, 2), "final", traj_i, ti, args)
        save_trajopt_params(scores.reshape(bs, S, 3), "scores", traj_i, ti, args)
        print("trajopt batch %d: %d rows x %d iters in %.3f s (%.3e row-iterations/s), acc %.3f" % (
            bi, N, args.traj_opt_iters, dt, N * args.traj_opt_iters / dt, md["avg_acc"]))
    return md


# ---------------------------------------------------------------------------------------------------------------
# RefineNet training (reference main loop nusc_train.py:1237-1632 for --rect_head: configs e7_ours / e8_ours_ablation)
# ---------------------------------------------------------------------------------------------------------------
def run_training(data_loader, net, coeffs, args):
    """Optimises rect_net (the only parameters in the reference's optimiser without --joint, nusc_train.py:1230-1233;
    with --joint the whole net: the three scene encoders and merge_net receive gradients too, policy_net none):
    per batch, sampling under no-grad, candidate selection, RefineNet forward/backward under
      --diverse_loss:  loss_stl*stl_weight + loss_reg*rect_reg_loss + loss_diversity      (e7_ours,  :442-467)
      otherwise:       loss_stl*stl_weight                                                 (e8_ours_ablation, :468-478)
    and torch.optim.Adam.step().  Writes <model_dir>/model_last.ckpt after every epoch (utils.py:81-85)."""
    from .engine import RectTrainer
    if not args.rect_head:
        raise SystemExit("training of the denoiser itself (e5_ddpm) is outside this path; --rect_head trains RefineNet")
    if args.joint:      # reference :1230-1231
        optimizer = torch.optim.Adam(net.parameters(), lr=args.lr)
        params = dict(net.named_parameters())
    else:
        optimizer = torch.optim.Adam(net.rect_net.parameters(), lr=args.lr)
        params = {"rect_net." + k: p for k, p in net.rect_net.named_parameters()}
    e7 = None
    if args.diverse_loss:
        e7 = dict(stl_weight=args.stl_weight, diversity_weight=args.diversity_weight, diversity_scale=args.diversity_scale,
                  rect_reg_loss=args.rect_reg_loss, detach=args.diverse_detach)
    md = MeterDict()
    step = 0
    for epi in range(args.epochs):
        for bi, batch in enumerate(data_loader):
            batch_cuda = dict_to_cuda(batch)
            new_batch = {k: batch_cuda[k] for k in ("ego_traj", "neighbors", "currlane_wpts", "leftlane_wpts",
                                                    "rightlane_wpts", "curr_id", "left_id", "right_id", "gt_high_level",
                                                    "pre_stlp") if k in batch_cuda}
            new_batch["neighbor_trajs_aug"] = batch_cuda["neighbors_traj"][..., :7]
            # STL parameters of the ground-truth mode, as the sampling harness and the traj-opt loop get them (reference
            # :1279): from the file format's own keys, not from anything only the synthetic generator writes
            gt_stlp = infer_gt_stlp(new_batch, batch_cuda["ego_traj"][..., :4], args)
            new_batch = augment_batch_data(new_batch, gt_stlp, args)
            sb = new_batch["_pstl"]
            tr = RectTrainer(Sampler(net.packed(), net.hparams(), chain_waves=net.chain_arith()))     # packed() re-packs after the optimiser moved weights
            step += 1
            loss, scores = tr.train_step(sb, params, optimizer, args.diffusion_steps, seed=args.seed * 100003 + step,
                                         multi_cands=args.multi_cands or 1, coeffs=coeffs, e7=e7, stl_weight=args.stl_weight,
                                         merge=bool(args.diverse_loss and not args.no_arch), clip_rect=bool(args.clip_rect),
                                         joint=bool(args.joint))
            if tr.sm.chain_fallback and net.chain_waves != 8:
                # train_step found the split-f16 domain flag set (or a weight outside |w| < 63.9), repeated the step on the
                # exact-fp32 kernels and warned; the run stays there: weights that grew past the domain rarely come back
                net.chain_waves = 8
            counts, _ = tr.sm.metrics(sb, scores)
            acc, _ = acc_from_counts(counts)
            md.update("loss", float(loss))
            md.update("acc", acc)
            if bi % args.print_freq == 0:
                print("epoch %03d batch %03d loss %.5f (avg %.5f) acc %.3f" % (epi, bi, md["loss"], md("loss"), md["acc"]))
        if args.model_dir:
            from . import nusc_dataset
            nusc_dataset.save_checkpoint(net.state_dict(), args.model_dir)
    return md


def generate_parser(argv=None):
    """The reference's flags and post-parse overrides (nusc_train.py:1635-1814)."""
    parser = argparse.ArgumentParser("")
    add = parser.add_argument
    add("--seed", type=int, default=1007)
    add("--exp_name", "-e", type=str, default=None)
    add("--gpus", type=str, default="0")
    add("--epochs", type=int, default=500)
    add("--test", action="store_true", default=False)
    add("--net_pretrained_path", "-P", type=str, default=None)
    add("--allow_random_init", action="store_true", default=False)   # not a reference flag: -P may name a missing file
    # not a reference flag: x_T and the per-step noise of diffusion_rollout are drawn inside the HIP kernels (Philox4x32-7
    # keyed by a seed taken from torch's generator) instead of by one torch.randn_like call per reverse step
    add("--kernel_noise", action="store_true", default=False)
    add("--no_graph", action="store_true", default=False)     # --kernel_noise: eager launches instead of one HIP-graph replay per batch
    add("--num_workers", type=int, default=8)
    add("--batch_size", "-b", type=int, default=128)
    add("--lr", type=float, default=3e-4)
    add("--hiddens", type=int, nargs="+", default=[256, 256])
    add("--mini", action="store_true", default=False)
    add("--n_neighbors", "-N", type=int, default=8)
    add("--n_randoms", type=int, default=64)
    add("--n_segs", type=int, default=15)
    add("--ego_L", type=float, default=4.084)
    add("--ego_W", type=float, default=1.730)
    add("--refined_nL", type=int, default=4)
    add("--refined_nW", type=int, default=1)
    add("--nt", type=int, default=20)
    add("--dt", type=float, default=0.5)
    add("--mul_w_max", type=float, default=0.5)
    add("--mul_a_max", type=float, default=5.0)
    add("--smoothing_factor", type=float, default=100.0)
    add("--skip_nusc_load", action="store_true", default=False)
    add("--clip_dist", action="store_true", default=False)
    add("--stl_nn_thres", type=float, default=0.0005)
    add("--inline", action="store_true", default=False)
    add("--use_init_hint", action="store_true", default=False)
    add("--norm_stl", action="store_true", default=False)
    add("--flex", action="store_true", default=False)
    add("--load_stlp", action="store_true", default=False)
    add("--load_tj", action="store_true", default=False)
    add("--stl_weight", type=float, default=1.0)
    add("--bc", action="store_true", default=False)
    add("--vae", action="store_true", default=False)
    add("--diffusion", action="store_true", default=False)
    add("--diffusion_steps", type=int, default=100)
    add("--beta_start", type=float, default=1e-4)
    add("--beta_end", type=float, default=0.02)
    add("--cos", action="store_true", default=False)
    add("--grad_rollout", action="store_true", default=False)
    add("--rect_head", action="store_true", default=False)
    add("--rect_hiddens", type=int, nargs="+", default=[256, 256])
    add("--not_use_rect", action="store_true", default=False)
    add("--measure_diversity", action="store_true", default=False)
    add("--extra_diversity", action="store_true", default=False)
    add("--viz_correct", action="store_true", default=False)
    add("--run_sampling_test", action="store_true", default=False)
    add("--sampling_size", type=int, default=64)
    add("--n_trials", type=int, default=100)
    add("--diff_full", action="store_true", default=False)
    add("--diverse_loss", action="store_true", default=False)
    add("--no_arch", action="store_true", default=False)
    add("--n_shards", type=int, default=4)
    add("--diverse_fuse_type", type=str, default="add")
    add("--interval", action="store_true", default=False)
    add("--diffusion_clip", action="store_true", default=False)
    add("--multi_cands", type=int, default=None)
    add("--gt_data_training", action="store_true", default=False)
    add("--collision_loss", type=float, default=None)
    add("--guidance", action="store_true", default=False)
    add("--guidance_niters", type=int, default=3)
    add("--guidance_before", type=int, default=1000)
    add("--guidance_lr", type=float, default=0.01)
    add("--guidance_reverse", action="store_true", default=False)
    add("--guidance_sets", nargs="+", type=int, default=None)
    add("--guidance_freq", type=int, default=None)
    add("--oracle_filter", action="store_true", default=False)
    add("--clip_rect", action="store_true", default=False)
    add("--ego", action="store_true", default=False)
    add("--other", action="store_true", default=False)
    add("--n_rolls", type=int, default=None)
    add("--suffix", type=str, default=None)
    add("--no_refinenet", action="store_true", default=False)
    add("--time_profile", action="store_true", default=False)
    add("--stl_trajopt_thres", type=float, default=0.01)
    add("--trajopt_only", action="store_true", default=False)
    add("--traj_opt_iters", type=int, default=2000)
    add("--trajopt_lr", type=float, default=0.005)
    add("--opt_epochs", type=int, default=0)
    add("--reg_loss", type=float, default=10.0)
    add("--model_dir", type=str, default=None, help="where params_*.npy / scores_*.npy go (reference: exps/<run>/models)")
    add("--rect_reg_loss", type=float, default=0.0)
    add("--joint", action="store_true", default=False)
    add("--diversity_weight", type=float, default=1.0)
    add("--diversity_scale", type=float, default=1.0)
    add("--diverse_detach", action="store_true", default=False)
    add("--print_freq", type=int, default=10)
    add("--offline", action="store_true", default=False)
    add("--cache_path", type=str, default=None,
        help="experiment directory holding cache.npz, *_split.txt and models/ (nusc_dataset.write_synthetic_experiment); "
             "without it batches come straight from the synthetic scene generator")
    add("--test_t1", action="store_true", default=False)
    # the rest of the reference's flags (nusc_train.py:1635-1778), accepted so that its command lines parse unchanged;
    # the ones that select functionality outside this path are refused in main() with a message
    add("--anno_path", type=str, default="annotated_data_trainval")
    add("--backup", action="store_true", default=False)
    add("--bc_weight", type=float, default=0.0)
    add("--check_stl_params", action="store_true", default=False)
    add("--collect_data", action="store_true", default=False)
    add("--debug", action="store_true", default=False)
    add("--diffusion_weight", type=float, default=1.0)
    add("--epi_print_freq", type=int, default=1)
    add("--extra_rect_reg", type=float, default=0.0)
    add("--filter_traj", type=int, nargs="+", default=None)
    add("--generate_split_on_the_fly", action="store_true", default=False)
    add("--gt_nei", action="store_true", default=False)
    add("--lite_refine", action="store_true", default=False)
    add("--n_expands", type=int, default=4)
    add("--no_viz", action="store_true", default=False)
    add("--num_viz", type=int, default=10)
    add("--params_load_path", "-P2", type=str, default="e1_nusc_trajopt")
    add("--raw_refinement", action="store_true", default=False)
    add("--refined_safety", action="store_true", default=False)
    add("--refinement", action="store_true", default=False)
    add("--replace_hint", action="store_true", default=False)
    add("--save_freq", type=int, default=100)
    add("--stl_bc_mask", action="store_true", default=False)
    add("--test_aggressive", action="store_true", default=False)
    add("--test_scenes", action="store_true", default=False)
    add("--train_ratio", type=float, default=0.7)
    add("--trajopt_save_freq", type=int, default=1000)
    add("--use_gt_stlp", action="store_true", default=False)
    add("--vae_dim", type=int, default=64)
    add("--viz_freq", type=int, default=50)
    add("--viz_last", action="store_true", default=False)
    add("--weight_vae_bc", type=float, default=1.0)
    add("--weight_vae_kl", type=float, default=1.0)
    args = parser.parse_args(argv)
    args.cos = True
    args.measure_diversity = True
    if args.run_sampling_test:
        args.test = True
        args.extra_diversity = True
    if args.trajopt_only:      # reference nusc_train.py:1794-1801
        args.opt_epochs = 1
        args.epochs = 1
        args.batch_size = 1024
        args.diffusion = True
        args.flex = True
    if args.opt_epochs > 0:
        args.epochs = args.opt_epochs
    if args.load_stlp:
        args.load_tj = True
    if args.rect_head:
        args.interval = True
        args.diffusion_clip = True
        args.diff_full = True
    if args.test:
        args.epochs = 1
    return args


def main(argv=None):
    args = generate_parser(argv)
    for flag, why in (("collect_data", "dataset extraction needs the nuScenes devkit"), ("bc", "the BC baseline"),
                      ("vae", "the VAE baseline"),
                      ("gt_data_training", "the mono (GT-data) training mode"), ("check_stl_params", "a data-inspection tool")):
        if getattr(args, flag, False):
            raise SystemExit("--%s selects %s, which is outside the path this package implements" % (flag, why))

    def loader_for(split, n_batches=None):
        if args.cache_path:      # from files, as the reference does (cache.npz + split file + models/*.npy)
            from . import nusc_dataset
            return nusc_dataset.get_dataloader(args, args.cache_path, split=split)
        return SyntheticLoader(args, n_batches=n_batches)

    if args.trajopt_only:
        if args.norm_stl:
            raise SystemExit("--norm_stl is not supported together with --trajopt_only (the traj-opt loop is built for the "
                             "default formulas)")
        torch.manual_seed(args.seed)
        if args.cache_path and not args.model_dir:
            import os
            args.model_dir = os.path.join(args.cache_path, "models")
        return run_trajopt(loader_for("train", n_batches=min(args.n_trials, 2)), args)
    if args.sampling_size != args.n_randoms:
        raise SystemExit("--sampling_size must equal --n_randoms (merge_net pooling, reference nusc_model.py:187-196)")
    if args.norm_stl and args.refinement:
        # the fused scoring / guidance / training kernels honour --norm_stl (PSTL_FLAG_NORM_STL); the two many-iteration
        # loops (--refinement, --trajopt_only) are built for the default formulas
        raise SystemExit("--norm_stl is not supported together with --refinement")
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    loader = loader_for("val")
    stls_cac = build_stl_cache(args)
    net = Net(args).cuda()
    if args.net_pretrained_path is not None:
        import os
        path = args.net_pretrained_path
        if not os.path.isfile(path):
            path = os.path.join("exps", path, "models", "model_last.ckpt")
        if os.path.isfile(path):
            net.load_state_dict(torch.load(path, map_location="cuda"), strict=(not args.rect_head))
        elif args.allow_random_init:
            print("checkpoint %s not found: running with random-init weights (seed %d)" % (path, args.seed))
        else:   # the reference fails in torch.load here; numbers from an untrained net must not pass for a run of -P
            raise SystemExit("checkpoint %s not found (pass --allow_random_init to run with random-init weights)" % path)
    coeffs = get_diffusion_coeffs(args)
    if not args.run_sampling_test:
        if args.test:
            raise SystemExit("--test without --run_sampling_test: nothing to do on this path")
        if args.cache_path and not args.model_dir:
            import os
            args.model_dir = os.path.join(args.cache_path, "models")
        return run_training(loader_for("train", n_batches=min(args.n_trials, 4)), net, coeffs, args)
    return run_sampling_test(stls_cac, loader, net, coeffs, args, None, None)


if __name__ == "__main__":
    t1 = time.time()
    main()
    print("Finished in %.3f seconds" % (time.time() - t1))
