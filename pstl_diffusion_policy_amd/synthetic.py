"""Seeded synthetic scene batches with the schema the reference dataset hands the hot path.

Schema follows SURVEY.md section 3.0 (reference: nusc_dataset.py:109-244, nusc_api.py:126-144,613-683):
  ego_traj (bs,nt,6) x,y,th,v,L,W ; neighbors (bs,K,7) valid,x,y,th,v,L,W ; neighbors_traj (bs,K,nt,7) ;
  {curr,left,right}lane_wpts (bs,n_segs,3) x,y,th (all-zero when the lane is invalid) ; {curr,left,right}_id (bs,1) ;
  gt_high_level (bs,1) ; stlp_modes (bs,3,6) = (vmin,vmax,dmin,dmax,dsafe,thmax) per (scene, mode) ;
  pre_stlp (bs,S,3,1,6) = stlp_modes broadcast over samples (reference: nusc_train.py:745).

Everything is generated on the CPU with a private torch.Generator so that results do not depend on
global RNG state, then handed to the caller (which moves it to the device).
"""
import math

import torch


def make_scene_batch(bs, K=8, nt=20, n_segs=15, S=64, seed=0, dt=0.5, invalid_lane_frac=0.0,
                     random_pose=True, curved=True, stlp_mode="loose"):
    g = torch.Generator().manual_seed(int(seed))

    def U(lo, hi, *shape):
        return torch.rand(*shape, generator=g) * (hi - lo) + lo

    # --- road frame: three roughly parallel lanes, ego at the origin heading +x -------------------
    xs = torch.linspace(0.0, 60.0, n_segs).reshape(1, n_segs).repeat(bs, 1) - 5.0
    if curved:
        kappa = U(-0.004, 0.004, bs, 1)
    else:
        kappa = torch.zeros(bs, 1)
    lane_off = {"curr": 0.0, "left": 4.0, "right": -4.0}
    lanes_local = {}
    for key, off in lane_off.items():
        y = 0.5 * kappa * xs * xs + off
        th = torch.atan(kappa * xs)
        lanes_local[key] = torch.stack([xs, y, th], dim=-1)

    ego_v0 = U(5.0, 8.0, bs)
    ego_y0 = U(-0.4, 0.4, bs)
    ego_th0 = U(-0.05, 0.05, bs)
    tt = torch.arange(nt).float().reshape(1, nt) * dt
    ego_x = ego_v0[:, None] * tt * torch.cos(ego_th0)[:, None]
    ego_y = ego_y0[:, None] + ego_v0[:, None] * tt * torch.sin(ego_th0)[:, None]
    ego_local = torch.stack([ego_x, ego_y, ego_th0[:, None].repeat(1, nt), ego_v0[:, None].repeat(1, nt),
                             torch.full((bs, nt), 4.0), torch.full((bs, nt), 2.0)], dim=-1)

    valid = (torch.rand(bs, K, generator=g) < 0.5).float()
    nx0 = U(10.0, 50.0, bs, K)
    lane_pick = torch.randint(0, 3, (bs, K), generator=g)
    ny0 = torch.tensor([-4.0, 0.0, 4.0])[lane_pick] + U(-0.3, 0.3, bs, K)
    nv = U(3.0, 7.0, bs, K)
    nth = U(-0.03, 0.03, bs, K)
    nL = U(4.2, 4.8, bs, K)
    nW = U(1.8, 2.0, bs, K)
    ntt = tt.reshape(1, 1, nt)
    nxt = nx0[..., None] + nv[..., None] * ntt * torch.cos(nth)[..., None]
    nyt = ny0[..., None] + nv[..., None] * ntt * torch.sin(nth)[..., None]
    nei_local = torch.stack([nxt, nyt, nth[..., None].repeat(1, 1, nt), nv[..., None].repeat(1, 1, nt),
                             nL[..., None].repeat(1, 1, nt), nW[..., None].repeat(1, 1, nt)], dim=-1)  # (bs,K,nt,6)

    # --- random rigid transform into a world frame (exercises the ego-frame normalisation) --------
    if random_pose:
        wx = U(-200.0, 200.0, bs)
        wy = U(-200.0, 200.0, bs)
        wth = U(-math.pi, math.pi, bs)
    else:
        wx = torch.zeros(bs)
        wy = torch.zeros(bs)
        wth = torch.zeros(bs)

    def to_world(xyth, extra_dims):
        shp = [bs] + [1] * extra_dims
        c = torch.cos(wth).reshape(shp)
        s = torch.sin(wth).reshape(shp)
        x, y, th = xyth[..., 0], xyth[..., 1], xyth[..., 2]
        X = x * c - y * s + wx.reshape(shp)
        Y = x * s + y * c + wy.reshape(shp)
        TH = th + wth.reshape(shp)
        return torch.stack([X, Y, TH], dim=-1)

    ego_traj = torch.cat([to_world(ego_local[..., :3], 1), ego_local[..., 3:]], dim=-1)
    nei_w = torch.cat([to_world(nei_local[..., :3], 2), nei_local[..., 3:]], dim=-1)
    neighbors_traj = torch.cat([valid[:, :, None, None].repeat(1, 1, nt, 1), nei_w], dim=-1)
    neighbors_traj = neighbors_traj * valid[:, :, None, None]  # invalid neighbours are all-zero rows
    neighbors = neighbors_traj[:, :, 0, :].clone()

    batch = {"ego_traj": ego_traj.contiguous(), "neighbors": neighbors.contiguous(),
             "neighbors_traj": neighbors_traj.contiguous()}
    ids = {}
    for key in ["curr", "left", "right"]:
        w = to_world(lanes_local[key], 1)
        if key != "curr" and invalid_lane_frac > 0:
            ok = (torch.rand(bs, generator=g) >= invalid_lane_frac).float()
        else:
            ok = torch.ones(bs)
        batch["%slane_wpts" % key] = (w * ok[:, None, None]).contiguous()
        ids[key] = ok
        batch["%s_id" % key] = ok.reshape(bs, 1)
    batch["gt_high_level"] = torch.zeros(bs, 1)

    # --- STL parameters per (scene, mode) ----------------------------------------------------------
    if stlp_mode == "fixed":   # the closed-loop driver's constants (reference: nusc_sim.py:467-472)
        stlp = torch.tensor([1.0, 9.0, -3.0, 2.0, 0.1, 0.2]).reshape(1, 1, 6).repeat(bs, 3, 1)
    elif stlp_mode == "wide":  # thresholds around the quantiles of a random-init sampler: roughly half the rows satisfy
        vmin = ego_v0[:, None] - U(5.0, 14.0, bs, 3)
        vmax = ego_v0[:, None] + U(5.0, 14.0, bs, 3)
        stlp = torch.stack([vmin, vmax, -U(5.0, 40.0, bs, 3), U(5.0, 40.0, bs, 3), -U(0.0, 3.0, bs, 3),
                            U(0.3, 1.5, bs, 3)], dim=-1)
    else:
        wide = 1.0 if stlp_mode == "loose" else 0.3
        vmin = ego_v0[:, None] - U(1.3, 3.0, bs, 3) * wide
        vmax = ego_v0[:, None] + U(1.3, 3.0, bs, 3) * wide
        dmin = U(-2.5, -0.5, bs, 3)
        dmax = U(0.5, 2.5, bs, 3)
        dsafe = U(0.0, 0.6, bs, 3)
        thmax = U(0.2, 0.6, bs, 3)
        stlp = torch.stack([vmin, vmax, dmin, dmax, dsafe, thmax], dim=-1)
    batch["stlp_modes"] = stlp.contiguous()
    batch["pre_stlp"] = stlp.reshape(bs, 1, 3, 1, 6).repeat(1, S, 1, 1, 1).contiguous()
    # traj-opt controls of the data generator (only consumed by the TJ reference branch of the harness)
    w = U(-0.05, 0.05, bs, S, 3, nt)
    a = U(-1.0, 1.0, bs, S, 3, nt)
    batch["params"] = torch.stack([w, a], dim=-1).contiguous()
    batch["tj_scores_prior"] = torch.zeros(bs, S, 3)
    return batch


def default_hparams():
    """Hot-path constants = the reference parser defaults (nusc_train.py:1665-1673,1683,1742)."""
    return dict(nt=20, dt=0.5, mul_w_max=0.5, mul_a_max=5.0, smoothing_factor=100.0, stl_nn_thres=5e-4,
                ego_L=4.084, ego_W=1.730, refined_nL=4, refined_nW=1, n_segs=15, n_shards=4)
