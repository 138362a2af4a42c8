"""CPU oracle for the DDPM-sampling + STL hot path.  TEST INFRASTRUCTURE ONLY.

This file is a float32 restatement (torch on the CPU) of the reference algorithm for the path named
in BASELINE.json (reference region nusc_train.py:957-1105).  It is the *checker*: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The product path
(pstl_diffusion_policy_amd/) never imports it and fails loudly without its HIP library.

Parity pinning: the reference holds no tests/golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, run in the build container and committed
as fixtures under tests/golden/ (generator: tests/golden/make_golden.py; check:
tests/test_oracle_golden.py).

Each function cites the reference lines it restates.  Inputs are scene-indexed (bs scenes, S samples,
3 modes; row r = (b*S + s)*3 + mode, reference nusc_train.py:20,724-754) and are expanded to rows here.
Weights are a dict with the reference's state_dict keys (numpy or torch float32).
"""
import numpy as np
import torch

F32 = torch.float32


def _t(x):
    if isinstance(x, torch.Tensor):
        return x.to(F32)
    return torch.from_numpy(np.ascontiguousarray(x)).to(F32)


def rows_from_scenes(x, reps):
    """(bs, ...) -> (bs*reps, ...), each scene repeated `reps` times contiguously (reference dup(), nusc_train.py:20)."""
    return torch.repeat_interleave(x, reps, dim=0)


# ------------------------------------------------------------------------------------------------
# A4  noise schedule (reference nusc_train.py:528-537, cosine branch; --cos is forced on at :1782)
# ------------------------------------------------------------------------------------------------
def diffusion_coeffs(steps):
    t = torch.linspace(0, 1, steps + 1)
    ab = torch.cos((t + 0.008) / 1.008 * np.pi / 2) ** 2
    beta = torch.clip(1 - ab[1:] / ab[:-1], 0, 0.999) * 0.2
    alpha = 1.0 - beta
    alpha_hat = torch.cumprod(alpha, dim=0)
    return beta, alpha, alpha_hat


# ------------------------------------------------------------------------------------------------
# A1-A3  network (reference nusc_model.py:20-53,55-95,97-180,238-263; utils.py:91-101)
# ------------------------------------------------------------------------------------------------
def relu_mlp(sd, prefix, x):
    h = x
    for i, idx in enumerate((0, 2, 4)):
        w = _t(sd["%s.%d.weight" % (prefix, idx)])
        b = _t(sd["%s.%d.bias" % (prefix, idx)])
        h = torch.addmm(b, h.reshape(-1, h.shape[-1]), w.t()).reshape(h.shape[:-1] + (w.shape[0],))
        if i < 2:
            h = torch.relu(h)
    return h


def pos_encoding(t, channels=32):
    """t: (n,1) float; [sin(t f_k) | cos(t f_k)], f_k = 10000^(-2k/channels)  (nusc_model.py:48-53)."""
    inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
    arg = t.repeat(1, channels // 2) * inv_freq
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


def to_ego_frame(state, base, valid=None):
    """nusc_model.py:238-263 (translation/heading are only removed where valid==1)."""
    x, y, th = state[..., 0], state[..., 1], state[..., 2]
    bx, by, bth = base[..., 0], base[..., 1], base[..., 2]
    if valid is not None:
        xt, yt, tht = x - bx * valid, y - by * valid, th - bth * valid
    else:
        xt, yt, tht = x - bx, y - by, th - bth
    xr = xt * torch.cos(bth) + yt * torch.sin(bth)
    yr = -xt * torch.sin(bth) + yt * torch.cos(bth)
    return torch.stack([xr, yr, tht], dim=-1)


def encode_feat(sd, scene):
    """Scene feature (bs,224): ego | min/mean/max over neighbours | 3 lanes  (nusc_model.py:55-95)."""
    ego = _t(scene["ego_traj"])[:, 0]                       # (bs,6)
    bs = ego.shape[0]
    ego_un = ego.unsqueeze(1)
    neis = _t(scene["neighbors"])                           # (bs,K,7)
    n_xyth = to_ego_frame(neis[..., 1:4], ego_un, neis[..., 0])
    nei_in = torch.cat([neis[..., 0:1], n_xyth, neis[..., 4:7]], dim=-1)
    lanes = []
    for key in ("curr", "left", "right"):
        lanes.append(to_ego_frame(_t(scene["%slane_wpts" % key]), ego_un, _t(scene["%s_id" % key])))
    lanes = torch.stack(lanes, dim=1)                       # (bs,3,nseg,3)
    lane_in = torch.cat([lanes[..., 0:1, :], lanes[..., 1:, :] - lanes[..., :-1, :]], dim=-2).reshape(bs, 3, -1)
    ego_in = torch.cat([to_ego_frame(ego[..., :3], ego[..., :3]), ego[..., 3:]], dim=-1)
    ego_f = relu_mlp(sd, "ego_encoder", ego_in)
    nei_f = relu_mlp(sd, "neighbor_encoder", nei_in)
    nei_f = torch.cat([nei_f.min(dim=1)[0], nei_f.mean(dim=1), nei_f.max(dim=1)[0]], dim=-1)
    lane_f = relu_mlp(sd, "lane_encoder", lane_in).reshape(bs, -1)
    return torch.cat([ego_f, nei_f, lane_f], dim=-1)


def policy_eps(sd, feature_rows, x, t_int, hl, stlp):
    """Predicted noise = policy_net([feature|x|pe(t)|hl|stlp]) + x   (nusc_model.py:119-121,159-162)."""
    n = x.shape[0]
    pe = pos_encoding(torch.full((n, 1), float(t_int)), 32)
    inp = torch.cat([feature_rows, x, pe, hl, stlp], dim=-1)
    return relu_mlp(sd, "policy_net", inp) + x


def normalize_controls(x, w_max, a_max, clip):
    """nusc_train.py:647-655."""
    x = x.reshape(x.shape[0], -1, 2)
    w = x[..., 0] * w_max
    a = x[..., 1] * a_max
    if clip:
        w = torch.clip(w, -w_max, w_max)
        a = torch.clip(a, -a_max, a_max)
    return torch.stack([w, a], dim=-1)


# ------------------------------------------------------------------------------------------------
# A6  unicycle rollout (reference nusc_train.py:29-49)
# ------------------------------------------------------------------------------------------------
def unicycle_rollout(s0, u, dt):
    states = [s0]
    for ti in range(u.shape[-2]):
        x, y, th, v = states[-1].unbind(dim=-1)
        ds = torch.stack([v * torch.cos(th), v * torch.sin(th), u[..., ti, 0], u[..., ti, 1]], dim=-1)
        states.append(states[-1] + ds * dt)
    return torch.stack(states, dim=-2)


# ------------------------------------------------------------------------------------------------
# A9  point-to-lane distance (reference nusc_api.py:685-739, "efficient" branch, inline=False, clip=False)
# ------------------------------------------------------------------------------------------------
def lane_distance(points, lane):
    """points (R,T,3), lane (R,nseg,3) -> signed lateral distance (R,T), heading error 1-cos (R,T)."""
    px, py, pth = points[..., 0], points[..., 1], points[..., 2]
    lx, ly, lth = lane[:, None, :, 0], lane[:, None, :, 1], lane[:, None, :, 2]
    dist = torch.sqrt((px[..., None] - lx) ** 2 + (py[..., None] - ly) ** 2)       # (R,T,nseg)
    j = torch.argmin(dist[..., :-1] + dist[..., 1:], dim=-1, keepdim=True)          # first index on ties
    T = points.shape[1]
    x2 = torch.gather(lx.expand(-1, T, -1), 2, j)[..., 0]
    y2 = torch.gather(ly.expand(-1, T, -1), 2, j)[..., 0]
    h2 = torch.gather(lth.expand(-1, T, -1), 2, j)[..., 0]
    x3 = torch.gather(lx.expand(-1, T, -1), 2, j + 1)[..., 0]
    y3 = torch.gather(ly.expand(-1, T, -1), 2, j + 1)[..., 0]
    area = px * (y2 - y3) + x2 * (y3 - py) + x3 * (py - y2)
    seg = torch.sqrt((x2 - x3) ** 2 + (y2 - y3) ** 2)
    pt = torch.clamp((px - x2) ** 2 + (py - y2) ** 2, 1e-3) ** 0.5
    normal = (seg != 0).float()
    d = normal * area / torch.clip(seg, 1e-7) + (1 - normal) * pt
    return d, 1 - torch.cos(h2 - pth)


# ------------------------------------------------------------------------------------------------
# A10 car-to-car clearance (reference nusc_train.py:142-148, utils.py:465-526; nL=4, nW=1)
# ------------------------------------------------------------------------------------------------
def _circle_row(x, y, th, L, W, nL=4):
    r = torch.minimum(torch.maximum(L / nL / 2, W / 2), W / 2)
    a = torch.linspace(0, 1, nL)
    off = (-L / 2 + r)[..., None] * (1 - a) + (L / 2 - r)[..., None] * a          # along the body axis
    lat = (-W / 2 + r)[..., None]                                                   # == 0 for nW=1
    cx = off * torch.cos(th[..., None]) - lat * torch.sin(th[..., None]) + x[..., None]
    cy = off * torch.sin(th[..., None]) + lat * torch.cos(th[..., None]) + y[..., None]
    return cx, cy, r


def neighbor_clearance(ego, nei, ego_L, ego_W, nL=4):
    """ego (R,T,>=3), nei (R,K,T,7)=valid,x,y,th,v,L,W -> min over neighbours of clipped clearance (R,T)."""
    e = ego.unsqueeze(1)
    ones = torch.ones_like(e[..., 0])
    ex, ey, er = _circle_row(e[..., 0], e[..., 1], e[..., 2], ego_L * ones, ego_W * ones, nL)
    nx, ny, nr = _circle_row(nei[..., 1], nei[..., 2], nei[..., 3], nei[..., 5], nei[..., 6], nL)
    dd = torch.sqrt((ex[..., :, None] - nx[..., None, :]) ** 2 + (ey[..., :, None] - ny[..., None, :]) ** 2)
    gap = dd.reshape(dd.shape[:-2] + (nL * nL,)).min(dim=-1)[0] - er - nr          # (R,K,T)
    valid = nei[..., 0]
    return torch.min(torch.clip(gap, -5, 20) * valid + (1 - valid) * 100, dim=1)[0]


# ------------------------------------------------------------------------------------------------
# A8  STL robustness (reference stl_d_lib.py:6-26,70-112,144-169; nusc_train.py:95-140,150-151,318-345)
# ------------------------------------------------------------------------------------------------
def soft_max(x, tau, dim=1):
    return torch.logsumexp(x * tau, dim=dim) / tau


def soft_min(x, tau, dim=1):
    return -soft_max(-x, tau, dim)


def always_from(s, tau):
    """G s [t] = softmin(s[t:T])  -- Always(0, nt) with the window clipped to the horizon (stl_d_lib.py:161-165)."""
    T = s.shape[1]
    return torch.stack([soft_min(s[:, t:T], tau) for t in range(T)], dim=1)


def eventually_within(s, tau, w):
    """F_w s [t] = softmax(s[t:min(t+w,T)])  -- Eventually(0, w), end exclusive (stl_d_lib.py:148-152)."""
    T = s.shape[1]
    return torch.stack([soft_max(s[:, t:min(t + w, T)], tau) for t in range(T)], dim=1)


def stl_signals(traj, nei, lanes, ego_L, ego_W):
    sig = {"v": traj[..., 3]}
    for key, lane in zip(("curr", "left", "right"), lanes):
        sig["d_" + key], sig["th_" + key] = lane_distance(traj[..., 0:3], lane)
    sig["nei"] = neighbor_clearance(traj, nei, ego_L, ego_W)
    return sig


def stl_scores_from_signals(sig, stlp, hl, tau, nt, norm=False):
    """norm: --norm_stl (nusc_train.py:88-91,97-113): the speed, lane-distance and clearance predicates are divided by
    v_factor = clip(vmax - vmin, 0.3), d_factor = clip((dmax - dmin) * 5, 0.3), safe_factor = clip(dsafe, 0.3)."""
    vmin, vmax, dmin, dmax, dsafe, thmax = [stlp[:, i:i + 1] for i in range(6)]
    g = lambda s: always_from(s, tau)
    f = lambda s: eventually_within(s, tau, nt // 2)
    if norm:
        vf, df, sf = torch.clip(vmax - vmin, 0.3), torch.clip((dmax - dmin) * 5, 0.3), torch.clip(dsafe, 0.3)
    else:
        vf = df = sf = None
    over = lambda a, fac: a / fac if norm else a
    keep_vmin = g(over(sig["v"] - vmin, vf))
    keep_vmax = g(over(-sig["v"] + vmax, vf))
    safe = g(over(sig["nei"] - dsafe, sf))
    out = []
    # stay in lane: conjunction of six "always" terms (nusc_train.py:115-118,130,132,136)
    terms = [keep_vmin, keep_vmax, g(over(sig["d_curr"] - dmin, df)), g(over(-sig["d_curr"] + dmax, df)),
             g((thmax - sig["th_curr"]) / thmax), safe]
    out.append(soft_min(torch.stack(terms, dim=1), tau)[:, 0])
    # change lane: eventually-always inside the target lane band and aligned (nusc_train.py:119-128,133-138)
    for key in ("left", "right"):
        band = -soft_max(torch.stack([-over(sig["d_" + key] - dmin, df), -over(-sig["d_" + key] + dmax, df)], dim=1), tau)
        terms = [keep_vmin, keep_vmax, f(g(band)), f(g((thmax - sig["th_" + key]) / thmax)), safe]
        out.append(soft_min(torch.stack(terms, dim=1), tau)[:, 0])
    scores3 = torch.stack(out, dim=0)
    hl = hl.reshape(-1)
    score = (scores3[0] * (hl == 0).float() + scores3[1] * (hl == 1).float() + scores3[2] * (hl == 2).float()
             + 1.0 * (hl == 3).float())
    return scores3, score


def stl_scores(traj, nei, lanes, stlp, hl, hp):
    """traj (R,T,4) -> scores of the three formulas (3,R) and the mode-selected score (R,)."""
    sig = stl_signals(traj, nei, lanes, hp["ego_L"], hp["ego_W"])
    s3, s = stl_scores_from_signals(sig, stlp, hl, hp["smoothing_factor"], traj.shape[1], norm=bool(hp.get("norm_stl", False)))
    return s3, s, sig


def mask_mean(x, m):
    """nusc_train.py:23-27."""
    return torch.mean(x * m) / torch.clip(torch.mean(m), 1e-2)


def stl_metrics(scores, valid_rows, S):
    """acc and scene_acc (nusc_train.py:332,339-343). valid_rows (R,) 0/1."""
    acc = mask_mean((scores > 0).float(), valid_rows)
    cube = scores.reshape(-1, S, 3)
    vcube = valid_rows.reshape(-1, S, 3)
    scene_acc = mask_mean((cube.max(dim=1)[0] > 0).float(), vcube[:, 0, :])
    return acc, scene_acc


# ------------------------------------------------------------------------------------------------
# Scene -> row expansion shared by the pieces below
# ------------------------------------------------------------------------------------------------
class Rows:
    def __init__(self, scene, S, hp):
        self.S = S
        self.hp = hp
        self.bs = int(np.asarray(scene["ego_traj"]).shape[0])
        m = 3 * S
        self.N = self.bs * m
        self.s0 = rows_from_scenes(_t(scene["ego_traj"])[:, 0, :4], m)
        self.nei = rows_from_scenes(_t(scene["neighbors_traj"])[..., :7], m)
        self.lanes = [rows_from_scenes(_t(scene["%slane_wpts" % k]), m) for k in ("curr", "left", "right")]
        if "stlp_rows" in scene:      # per-row parameters (the traj-opt loop draws them per sample, get_dense_stlp)
            self.stlp = _t(scene["stlp_rows"]).reshape(self.N, 6)
        else:
            stlp_modes = _t(scene["stlp_modes"])                                      # (bs,3,6)
            self.stlp = stlp_modes[:, None].repeat(1, S, 1, 1).reshape(self.N, 6)    # nusc_train.py:745
        self.hl = torch.tensor([0.0, 1.0, 2.0]).repeat(self.bs * S).reshape(self.N, 1)  # nusc_train.py:753
        ids = torch.cat([_t(scene["curr_id"]), _t(scene["left_id"]), _t(scene["right_id"])], dim=-1)
        self.valid = rows_from_scenes(ids, S).reshape(self.N)                        # nusc_train.py:751-752

    def score(self, controls, reps=1):
        rep = lambda x: x.repeat((reps,) + (1,) * (x.dim() - 1))
        traj = unicycle_rollout(rep(self.s0), controls, self.hp["dt"])[:, :-1]
        return stl_scores(traj, rep(self.nei), [rep(l) for l in self.lanes], rep(self.stlp), rep(self.hl), self.hp)


# ------------------------------------------------------------------------------------------------
# A7  guidance (reference nusc_train.py:589-627)
# ------------------------------------------------------------------------------------------------
def guidance_update(rows, mu, beta_t, lr, niters, maximize=False):
    """Observable behaviour of the reference block, which is NOT what its source reads like at first sight:
    `mu_opt = mu_init.detach().requires_grad_()` (nusc_train.py:606) shares storage with `mu_init`, so the first
    in-place Adam step (:623) moves `mu_init` too; `abs(mu_opt - mu_init)` (:625) is then identically 0 and the
    first iteration is a plain, signed, un-clipped Adam step.  `mu_opt.data = ...` (:626) then re-points mu_opt at
    fresh storage, so from the second iteration on the update is `anchor + clip(|mu_opt - anchor|, -beta, beta)`
    with anchor = the value after the first Adam step.  (Confirmed against the golden vectors e7_guid*, e5_guid_all.)"""
    hp = rows.hp
    N = rows.N
    mu_opt = mu.reshape(N, -1, 2).detach().clone().requires_grad_()
    opt = torch.optim.Adam([mu_opt], lr=lr)
    scale = torch.tensor([hp["mul_w_max"], hp["mul_a_max"]])
    anchor = None
    for j in range(niters):
        with torch.enable_grad():
            _, scores, _ = rows.score(mu_opt * scale)
            thr = 100.0 if maximize else hp["stl_nn_thres"]
            loss = mask_mean(torch.relu(thr - scores), rows.valid)
        opt.zero_grad()
        loss.backward()
        opt.step()
        with torch.no_grad():
            if j == 0:
                anchor = mu_opt.detach().clone()
            else:
                b = float(beta_t)
                mu_opt.data = anchor + torch.clip(torch.abs(mu_opt - anchor), -b, b)
    return mu_opt.detach().reshape(N, -1)


def guidance_triggered(i, steps, g):
    """nusc_train.py:589-598."""
    if not g or not g.get("enabled", False):
        return False
    i_val = steps - 1 - i if g.get("reverse", False) else i
    if g.get("sets") is not None:
        return i_val in g["sets"]
    if g.get("freq") is not None:
        return i_val % g["freq"] == 0
    return i <= g.get("before", 1000)


# ------------------------------------------------------------------------------------------------
# A5  reverse diffusion (reference nusc_train.py:557-645)
# ------------------------------------------------------------------------------------------------
def rollout(sd, scene, rows, x_T, z, steps, guidance=None, feature=None):
    """Returns (list of `steps` un-normalised states x, scene feature (bs,224)).  z[k] is the noise of the k-th
    reverse step (i = steps-1-k); the reference draws zeros for the last step (i == 1)."""
    beta, alpha, alpha_hat = diffusion_coeffs(steps)
    if feature is None:
        feature = encode_feat(sd, scene)
    feat_rows = rows_from_scenes(feature, 3 * rows.S)
    x = _t(x_T)
    states = [x]
    with torch.no_grad():
        for k, i in enumerate(reversed(range(1, steps))):
            eps = policy_eps(sd, feat_rows, x, i, rows.hl, rows.stlp)
            a, ah, b = alpha[i], alpha_hat[i], beta[i]
            mu = 1 / torch.sqrt(a) * (x - ((1 - a) / (torch.sqrt(1 - ah))) * eps)
            if guidance_triggered(i, steps, guidance):
                mu = guidance_update(rows, mu, b.item(), guidance["lr"], guidance["niters"],
                                     guidance.get("maximize", False))
            noise = _t(z[k]) if i > 1 else torch.zeros_like(x)
            x = mu + torch.sqrt(b) * noise
            states.append(x)
    return states, feature


# ------------------------------------------------------------------------------------------------
# A11 candidate selection + RefineNet (reference nusc_train.py:993-1013, nusc_model.py:182-235)
# ------------------------------------------------------------------------------------------------
def select_candidates(cands, cand_scores):
    """cands (mc,N,T,2), cand_scores (mc,N): best score per row, lowest candidate index on ties."""
    best, idx = torch.max(cand_scores, dim=0)
    n = cands.shape[1]
    return cands[idx, torch.arange(n)], best, idx


def rect_forward(sd, feature, rows, init_controls, scores, n_shards, diverse=True, clip_rect=False):
    hp = rows.hp
    N, S, bs = rows.N, rows.S, rows.bs
    nt2 = init_controls.shape[1] * 2
    feat_rows = rows_from_scenes(feature, 3 * S)
    if diverse:
        fused = relu_mlp(sd, "merge_net", init_controls.reshape(N, nt2))
        fused = fused.reshape(bs, S, 3, nt2).permute(0, 2, 1, 3).reshape(bs, 3, n_shards, S // n_shards, nt2)
        fused = fused.max(dim=3, keepdim=True)[0].repeat(1, 1, 1, S // n_shards, 1)
        fused = fused.reshape(bs, 3, S, nt2).permute(0, 2, 1, 3).reshape(N, -1, 2)
        fused = init_controls + fused                                              # diverse_fuse_type == "add"
        inp = torch.cat([feat_rows, rows.hl, rows.stlp, fused.reshape(N, nt2)], dim=-1)
    else:
        inp = torch.cat([feat_rows, rows.hl, rows.stlp, init_controls.reshape(N, nt2)], dim=-1)
    raw = torch.tanh(relu_mlp(sd, "rect_net", inp).reshape(N, -1, 2))
    wm, am = hp["mul_w_max"], hp["mul_a_max"]
    iw, ia = init_controls[..., 0], init_controls[..., 1]
    wpos = (raw[..., 0] >= 0).float()
    apos = (raw[..., 1] >= 0).float()
    dw = raw[..., 0] * (iw - (-wm)) * (1 - wpos) + raw[..., 0] * (wm - iw) * wpos
    da = raw[..., 1] * (ia - (-am)) * (1 - apos) + raw[..., 1] * (am - ia) * apos
    delta = torch.stack([dw, da], dim=-1)
    out = init_controls + delta * (scores < 0).float()[:, None, None]
    if clip_rect:
        out = torch.stack([torch.clip(out[..., 0], -wm, wm), torch.clip(out[..., 1], -am, am)], dim=-1)
    return out


# ------------------------------------------------------------------------------------------------
# A12 the whole timed region (reference nusc_train.py:957-1105)
# ------------------------------------------------------------------------------------------------
REFINEMENT_LIST_IDX = (0, 50, 80, 85, 90, 95, 98)    # k_d_list[8] of the reference (nusc_train.py:1052-1055)


def refinement(rows, controls, clist, iters=50, lr=0.3, thres=0.0005, record=None):
    """--refinement (reference nusc_train.py:1034-1071): 50 Adam iterations (lr 0.3) over per-row mixing weights
    softmax(lambda) of K = 8 control sequences -- the current controls and entries 0, 50, 80, 85, 90, 95, 98 of the
    rollout's list -- under loss = mask_mean(relu(5e-4 - score), valid); only rows whose current score is <= 0 (and whose
    lane is valid) are mixed.  Returns the controls of the LAST forward pass (the mixing weights after iters-1 steps)."""
    with torch.no_grad():
        _, score0, _ = rows.score(controls)
    violated = ((score0 <= 0) & (rows.valid > 0)).float().reshape(rows.N, 1, 1)
    lam = torch.ones(rows.N, 8, requires_grad=True)
    opt = torch.optim.Adam([lam], lr=lr)
    base = controls.detach()
    others = [clist[i].detach() for i in REFINEMENT_LIST_IDX]
    optim = base
    for it in range(iters):
        ratios = torch.softmax(lam, dim=-1)
        comb = [others[i] * ratios[..., i + 1:i + 2, None] for i in range(7)]
        optim = base * ratios[..., 0:1, None] + torch.sum(torch.stack(comb, dim=-1), dim=-1)
        optim = base * (1 - violated) + violated * optim
        _, sc, _ = rows.score(optim)
        loss = mask_mean(torch.relu(thres - sc), rows.valid)
        opt.zero_grad()
        loss.backward()
        if record is not None:
            record.append(lam.grad.detach().clone())
        opt.step()
    return optim.detach()


def sampling_region(sd, scene, S, steps, hp, x_T, z, rect_head=False, multi_cands=None, refinenet=True,
                    guidance=None, n_rolls=None, diverse=True, n_shards=4, clip_rect=False, use_rect=True,
                    refinement_iters=None):
    """use_rect=False is --not_use_rect: --rect_head's side effects (clip, full list) stay, the RefineNet block is skipped
    (nusc_train.py:993 `if args.rect_head and not args.not_use_rect`).  refinement_iters: --refinement (50 in the
    reference), inside the same block."""
    rows = Rows(scene, S, hp)
    out = {}
    with torch.no_grad():
        states, feature = rollout(sd, scene, rows, x_T, z, steps, guidance)
        clip = bool(rect_head)                                                     # nusc_train.py:1806-1809
        clist = [normalize_controls(x, hp["mul_w_max"], hp["mul_a_max"], clip) for x in states]
        out["controls_list"] = torch.stack(clist, dim=0)
        out["feature_scene"] = feature
        controls = clist[-1]
        if rect_head and use_rect:
            if multi_cands is not None:
                cands = torch.stack(clist[-multi_cands:], dim=0)
                _, cs, _ = rows.score(cands.reshape(-1, cands.shape[2], 2), reps=multi_cands)
                cs = cs.reshape(multi_cands, rows.N)
                controls, best, idx = select_candidates(cands, cs)
                out.update(cand_scores=cs, sel_scores=best, sel_idx=idx, sel_controls=controls)
            else:
                _, best, _ = rows.score(controls)
            if refinenet:
                controls = rect_forward(sd, feature, rows, controls, best, n_shards, diverse, clip_rect)
                out["rect_controls"] = controls
            for ri in range(n_rolls or 0):
                _, sc, _ = rows.score(controls)
                controls = rect_forward(sd, feature, rows, controls, sc, n_shards, diverse, clip_rect)
                out["roll%d_scores" % ri] = sc
                out["roll%d_controls" % ri] = controls
            if refinement_iters:
                with torch.enable_grad():
                    controls = refinement(rows, controls, clist, iters=refinement_iters)
                out["refinement_controls"] = controls
        s3, score, sig = rows.score(controls)
        acc, scene_acc = stl_metrics(score, rows.valid, S)
        out.update(final_controls=controls, final_scores3=s3, final_scores=score, final_acc=acc,
                   final_scene_acc=scene_acc, signals=sig,
                   final_trajs=unicycle_rollout(rows.s0, controls, hp["dt"]))
    return out


# ------------------------------------------------------------------------------------------------
# N1  training step of RefineNet under the STL loss (reference nusc_train.py:1400-1427, compute_policy_loss
#     :370-478 with rect_head and no diverse_loss, optimizer :1233,1522-1525)
# ------------------------------------------------------------------------------------------------
def dpp_diversity(rect, scores, bs, S, n_shards, hp, scale=1.0, detach=False):
    """DPP diversity term of e7 training (reference nusc_train.py:442-464): per (scene, mode, shard) group of S/n_shards
    samples, L = diag(q) exp(-scale*dist) diag(q) with q = exp(score)*[score>0] (or just [score>0] with
    --diverse_detach), diversity = tr(I - (L+I)^-1); returns mean(-diversity) and the per-group diversities."""
    n = S // n_shards
    G = bs * 3 * n_shards
    x = rect.reshape(bs, S, 3, -1).permute(0, 2, 1, 3).reshape(G, n, -1, 2)
    x = (x / torch.tensor([hp["mul_w_max"], hp["mul_a_max"]])).reshape(G, n, -1)
    quality = scores.reshape(bs, S, 3).permute(0, 2, 1).reshape(G, n)
    dist = torch.norm(x[:, :, None] - x[:, None, :], dim=-1)
    sim = torch.exp(-scale * dist)
    pos = (quality > 0).float()
    q = pos.detach() if detach else torch.exp(quality) * pos
    nL = torch.bmm(torch.bmm(torch.diag_embed(q), sim), torch.diag_embed(q))
    eye = torch.eye(n)[None]
    div = torch.einsum("bii->b", eye - torch.inverse(nL + eye))
    return torch.mean(-div), div


JOINT_PREFIXES = ("rect_net.", "ego_encoder.", "neighbor_encoder.", "lane_encoder.", "merge_net.")


def rect_train_step(sd, scene, S, hp, init_controls, prev_scores, lr, diverse=False, n_shards=4, e7=None, merge=None,
                    clip_rect=False, joint=False):
    """e7 is None: loss = mask_mean(relu(thres - score(rect_controls)), valid) (config 5; zero-weight regularisers).
    e7 = dict(stl_weight, diversity_weight, diversity_scale, rect_reg_loss, detach): the --diverse_loss objective
    loss_stl*stl_weight + loss_reg*rect_reg_loss + loss_diversity*diversity_weight (reference nusc_train.py:442-467),
    with the merge_net architecture unless merge=False (--no_arch, nusc_model.py:185).  Returns the loss, the gradients of the six rect_net tensors (the only parameters
    in the reference's optimiser without --joint, :1230-1233) and the tensors after one Adam step.
    joint=True (--joint, :1230-1231: Adam over net.parameters()): the scene feature keeps its graph, so the three scene
    encoders and merge_net (when the architecture uses it) receive gradients too; policy_net gets none -- the rollout
    runs under no_grad and the --rect_head losses (:455-467) leave loss_diffusion out -- and Adam skips it."""
    rows = Rows(scene, S, hp)
    use_merge = (diverse or e7 is not None) if merge is None else bool(merge)
    prefixes = tuple(p for p in JOINT_PREFIXES if use_merge or p != "merge_net.") if joint else ("rect_net.",)
    params = {k: _t(v).clone().requires_grad_() for k, v in sd.items() if k.startswith(prefixes)}
    sd_live = {k: (params[k] if k in params else _t(v)) for k, v in sd.items()}
    feature = encode_feat(sd_live, scene)
    if not joint:
        feature = feature.detach()
    init = _t(init_controls)
    rect = rect_forward(sd_live, feature, rows, init, _t(prev_scores), n_shards, use_merge, clip_rect=clip_rect)
    _, score, _ = rows.score(rect)
    loss_stl = mask_mean(torch.relu(hp["stl_nn_thres"] - score), rows.valid)
    extra = {}
    if e7 is None:
        loss = loss_stl
    else:
        ldiv, div = dpp_diversity(rect, score, rows.bs, S, n_shards, hp, e7.get("diversity_scale", 1.0), e7.get("detach", False))
        lreg = mask_mean(torch.square(rect - init.reshape(rect.shape).detach()), (score[:, None, None] >= 0).float())
        loss = loss_stl * e7["stl_weight"] + lreg * e7.get("rect_reg_loss", 0.0) + ldiv * e7["diversity_weight"]
        extra = dict(loss_diversity=(ldiv * e7["diversity_weight"]).detach(), loss_reg=lreg.detach(), diversity=div.detach())
    names = sorted(params)
    grads = torch.autograd.grad(loss, [params[k] for k in names])
    opt = torch.optim.Adam([params[k] for k in names], lr=lr)
    for k, g in zip(names, grads):
        params[k].grad = g
    opt.step()
    return dict(loss=loss.detach(), rect_controls=rect.detach(), scores=score.detach(),
                grads={k: g for k, g in zip(names, grads)}, after={k: params[k].detach() for k in names}, **extra)


# ------------------------------------------------------------------------------------------------
# N4  trajectory optimisation (reference nusc_train.py:1302-1325, compute_trajopt_loss_lite :287-300)
# ------------------------------------------------------------------------------------------------
def trajopt(rows, params, iters, lr, thres, reg_loss, checkpoints=()):
    """params (N,T,2) controls in physical units.  `iters` Adam iterations on
    mean(relu(thres - score)*valid)/clip(mean(valid),1e-3) + reg_loss*(mean(relu(w^2-wmax^2)) + mean(relu(a^2-amax^2))).
    Returns dict(params, scores_last (scores of the iterate the last step started from), losses [(total, stl, reg, acc)],
    grad0, checkpoints {k: params after k iterations})."""
    hp = rows.hp
    p = params.detach().clone().requires_grad_()
    opt = torch.optim.Adam([p], lr=lr)
    losses, ck, grad0, scores = [], {}, None, None
    for ii in range(iters):
        _, scores, _ = rows.score(p)
        dense = torch.mean(torch.relu(thres - scores) * rows.valid) / torch.clip(torch.mean(rows.valid), 1e-3)
        reg = (torch.mean(torch.relu(p[..., 0] ** 2 - hp["mul_w_max"] ** 2))
               + torch.mean(torch.relu(p[..., 1] ** 2 - hp["mul_a_max"] ** 2))) * reg_loss
        loss = dense + reg
        acc = torch.mean((scores >= 0).float() * rows.valid) / torch.clip(torch.mean(rows.valid), 1e-3)
        opt.zero_grad()
        loss.backward()
        if ii == 0:
            grad0 = p.grad.detach().clone()
        opt.step()
        losses.append(tuple(float(v.detach()) for v in (loss, dense, reg, acc)))
        if ii + 1 in checkpoints:
            ck[ii + 1] = p.detach().clone()
    return dict(params=p.detach(), scores_last=scores.detach(), losses=losses, grad0=grad0, checkpoints=ck)
