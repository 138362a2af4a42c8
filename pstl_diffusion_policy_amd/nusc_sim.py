"""Closed-loop (receding-horizon) caller of the sampling path on a synthetic world.

Reference: nusc_sim.py:388-755.  Per simulation step the reference builds a ONE-scene batch (192 rows = sampling_size 64
x 3 modes), overwrites the STL parameters with fixed values (:467-472), runs `diffusion_rollout(..., maximize=True)`
with guidance (:481), candidate selection + RefineNet (:518-536), picks the best-scoring lane-keeping sample
(`scores_all[:, 1:3] = -10000; argmax`, :676-681), applies its first control with the unicycle model (:118) and asks the
nuScenes devkit for the next observation.  The devkit world is out of scope; here the world is a seeded synthetic
three-lane road with constant-velocity neighbours (synthetic.py), which is enough to exercise -- and time -- the
small-batch, latency-bound use of the same kernels (the chain kernel then runs 4 tiles per workgroup, see
tiles_per_group in csrc/mlp_kernels.hip).
"""
import math
import time

import torch

from .engine import GraphCapture, PackedWeights, Sampler, SceneBatch
from .synthetic import default_hparams, make_scene_batch

FIXED_STLP = (1.0, 9.0, -3.0, 2.0, 0.1, 0.2)   # vmin, vmax, dmin, dmax, dsafe, thmax (reference nusc_sim.py:467-472)


class SyntheticWorld:
    """One ego vehicle on a three-lane road with K constant-velocity neighbours; unicycle dynamics (nusc_train.py:29-37)."""

    def __init__(self, K=8, seed=0, dt=0.5, nt=20):
        self.K, self.dt, self.nt = K, dt, nt
        base = make_scene_batch(1, K=K, nt=nt, S=1, seed=seed, dt=dt, random_pose=False, curved=False, stlp_mode="fixed")
        self.lanes = {k: base["%slane_wpts" % k].clone() for k in ("curr", "left", "right")}
        self.state = base["ego_traj"][0, 0, :4].clone()                  # x, y, th, v
        self.nei0 = base["neighbors"][0].clone()                          # (K,7) valid,x,y,th,v,L,W
        self.t = 0

    def observation(self):
        """The one-scene batch the sampling path consumes (schema of SURVEY 3.0)."""
        tt = torch.arange(self.nt).float() * self.dt
        x, y, th, v = self.state.tolist()
        ego = torch.stack([x + v * tt * math.cos(th), y + v * tt * math.sin(th), torch.full_like(tt, th),
                           torch.full_like(tt, v), torch.full_like(tt, 4.084), torch.full_like(tt, 1.73)], dim=-1)[None]
        n = self.nei0
        adv = (self.t * self.dt + tt)[None, :]                            # neighbours keep their velocity
        nx = n[:, 1:2] + n[:, 4:5] * adv * torch.cos(n[:, 3:4])
        ny = n[:, 2:3] + n[:, 4:5] * adv * torch.sin(n[:, 3:4])
        traj = torch.stack([n[:, 0:1].expand(-1, self.nt), nx, ny, n[:, 3:4].expand(-1, self.nt),
                            n[:, 4:5].expand(-1, self.nt), n[:, 5:6].expand(-1, self.nt), n[:, 6:7].expand(-1, self.nt)], dim=-1)
        traj = traj * n[:, 0:1, None]
        # lanes re-anchored around the ego's longitudinal position, as the dataset does per sample
        shift = torch.tensor([x, 0.0, 0.0])
        obs = {"ego_traj": ego, "neighbors_traj": traj[None], "neighbors": traj[None, :, 0, :].contiguous(),
               "curr_id": torch.ones(1, 1), "left_id": torch.ones(1, 1), "right_id": torch.ones(1, 1),
               "stlp_modes": torch.tensor(FIXED_STLP).reshape(1, 1, 6).repeat(1, 3, 1)}
        for k in ("curr", "left", "right"):
            obs["%slane_wpts" % k] = self.lanes[k] + shift
        return obs

    def step(self, control):
        w, a = float(control[0]), float(control[1])
        x, y, th, v = self.state.tolist()
        self.state = torch.tensor([x + v * math.cos(th) * self.dt, y + v * math.sin(th) * self.dt, th + w * self.dt,
                                   v + a * self.dt])
        self.t += 1

    def min_clearance(self):
        """Centre distance to the closest valid neighbour now (coarse collision indicator for the report)."""
        n = self.nei0
        adv = self.t * self.dt
        nx = n[:, 1] + n[:, 4] * adv * torch.cos(n[:, 3])
        ny = n[:, 2] + n[:, 4] * adv * torch.sin(n[:, 3])
        d = torch.sqrt((nx - self.state[0]) ** 2 + (ny - self.state[1]) ** 2)
        d = torch.where(n[:, 0] > 0.5, d, torch.full_like(d, 1e9))
        return float(d.min())


_SCENE_KEYS = ("ego_traj", "neighbors", "neighbors_traj", "currlane_wpts", "leftlane_wpts", "rightlane_wpts", "curr_id",
               "left_id", "right_id", "stlp_modes")


def _plan_device(sm, scene, S, hp, dev, diffusion_steps, multi_cands, g, seed, dyn=None):
    """One planning step: the sampling region on the one-scene batch, then the best lane-keeping sample (the reference sets the
    other two modes' scores to -10000 before its argmax, nusc_sim.py:677-683).  Returns a device tensor (first control (2), its
    score, the bits of the chain-domain status word): everything the loop needs comes back in ONE 16-byte copy."""
    sb = SceneBatch(scene, S, hp, dev, dyn=dyn, scale_in_dyn=dyn is not None)
    out = sm.sampling_region(sb, diffusion_steps, None, None, rect_head=True, multi_cands=multi_cands, guidance=g,
                             seed=seed, want_scores3=False, want_counts=False)
    return sm.select_plan(sb, out["final_scores"], out["final_controls"])


def _plan(sm, obs, S, hp, dev, diffusion_steps, multi_cands, g, seed):
    return _plan_device(sm, obs, S, hp, dev, diffusion_steps, multi_cands, g, seed).cpu()


class GraphPlanner:
    """The ~45 launches of one planning step captured ONCE in a HIP graph and replayed per simulation step.

    What changes between simulation steps lives in one device buffer: the observation (every scene tensor, 16-byte slots) and
    a pstl_dyn block (the noise seed and the guidance-loss scale, computed on the host from the lane ids).  The kernels read seed and scale from that block (cfg.dyn, ABI 4) instead of taking them by value, so a replay with
    new inputs is: fill a pinned host mirror, ONE host-to-device copy, graph launch, ONE 16-byte copy back.  Same kernels, same
    arguments, same order as the eager path: bit-identical results (tested)."""

    def __init__(self, sm, obs, S, hp, dev, diffusion_steps, multi_cands, g):
        self.sm, self.S, self.hp, self.dev = sm, S, hp, dev
        self.args = (diffusion_steps, multi_cands, g)
        self.layout, o = [], 0
        for k in _SCENE_KEYS:
            t = torch.as_tensor(obs[k])
            self.layout.append((k, tuple(t.shape), o, t.numel()))
            o += t.numel() + (-t.numel() % 4)
        self.n_scene = o
        self.host = torch.zeros(o + 4, dtype=torch.float32).pin_memory()
        self.host_np = self.host.numpy()
        self.inp = torch.zeros(o + 4, dtype=torch.float32, device=dev)
        self.graph = None
        self.capture(obs)

    def _fill(self, obs, seed):
        for k, shape, o, n in self.layout:
            t = torch.as_tensor(obs[k])
            if tuple(t.shape) != shape:
                raise ValueError("observation tensor %s changed its shape (%s, captured %s)" % (k, tuple(t.shape), shape))
            self.host[o:o + n] = t.reshape(-1).to(torch.float32)
        seed = int(seed) & (2 ** 64 - 1)
        self.host_np[self.n_scene:self.n_scene + 2].view("uint32")[:] = (seed & 0xffffffff, seed >> 32)
        # the guidance-loss scale from the host copy of the lane ids (what SceneBatch computes for an eager step)
        vsum = float(sum(torch.as_tensor(obs[k]).to(torch.float32).sum() for k in ("curr_id", "left_id", "right_id"))) * self.S
        self.host_np[self.n_scene + 2] = SceneBatch.loss_scale(vsum, 3 * self.S * int(torch.as_tensor(obs["curr_id"]).shape[0]))

    def _body(self):
        scene = {k: self.inp[o:o + n].reshape(shape) for k, shape, o, n in self.layout}
        steps, mc, g = self.args
        return _plan_device(self.sm, scene, self.S, self.hp, self.dev, steps, mc, g, 0, dyn=self.inp[self.n_scene:])

    def capture(self, obs):
        """(Re-)captures the graph with the sampler's current arithmetic (after a domain fallback: the exact-fp32 kernels)."""
        self._fill(obs, 0)
        self.inp.copy_(self.host)
        self.graph = GraphCapture(self._body)

    def plan(self, obs, seed):
        self._fill(obs, seed)
        self.inp.copy_(self.host, non_blocking=True)
        return self.graph.replay().cpu()


def closed_loop(state_dict, n_sim_steps=20, K=8, S=64, diffusion_steps=100, multi_cands=5, guidance=True, guidance_before=10,
                guidance_lr=0.04, seed=0, device="cuda:0", verbose=True, chain_waves=None, graph=True):
    """Runs the receding-horizon loop; returns per-step records (latency in seconds with device sync, score, state).
    graph: replay one captured HIP graph per simulation step (GraphPlanner) instead of ~45 eager launches; same results.
    chain_waves: arithmetic of the MLP chains (None: PSTL_CHAIN_WAVES or the default split-f16 form; a step that leaves its
    domain is planned again on the exact-fp32 kernels, with a RuntimeWarning, and the loop stays on them)."""
    dev = torch.device(device)
    hp = default_hparams()
    sm = Sampler(PackedWeights(state_dict, dev), hp, chain_waves=chain_waves)
    world = SyntheticWorld(K=K, seed=seed, dt=hp["dt"], nt=hp["nt"])
    g = dict(enabled=True, before=guidance_before, niters=1, lr=guidance_lr, maximize=True) if guidance else None
    records = []
    planner = GraphPlanner(sm, world.observation(), S, hp, dev, diffusion_steps, multi_cands, g) if graph else None
    if planner is not None:
        plan = planner.plan
        sm.w.chain_overflowed(clear=True)      # (the capture's warm-up ran on a zero seed: whatever it flagged is re-detected)
    else:
        plan = lambda obs, sd: _plan(sm, obs, S, hp, dev, diffusion_steps, multi_cands, g, sd)
    for it in range(n_sim_steps):
        obs = world.observation()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pick = plan(obs, seed * 100003 + it)
        # The split-f16 domain flag rides in the same copy as the control (bit pattern of the packed buffer's status word 2):
        # a layer input beyond the half range leaves plausible garbage, not NaNs, and this loop would drive the vehicle with it.
        # The step is then planned again on the exact-fp32 kernels (same Philox seed), and so is every later one.
        if pick[3:4].view(torch.int32).item() != 0 and sm.check_chain_domain():
            if planner is not None:
                planner.capture(obs)           # the graph holds the old arithmetic's launches: captured again
            pick = plan(obs, seed * 100003 + it)
        ctrl = pick[:2]
        lat = time.perf_counter() - t0
        world.step(ctrl)
        rec = dict(step=it, latency_s=lat, best_score=float(pick[2]), x=float(world.state[0]),
                   y=float(world.state[1]), v=float(world.state[3]), clearance=world.min_clearance())
        records.append(rec)
        if verbose:
            print("sim %02d latency %.2f ms score %.3f x %.1f y %.2f v %.2f clearance %.1f" % (
                it, lat * 1e3, rec["best_score"], rec["x"], rec["y"], rec["v"], rec["clearance"]))
    return records


def main(argv=None):
    """The reference's nusc_sim command lines (README: `nusc_sim.py -e e7_ours --diffusion ... --test -P e7_ours
    --filter_traj 0 --test_scenes --viz_last [--guidance --guidance_before 10 --guidance_niters 1 --guidance_lr 0.04]
    --suffix sim`) on the synthetic world: same flags (nusc_train's parser), a checkpoint from -P when it exists,
    random-init weights under --seed otherwise.  The devkit replay / rendering flags are accepted and ignored."""
    import os
    from . import nusc_train as nt
    from .nusc_model import init_state_dict
    args = nt.generate_parser(argv)
    sd = init_state_dict(args.seed, rect_head=True, diverse_loss=True)
    if args.net_pretrained_path:
        path = args.net_pretrained_path
        if not os.path.isfile(path):
            path = os.path.join("exps", path, "models", "model_last.ckpt")
        if os.path.isfile(path):
            sd.update(torch.load(path, map_location="cpu"))
        elif args.allow_random_init:
            print("checkpoint %s not found: random-init weights (seed %d)" % (path, args.seed))
        else:
            raise SystemExit("checkpoint %s not found (pass --allow_random_init to run with random-init weights)" % path)
    recs = closed_loop(sd, n_sim_steps=max(args.n_trials, 1) if args.n_trials < 100 else 20, K=args.n_neighbors,
                       S=args.n_randoms, diffusion_steps=args.diffusion_steps, multi_cands=args.multi_cands or 5,
                       guidance=args.guidance, guidance_before=args.guidance_before, guidance_lr=args.guidance_lr,
                       seed=args.seed)
    lats = sorted(r["latency_s"] for r in recs[min(2, len(recs) - 1):])
    print("median latency per simulation step: %.2f ms (%d rows, %d diffusion steps, guidance %s)"
          % (lats[len(lats) // 2] * 1e3, args.n_randoms * 3, args.diffusion_steps, "on" if args.guidance else "off"))
    return recs


if __name__ == "__main__":
    main()
