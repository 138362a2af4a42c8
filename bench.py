#!/usr/bin/env python3
"""bench.py -- throughput of the DDPM-sampling + STL hot path on MI355X (contract: see the round prompt).

The line also carries (single-GPU runs; skipped with --no_extras): "roofline.fp32_exact" (the same step on the exact-fp32
MFMA chains), "also" (BASELINE configs 2, 3 and 5 -- e5, e7, e8_train -- three timed steps each, each with its own dominant-kernel
roofline fraction; e8_train with the HBM rate of RefineNet's backward), "sweep" (the headline workload at 192 ... 786 432 rows)
and "paper_metric" (md["time"] of the reference's two README command lines through the CLI mirror).

A "step" is one pass of the timed region of the reference's sampling harness (nusc_train.py:957-1105) over one
synthetic batch already resident in HBM: row constants, scene preparation, scene encoder, noise generation,
`diffusion_steps-1` denoiser evaluations (+ STL guidance on the last `guidance_before` steps), candidate scoring +
selection, RefineNet, final STL scoring, the satisfaction counts and the diversity metrics (std, hull volume, entropies,
occupancy area, ADE/FDE) (+ one RCCL all-gather of 20 eight-byte words -- counters and diversity totals -- when N > 1).
Unit: sampled trajectories per second = rows (scenes x sampling_size x 3 modes) / wall time, whole job.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_STEP_MIN = 2 * (40 * 256 + 256 * 256 + 256 * 40)   # algorithmic FLOP per row per denoiser evaluation
PEAK_FP32_MATRIX_TFLOPS = 157.3                       # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 16 * 157.3                  # same guide: the f32 MFMA rate is 1/16 of the dense BF16 rate (~2.5 PF)
# split-bf16 chain kernel: MFMA FLOP issued per row and denoiser evaluation (69 v_mfma_f32_16x16x32_bf16 per wave and
# 16-row tile, 8 waves): 3 bf16 products per fp32 product, layer 1 padded 40 -> 64 columns, layer 3 padded 40 -> 48 rows
F_STEP_ISSUED_BF16 = 8 * 69 * (2 * 16 * 16 * 32) / 16


_T0 = time.time()


def log(msg):
    """Progress on stderr (the driver keeps stdout for the ONE JSON line): which phase a slow or stuck run is in."""
    print("bench [%6.1f s] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="e7_guid", choices=["e5", "e7", "e7_guid", "e8_train", "e7_train", "trajopt"])
    p.add_argument("--joint", action="store_true", help="train workloads: Adam over the whole net (reference --joint)")
    p.add_argument("--scenes", type=int, default=4096, help="scenes per GPU (weak scaling)")
    p.add_argument("--sampling_size", type=int, default=64)
    p.add_argument("--neighbors", type=int, default=2)
    p.add_argument("--diffusion_steps", type=int, default=50)
    p.add_argument("--multi_cands", type=int, default=5)
    p.add_argument("--chain_waves", type=int, default=0)
    p.add_argument("--noise", default="kernel", choices=["kernel", "torch"],
                   help="kernel: Philox noise drawn inside the HIP kernels; torch: torch.randn tensors (parity mode)")
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="weak: --scenes per GPU; strong: --total_scenes split over the GPUs in contiguous blocks (BASELINE config 4 "
                        "as written: 32 768 scenes over 8 GPUs)")
    p.add_argument("--total_scenes", type=int, default=32768, help="--scaling strong: scenes of the whole job")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--cpu_scenes", type=int, default=32, help="scenes of the cpu_baseline TIMING leg (32 = the size BASELINE.md probed)")
    p.add_argument("--cpu_parity_scenes", type=int, default=12, help="scenes of the same-inputs parity leg (HIP against the oracle)")
    p.add_argument("--no_extras", action="store_true", help="skip the fp32-exact leg, the 'also' workloads and the batch-size sweep")
    p.add_argument("--no_graph", action="store_true", help="small batches: eager launches instead of one HIP-graph replay per step")
    p.add_argument("--trajopt_iters", type=int, default=50, help="Adam iterations per step of the trajopt workload")
    return p.parse_args()


class _AdamTap:
    """Records the gradient every torch.optim.Adam.step() consumes while active (the oracle's guidance block builds a fresh Adam
    per guided step, as the reference does: nusc_train.py:606-623) -- what the guided-outlier gate below is checked against."""

    def __enter__(self):
        self.grads = []
        self._orig = torch.optim.Adam.step
        tap, orig = self.grads, self._orig

        def tapped(opt, *args, **kw):
            for grp in opt.param_groups:
                for p in grp["params"]:
                    if p.grad is not None:
                        tap.append(p.grad.detach().clone())
            return orig(opt, *args, **kw)

        torch.optim.Adam.step = tapped
        return self

    def __exit__(self, *exc):
        torch.optim.Adam.step = self._orig


def guided_gate(err_list, adam_grads, steps, guidance, tol=1e-4, g_eps=1e-6):
    """The gate the -m gpu tests apply to a guided run (tests/conftest.py: guided_outlier_rows), inside the driver's own run.
    Adam's normalised step lr g / (|g| + 1e-8) is discontinuous as g -> 0: where the oracle's own STL gradient of an element is
    a near-cancellation below g_eps, a float32-level difference in g moves the control by up to lr, and no two float32
    implementations agree.  Rows in which an element of the per-step list leaves `tol` are counted and excluded -- and the
    explanation is CHECKED: the first step at which such a row leaves `tol` must be a guided step, and every element that left
    it there must have |g| < g_eps in one of that step's Adam iterations.  err_list: (steps, N, 40) |HIP - oracle| of the
    normalised per-step list; adam_grads: the oracle's recorded gradients, one (N,20,2) tensor per Adam.step()."""
    from pstl_diffusion_policy_amd.engine import guidance_triggered
    N = err_list.shape[1]
    guided = [i for i in range(steps - 1, 0, -1) if guidance_triggered(i, steps, guidance)]
    nit = int(guidance["niters"])
    g = torch.stack([x.reshape(N, 40).abs() for x in adam_grads]).reshape(len(guided), nit, N, 40)
    bad = err_list > tol
    bad_rows = bad.any(dim=2).any(dim=0)
    explained = True
    for r in torch.nonzero(bad_rows).flatten().tolist():
        k0 = int(torch.nonzero(bad[:, r].any(dim=1)).flatten()[0])      # list entry k = the state after reverse step steps - k
        i0 = steps - k0
        if i0 not in guided:
            explained = False
            continue
        gmin = g[guided.index(i0), :, r].min(dim=0).values
        if not bool((gmin[bad[k0, r]] < g_eps).all()):
            explained = False
    return bad_rows, explained, int(((g < g_eps) & (g > 0)).sum())


def cpu_timing_run(a, hp, sd, guidance, rect_head):
    """One timed pass of the CPU oracle over `--cpu_scenes` scenes of the workload (called in a child process of its own by
    cpu_baseline, with the thread count in OMP_NUM_THREADS)."""
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    S, steps = a.sampling_size, a.diffusion_steps
    scene = {k: v.numpy() for k, v in make_scene_batch(a.cpu_scenes, K=a.neighbors, S=S, seed=78, stlp_mode="wide").items()}
    N = a.cpu_scenes * S * 3
    sdn = {k: v.cpu().numpy() for k, v in sd.items()}
    g = torch.Generator().manual_seed(5)
    t0 = time.time()
    x_T = torch.randn(N, 40, generator=g)
    z = torch.randn(steps - 1, N, 40, generator=g)
    ref = orc.sampling_region(sdn, scene, S, steps, hp, x_T, z, rect_head=rect_head, multi_cands=a.multi_cands if rect_head else None,
                              guidance=guidance)
    dt = time.time() - t0
    return {"seconds": dt, "trajectories_per_s": N / dt, "stl_sat_rate": float(ref["final_acc"]), "torch_threads": torch.get_num_threads()}


def cpu_baseline(a, hp, sd, guidance, rect_head, sampler=None, dev=None, sampler_exact=None):
    """The CPU oracle (a float32 torch restatement of the reference path; the reference itself cannot travel to the GPU box)
    on the host cores, in two legs:
      * timing: `--cpu_scenes` (32: the size BASELINE.md probed) scenes of the same workload, once per thread count in
        {8, 32, all cores} -- the best is the reported value, with the thread count that gave it;
      * parity: `--cpu_parity_scenes` (12) scenes through the oracle and, on the same scenes and noise, through the HIP path
        (both arithmetics), with the guided-outlier gate of the -m gpu tests applied to the comparison."""
    from oracle import pstl_oracle as orc
    from pstl_diffusion_policy_amd.synthetic import make_scene_batch
    S, steps = a.sampling_size, a.diffusion_steps
    sdn = {k: v.cpu().numpy() for k, v in sd.items()}
    mc = a.multi_cands if rect_head else None
    ncpu = os.cpu_count() or 1
    threads0 = torch.get_num_threads()

    def run(bs, seed, tap=False):
        scene_t = make_scene_batch(bs, K=a.neighbors, S=S, seed=seed, stlp_mode="wide")
        scene = {k: v.numpy() for k, v in scene_t.items()}
        N = bs * S * 3
        g = torch.Generator().manual_seed(5)
        t0 = time.time()
        x_T = torch.randn(N, 40, generator=g)
        z = torch.randn(steps - 1, N, 40, generator=g)
        if tap:
            with _AdamTap() as rec:
                ref = orc.sampling_region(sdn, scene, S, steps, hp, x_T, z, rect_head=rect_head, multi_cands=mc, guidance=guidance)
            grads = rec.grads
        else:
            ref, grads = orc.sampling_region(sdn, scene, S, steps, hp, x_T, z, rect_head=rect_head, multi_cands=mc, guidance=guidance), None
        return scene_t, N, x_T, z, ref, time.time() - t0, grads

    # ---- timing leg: one CHILD process per thread count (CPU only: it never touches the GPU), each under a wall-clock limit --
    # an oversubscribed shared host (the pool's boxes: 128 cores, other tenants) has been seen to stretch a 5 s run to many
    # minutes, and the driver's bench run must end within a few
    sweep = []
    child = ("import sys, json, time, torch\n"
             "sys.path.insert(0, %r)\n"
             "import bench\n"
             "from pstl_diffusion_policy_amd.nusc_model import init_state_dict\n"
             "from pstl_diffusion_policy_amd.synthetic import default_hparams\n"
             "a = bench.parse()\n"
             "print(json.dumps(bench.cpu_timing_run(a, default_hparams(), init_state_dict(1007), json.loads(%r), %r)))\n")
    limit = float(os.environ.get("PSTL_CPU_BASELINE_LIMIT_S", "150"))
    for nt in sorted({min(8, ncpu), min(32, ncpu), ncpu}):
        done = [r for r in sweep if "seconds" in r]
        if len(done) >= 2 and done[-1]["trajectories_per_s"] < done[-2]["trajectories_per_s"]:
            # more threads already ran slower (the oracle's tensors are small: 6 144 rows): the all-cores run would only add
            # minutes of oversubscribed host time (round 5: 38 s for 2 304 rows on 128 threads)
            sweep.append({"threads": nt, "skipped": "the rate fell from %d to %d threads" % (done[-2]["threads"], done[-1]["threads"])})
            continue
        env = dict(os.environ, OMP_NUM_THREADS=str(nt), MKL_NUM_THREADS=str(nt), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        log("cpu_baseline: timing the oracle on %d thread(s), limit %.0f s" % (nt, limit))
        try:
            r = subprocess.run([sys.executable, "-c", child % (ROOT, json.dumps(guidance), bool(rect_head))] + sys.argv[1:],
                               env=env, capture_output=True, text=True, timeout=limit)
            rec = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"failed": r.stderr[-300:]}
        except subprocess.TimeoutExpired:
            rec = {"timed_out_after_s": limit}
        rec["threads"] = nt
        sweep.append(rec)
    done = [r for r in sweep if "seconds" in r]
    if not done:
        return {"value": None, "unit": "trajectories/s", "cores": None, "kind": "port", "thread_sweep": sweep, "host_cores": ncpu,
                "sample": "no timing run of the oracle finished within %.0f s on this host" % limit}
    best = max(done, key=lambda r: r["trajectories_per_s"])
    N_t = a.cpu_scenes * S * 3
    out = {"value": best["trajectories_per_s"], "unit": "trajectories/s", "cores": best["threads"], "kind": "port",
           "sample": "%s, %d scenes x %d samples x 3 modes = %d rows, %d diffusion steps, K=%d, %.1f s on %d of the host's %d "
                     "cores (best of the thread counts tried), torch %s CPU"
                     % (a.workload, a.cpu_scenes, S, N_t, steps, a.neighbors, best["seconds"], best["threads"], ncpu, torch.__version__),
           "thread_sweep": sweep, "host_cores": ncpu, "stl_sat_rate": best["stl_sat_rate"],
           "note": "BASELINE.md section 3 probed the reference ITSELF at this batch size on 8 threads: ~560 trajectories/s for e7 + "
                   "guidance at 100 steps and K = 8, i.e. twice the denoiser evaluations per trajectory of this workload (%d steps, "
                   "K = %d)" % (steps, a.neighbors)}
    # ---- parity leg (the thread count that won the timing leg)
    torch.set_num_threads(best["threads"])
    log("cpu_baseline: parity leg, %d scenes through the oracle and the HIP path" % a.cpu_parity_scenes)
    if sampler is not None and best["seconds"] > 90.0:
        out["gpu_same_inputs"] = {"skipped": "the host took %.0f s for the timing run: the in-process oracle pass of the parity leg "
                                             "cannot be bounded, and the driver's run must end" % best["seconds"]}
    elif sampler is not None:
        from pstl_diffusion_policy_amd.engine import SceneBatch, acc_from_counts
        scene_t, N, x_T, z, ref, _, grads = run(a.cpu_parity_scenes, 77, tap=bool(guidance))
        sb = SceneBatch(scene_t, S, hp, dev)

        def same_inputs(sm):
            got = sm.sampling_region(sb, steps, x_T.to(dev), z.to(dev), rect_head=rect_head, multi_cands=mc, guidance=guidance,
                                     want_scores3=False, full_list=True)
            torch.cuda.synchronize()
            acc, _ = acc_from_counts(got["counts"])
            err = (got["final_controls"].reshape(N, 20, 2).cpu() - ref["final_controls"]).abs().reshape(N, 40)
            rec = {"chain_waves": sm.chain_waves, "rows": N, "stl_sat_rate": acc, "oracle_stl_sat_rate": float(ref["final_acc"]),
                   "max_abs_dcontrols_all_rows": float(err.max()),
                   "masks_differing": int(((got["final_scores"].cpu() > 0) != (ref["final_scores"] > 0)).sum())}
            if guidance:
                err_list = (got["controls_list"].cpu().reshape(steps, N, 40) - ref["controls_list"].reshape(steps, N, 40)).abs()
                bad_rows, explained, tiny = guided_gate(err_list, grads, steps, guidance)
                keep = ~bad_rows
                # rows that share a merge_net max-pool group with an excluded row see its output through RefineNet
                if rect_head and bad_rows.any() and S % int(hp.get("n_shards", 4)) == 0:
                    nsh = int(hp.get("n_shards", 4))
                    grp = bad_rows.reshape(-1, nsh, S // nsh, 3).any(dim=2, keepdim=True)
                    keep = ~grp.expand(-1, nsh, S // nsh, 3).reshape(N)
                rec.update(rows_excluded=int(bad_rows.sum()), rows_excluded_with_their_pool_groups=int((~keep).sum()),
                           outliers_all_in_adam_eps_regime=bool(explained), adam_eps_regime_elements=tiny,
                           max_abs_dcontrols_kept_rows=float(err[keep].max()) if keep.any() else 0.0,
                           frac_controls_within_1e_4_kept_rows=float((err[keep] <= 1e-4).float().mean()) if keep.any() else 1.0,
                           masks_differing_kept_rows=int((((got["final_scores"].cpu() > 0) != (ref["final_scores"] > 0)) & keep).sum()),
                           gate="rows with a per-step list element > 1e-4 are excluded; the first such step must be a guided "
                                "step and each such element must have an oracle |dL/du| < 1e-6 in one of that step's Adam "
                                "iterations (tests/conftest.py: guided_outlier_rows)")
            else:
                rec.update(frac_controls_within_1e_4=float((err <= 1e-4).float().mean()))
            return rec

        out.update(gpu_same_inputs=same_inputs(sampler))
        if sampler_exact is not None:     # the same inputs through the exact-fp32 MFMA kernels: the second opinion
            out.update(gpu_same_inputs_fp32_exact=same_inputs(sampler_exact))
    torch.set_num_threads(threads0)
    return out


def count_gpus():
    """Number of GPUs this process may use, WITHOUT touching torch's CUDA module or the HIP runtime (the parent of the rank
    processes must stay provably GPU-free): the KFD topology in sysfs (nodes with SIMDs are GPUs), narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one of them is set."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    n = None
    if os.path.isdir(base):
        n = 0
        for node in os.listdir(base):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                # a container may see the whole topology but only some of the devices: a GPU counts when its render node is
                # there and usable (drm_render_minor; older KFDs do not list it: counted)
                minor = int(props.get("drm_render_minor", "0"))
                dev = "/dev/dri/renderD%d" % minor
                if minor > 0 and os.path.isdir("/dev/dri") and not (os.path.exists(dev) and os.access(dev, os.R_OK | os.W_OK)):
                    continue
                n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([x for x in v.split(",") if x.strip() != ""])
            n = listed if n is None else min(n, listed)
    if n is None:       # no KFD sysfs (not a ROCm box): the last resort does not initialise the GPU either on this image
        n = torch.cuda.device_count()
    return n


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: this process starts the N ranks itself (one per GPU, rendezvous on
    127.0.0.1) BEFORE anything here touches the GPU, waits for them and exits with their status.  Rank 0 prints the JSON
    line on the inherited stdout.  Children are separate processes started with Popen -- never an exec of this one."""
    n = a.gpus
    backend = os.environ.get("PSTL_BENCH_BACKEND", "nccl")
    ndev = count_gpus()                        # sysfs + environment only: this process never touches the GPU
    if backend == "nccl" and ndev < n:
        print("bench: --gpus %d but only %d GPU(s) visible; refusing to measure fewer ranks than asked for" % (n, ndev),
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:               # a rank died: its peers would wait in a collective for ever
                    q.terminate()
    return rc


# algorithmic HBM bytes per row of RefineNet's backward (pstl_refine_backward): what must be read once -- the two saved
# activations h1, h2 (2 x 256 floats), dcontrols / pre / init (3 x 40), hl | stlp (7) -- the 145 704 gradients it writes are noise
BWD_BYTES_PER_ROW = 4 * (2 * 256 + 3 * 40 + 7)
GRAPH_MAX_ROWS = 98304      # batches up to this many rows per GPU replay a captured HIP graph per step (see Job)


def chain_layout(cfg):
    """Which kernel and layout pstl_rollout picks for the multi-step launches of a batch: asked of the library
    (pstl_rollout_layout), not re-derived here."""
    from pstl_diffusion_policy_amd import ffi
    kern, g, rounds = ffi.rollout_layout(cfg)
    return ["k_chain, latency layout (%d tile(s) per workgroup, empty pipeline slots)", "k_chain, throughput layout (%d tiles per "
            "workgroup)", "k_chain2 (row-stationary: %%d tiles = %d rows per workgroup)" % (16 * g), "k_chain, exact arithmetic (%d "
            "tiles per workgroup)"][kern] % g + ", %d round(s) of workgroups" % rounds


class Job:
    """One workload on one synthetic scene shard, resident in HBM: step() enqueues one pass of the timed region."""

    def __init__(self, a, workload, bs, dev, rank, world, cpu_group, weights, hp, sd, chain_waves, joint=False, plan_rows=0,
                 row_offset=None):
        from pstl_diffusion_policy_amd.engine import Sampler, diffusion_coeffs
        from pstl_diffusion_policy_amd.shard import device_identity
        from pstl_diffusion_policy_amd.synthetic import make_scene_batch
        self.a, self.workload, self.bs, self.dev, self.rank, self.world, self.cpu_group = a, workload, bs, dev, rank, world, cpu_group
        # plan_rows: the rows of the job's largest shard -- every rank hands the library the same number, so that the default
        # arithmetic picks the same denoiser kernel on every shard (pstl_cfg.plan_rows); row_offset: global index of this
        # shard's first row (in-kernel noise is keyed by the global row)
        self.plan_rows = int(plan_rows)
        self.ident, self.ident_text = device_identity(dev)
        self.seen = {}
        self.exchange_s, self.exchange_n = 0.0, 0
        self.hp, self.sd, self.chain_waves, self.joint = hp, sd, chain_waves, joint
        self.rect_head = workload != "e5"
        self.guidance = dict(enabled=True, before=10, niters=1, lr=0.01) if workload == "e7_guid" else None
        self.train = workload in ("e8_train", "e7_train")   # one optimisation step of RefineNet: config 5 / e7
        self.e7 = dict(stl_weight=0.0, diversity_weight=1.0) if workload == "e7_train" else None
        self.trajopt = workload == "trajopt"    # N4: the data-augmentation loop, trajopt_iters Adam iterations per step
        self.S, self.steps = a.sampling_size, a.diffusion_steps
        # every rank owns its own contiguous block of scenes (seeded by the global scene offset); no data-path collective
        scene = make_scene_batch(bs, K=a.neighbors, S=self.S, seed=1000 + rank, invalid_lane_frac=0.2, stlp_mode="wide")
        self.ids_host = float(sum(scene[k].sum().item() for k in ("curr_id", "left_id", "right_id")))   # from the CPU copy, as a
        self.scene = {k: v.to(dev) for k, v in scene.items() if k not in ("pre_stlp", "tj_scores_prior")}   # data loader would
        self.sampler = Sampler(weights, hp, chain_waves=chain_waves)
        if self.train:
            from pstl_diffusion_policy_amd.engine import RectTrainer
            self.tnames = RectTrainer.joint_names(self.e7 is not None) if joint else RectTrainer.NAMES   # --joint: encoders (+ merge_net) too
            self.tparams = {k: sd[k].to(dev).clone().requires_grad_() for k in self.tnames}
            self.topt = torch.optim.Adam([self.tparams[k] for k in self.tnames], lr=3e-4)
            self.sd_live = {k: (self.tparams[k] if k in self.tparams else v) for k, v in sd.items()}
        self.coeffs = diffusion_coeffs(self.steps, dev)
        self.N = bs * self.S * 3
        self.row_offset = rank * self.N if row_offset is None else int(row_offset)
        self.gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.call = 0
        # Launch-bound sizes (a step is ~75 launches; below ~100 k rows most of them are 5-40 us): the step's launch sequence is
        # captured once in a HIP graph and replayed, the per-step seed travelling through a pstl_dyn block in device memory
        # (engine.GraphCapture; same kernels, same arguments, same results as the eager launches).  Single-GPU, in-kernel noise,
        # sampling workloads only; --no_graph measures the eager launches.
        self.use_graph = (world == 1 and not a.no_graph and a.noise == "kernel" and not self.train and not self.trajopt
                          and self.N <= GRAPH_MAX_ROWS)
        self.graph = None

    def step(self):
        from pstl_diffusion_policy_amd.engine import PackedWeights, RectTrainer, Sampler, SceneBatch
        from pstl_diffusion_policy_amd.shard import gather_final, global_valid_stats
        a, dev, N, S, steps, sampler = self.a, self.dev, self.N, self.S, self.steps, self.sampler
        # global mean(valid) of the guidance loss (one tiny all-reduce; the shard split must not change results).  Its host
        # wall time is booked separately (`host_scalar_exchange_ms_per_step` of the line): with real ranks it is a per-step
        # host round trip over the gloo side group
        t_ex = time.perf_counter()
        vsum, vrows = global_valid_stats(self.ids_host * S, N, torch.device("cpu") if (self.cpu_group or self.world == 1) else dev,
                                         group=self.cpu_group)
        self.exchange_s += time.perf_counter() - t_ex
        self.exchange_n += 1
        sb = SceneBatch(self.scene, S, self.hp, dev, global_valid_sum=vsum, global_rows=vrows, row_offset=self.row_offset,
                        plan_rows=self.plan_rows)
        if a.noise == "torch":
            x_T = torch.randn(N, 40, device=dev, generator=self.gen)
            z = torch.randn(steps - 1, N, 40, device=dev, generator=self.gen)
            seed = None
        else:
            x_T = z = None
            self.call += 1
            seed = 987654321 + self.call
        if self.use_graph:
            if self.graph is None:
                from pstl_diffusion_policy_amd.engine import DynBlock, GraphCapture
                self.dyn = DynBlock(dev)
                self.dyn.set(seed, SceneBatch.loss_scale(vsum, vrows))

                def body():
                    sbg = SceneBatch(self.scene, S, self.hp, dev, global_valid_sum=vsum, global_rows=vrows, row_offset=self.row_offset,
                                     dyn=self.dyn.dev, scale_in_dyn=True, plan_rows=self.plan_rows)
                    o = sampler.sampling_region(sbg, steps, None, None, rect_head=self.rect_head,
                                                multi_cands=a.multi_cands if self.rect_head else None, guidance=self.guidance,
                                                coeffs=self.coeffs, want_scores3=False, seed=0, diversity=True)
                    return o["counts"], o["div_totals"]

                self.graph = GraphCapture(body)
            self.dyn.set(seed)
            return self.graph.replay()
        if self.trajopt:
            params = self.scene["params"].reshape(N, 40).clone()
            sc, _ = sampler.trajopt(sb, params, a.trajopt_iters, 0.005, 0.01, 10.0, global_valid_sum=vsum, global_rows=vrows)
            counts, _ = sampler.metrics(sb, sc)
            return gather_final(counts, torch.zeros(12, dtype=torch.float64, device=dev))
        if self.train:
            # the optimiser changed the trained tensors: their networks are packed again, in place (rect_net only without
            # --joint); max |w| is not read back -- the kernels compare it themselves and train_step reads the domain word
            if getattr(self, "pw_train", None) is None:
                self.pw_train = PackedWeights(self.sd_live, dev)
            else:
                nets = {k.split(".")[0] for k in self.tnames}
                self.pw_train.update({k: v for k, v in self.sd_live.items() if k.split(".")[0] in nets}, read_status=False)
            sm_t = Sampler(self.pw_train, self.hp, chain_waves=self.chain_waves)
            sm_t.trace, sm_t.trace_bwd = sampler.trace, sampler.trace_bwd
            loss, scores = RectTrainer(sm_t).train_step(sb, self.tparams, self.topt, steps, x_T=x_T, noise=z, seed=seed,
                                                        multi_cands=a.multi_cands, coeffs=self.coeffs, e7=self.e7, joint=self.joint)
            if sm_t.chain_fallback and self.chain_waves in (0, 16, 2):
                # train_step reads the split-f16 domain flag before the optimiser consumes the gradients and repeats the step
                # on the exact-fp32 kernels when it is set; a bench line must not silently mix the two arithmetics
                raise FloatingPointError("bench: " + sm_t.chain_fallback)
            counts, _ = sm_t.metrics(sb, scores)
            return gather_final(counts, torch.zeros(12, dtype=torch.float64, device=dev))
        out = sampler.sampling_region(sb, steps, x_T, z, rect_head=self.rect_head,
                                      multi_cands=a.multi_cands if self.rect_head else None, guidance=self.guidance,
                                      coeffs=self.coeffs, want_scores3=False, seed=seed, diversity=True)
        # the only exchange after the rollout: the final diversity / STL-satisfaction reduction -- 8 counters + 12
        # diversity totals (+ 2 words of device identity) per rank in one RCCL all-gather over xGMI (when N > 1)
        return gather_final(out["counts"], out["div_totals"], ident=self.ident, seen=self.seen)

    def measure_best(self, steps, warmup, repeats):
        """measure() `repeats` times (warm-up once); the repetition with the smallest wall time.  For the secondary blocks of
        the line ("also", "sweep": a few short steps each), where one descheduling of this process on a shared host would
        otherwise be a third of the measurement; the headline is never measured this way."""
        best = None
        for r in range(repeats):
            m = self.measure(steps, warmup if r == 0 else 0)
            if best is None or m["dt"] < best["dt"]:
                best = m
        best["timing"] = "best of %d x %d steps" % (repeats, steps)
        return best

    def measure(self, steps, warmup, dist=None):
        """`warmup` untimed steps, then exactly `steps` timed ones between barrier + synchronize on both sides; MAX over ranks."""
        for _ in range(warmup):
            self.step()
        self.exchange_s, self.exchange_n = 0.0, 0
        sm = self.sampler
        if self.use_graph and self.graph is None:
            self.step()          # (the capture itself: never inside the timed region)
        # per-kernel HIP events are recorded by the eager launches only (events cannot be timed inside a graph): a graph-replaying
        # job takes its kernel timings from one extra eager step AFTER the timed ones
        if not self.use_graph:
            sm.trace, sm.trace_stl, sm.trace_bwd = [], {}, ([] if self.train else None)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            counts, div_totals = self.step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        if self.use_graph:
            self.use_graph = False
            sm.trace, sm.trace_stl = [], {}
            self.step()
            torch.cuda.synchronize()
            self.use_graph = True
        if not self.train and not self.trajopt:      # (train workloads: checked inside every train_step, see step())
            sm.check_chain_domain(fallback=False)    # a split-f16 launch that left its domain leaves undefined results
        res = {"dt": dt, "ms_per_step": dt / steps * 1e3,
               "value": getattr(self, "total_rows", self.world * self.N) * steps / dt, "counts": counts,
               "div_totals": div_totals, "steps": steps,
               "host_scalar_exchange_ms_per_step": self.exchange_s / max(self.exchange_n, 1) * 1e3}
        multi = [(e0.elapsed_time(e1), n, rows) for (e0, e1, n, rows) in (sm.trace or [])]
        if multi:      # dominant kernel (k_chain, the multi-step denoiser launch): HIP events on the launch stream
            nst, nrows = multi[0][1], multi[0][2]
            k_ms = float(np.mean([m for m, n, _ in multi if n == nst]))
            flop = float(nrows) * nst * F_STEP_MIN
            res.update(kernel_ms=k_ms, kernel_steps=nst, flop=flop, achieved=flop / (k_ms * 1e-3) / 1e12)
        # the STL kernels (one row per lane): row-evaluations/s and what that means against the HBM roofline.  Algorithmic
        # bytes per row-evaluation in the scene-shared layout: 160 B controls + 16 B s0 + 24 B stlp + 4 B score + scene
        # tables amortised over the 192 rows of a scene -- these kernels are VALU-bound, not HBM-bound.
        K = self.a.neighbors
        stl_bytes = 160 + 16 + 24 + 4 + (K * 20 * 7 * 4 + 540) / (3.0 * self.S)
        stl_info = {}
        traced_steps = 1 if self.use_graph else steps
        for kind, evs in (sm.trace_stl or {}).items():
            ms_k = sum(e0.elapsed_time(e1) for (e0, e1, _) in evs)
            evals = sum(n for (_, _, n) in evs)
            if ms_k > 0:
                rate = evals / (ms_k * 1e-3)
                stl_info[kind] = {"kernel": "k_guidance_iter (forward + adjoint + Adam)" if kind == "guidance" else "k_stl_forward",
                                  "row_evals_per_s": rate, "ms_per_step": ms_k / traced_steps,
                                  "algorithmic_bytes_per_row_eval": stl_bytes * (2 if kind == "guidance" else 1),
                                  "achieved_GBps": rate * stl_bytes * (2 if kind == "guidance" else 1) / 1e9,
                                  "frac_of_hbm_peak": rate * stl_bytes * (2 if kind == "guidance" else 1) / 8e12}
        res["stl_info"] = stl_info
        if self.train and sm.trace_bwd:
            b_ms = float(np.mean([e0.elapsed_time(e1) for (e0, e1, _) in sm.trace_bwd]))
            gbps = self.N * BWD_BYTES_PER_ROW / (b_ms * 1e-3) / 1e9
            res["backward"] = {"kernel": "pstl_refine_backward (head, dgrad x2, wgrad x4, column/scene sums)", "ms": b_ms,
                               "algorithmic_bytes": self.N * BWD_BYTES_PER_ROW, "achieved_GBps": gbps, "peak_GBps": 8000.0,
                               "frac_of_hbm_peak": gbps / 8000.0,
                               "note": "algorithmic bytes = h1, h2, dcontrols, pre, init, hl|stlp read once (%d B/row)" % BWD_BYTES_PER_ROW}
        sm.trace, sm.trace_stl, sm.trace_bwd = None, None, None
        return res


def paper_metric(batches=8):
    """The paper's own metric inside the driver's run (missing #2): md["time"] of run_sampling_test -- the wall time of
    the region reference nusc_train.py:957-1105 per batch of 128 scenes x 64 x 3 = 24 576 rows at 100 diffusion steps and 8
    neighbours, device synchronised on both sides -- for the README command lines "Ours" and "Ours+guidance" (reference
    README.md:114,120) through the CLI mirror (pstl_diffusion_policy_amd.nusc_train.main) with in-kernel noise, on synthetic
    scenes with random-init weights; median over the batches after the first.  Context, not target: the paper's numbers are
    from an unspecified NVIDIA GPU on real nuScenes data, and the reference's timer does not synchronise the device."""
    import contextlib
    import io
    import statistics
    from pstl_diffusion_policy_amd import nusc_train as nt
    common = ["-e", "e7_ours", "--diffusion", "--stl_weight", "0.0", "--load_stlp", "--rect_head", "--flex", "--diverse_loss",
              "--test", "-P", "e7_ours", "--run_sampling_test", "--skip_nusc_load", "--viz_correct", "--allow_random_init",
              "--kernel_noise", "--n_trials", str(batches - 1)]
    configs = {"ours": (common + ["--multi_cands", "5"], 0.174),
               "ours_guidance": (common + ["--multi_cands", "10", "--guidance", "--guidance_before", "10", "--guidance_niters", "1",
                                           "--guidance_lr", "0.01", "--n_rolls", "3", "--other"], 0.786)}
    out = {"rows_per_batch": 128 * 64 * 3, "batches": batches, "through": "nusc_train.main (the reference's CLI surface), --kernel_noise"}
    out["timing"] = "median over the batches after the first, the better of two runs of the command line"
    for name, (argv, paper_s) in configs.items():
        med = None
        for _ in range(2):   # (md["time"] includes the host: one busy spell of a shared box -- seen once: 68 ms -- must not become the figure)
            with contextlib.redirect_stdout(io.StringIO()):
                md = nt.main(argv)
            t = md.hist["time"]
            m = statistics.median(t[1:] if len(t) > 1 else t)
            med = m if med is None else min(med, m)
        out[name] = {"time_ms_median": med * 1e3, "trajectories_per_s": out["rows_per_batch"] / med, "stl_sat_rate": md("acc"),
                     "paper_time_ms": paper_s * 1e3}
    return out


def chain_peak(chain_waves):
    # split forms: three 16-bit MFMA products per fp32 product, so the roofline of the fp32 work they deliver is the
    # dense 16-bit matrix peak / 3
    return PEAK_BF16_MATRIX_TFLOPS / 3 if chain_waves in (0, 16, 32) else PEAK_FP32_MATRIX_TFLOPS


def main():
    a = parse()
    if a.gpus < 1:
        sys.exit("bench: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("bench: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (a.gpus, world))
    import torch.distributed as dist
    # PSTL_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a box with fewer GPUs than ranks (ranks then
    # share devices); the real runs use nccl (= RCCL over xGMI), one process per GPU.
    backend = os.environ.get("PSTL_BENCH_BACKEND", "nccl")
    ndev = max(count_gpus(), 1)
    if backend == "nccl" and world > ndev:
        sys.exit("bench: %d ranks but %d GPU(s): one process per GPU" % (world, ndev))
    local = local % ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != a.gpus:
            sys.exit("bench: the process group has %d ranks, --gpus %d" % (dist.get_world_size(), a.gpus))
    # host-side scalars (the global valid-row statistics of the guidance loss) travel over a gloo side group, so that a
    # step never has to wait for the GPU: batches are enqueued back to back
    cpu_group = None
    if world > 1 and backend == "nccl":
        try:
            cpu_group = dist.new_group(backend="gloo")
        except Exception as e:      # no gloo transport on this node: fall back to the device all-reduce (one sync per step)
            print("bench: gloo side group unavailable (%s); host scalars go through RCCL" % e, file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from pstl_diffusion_policy_amd.engine import PackedWeights, acc_from_counts, diversity_from_totals
    from pstl_diffusion_policy_amd.nusc_model import init_state_dict
    from pstl_diffusion_policy_amd.synthetic import default_hparams

    hp = default_hparams()
    sd = init_state_dict(1007)     # random init as in the reference under seed 1007 (no checkpoints offline)
    weights = PackedWeights(sd, dev)
    from pstl_diffusion_policy_amd.shard import distinct_devices, plan_rows, shard_range
    rps = 3 * a.sampling_size
    if a.scaling == "strong":      # a fixed job split over the ranks in contiguous blocks of scenes (BASELINE config 4 as written)
        if a.total_scenes < world:
            sys.exit("bench: --scaling strong needs at least one scene per rank")
        lo, hi = shard_range(a.total_scenes, rank, world)
        my_scenes, my_offset, total_rows = hi - lo, lo * rps, a.total_scenes * rps
        plan = plan_rows(a.total_scenes, world, rps)
    else:
        my_scenes, my_offset, total_rows = a.scenes, rank * a.scenes * rps, world * a.scenes * rps
        plan = a.scenes * rps
    mk = lambda workload, bs, chain_waves=a.chain_waves, **kw: Job(a, workload, bs, dev, rank, world, cpu_group, weights, hp, sd,
                                                                   chain_waves, joint=a.joint, **kw)
    job = mk(a.workload, my_scenes, plan_rows=plan if world > 1 else 0, row_offset=my_offset)
    job.total_rows = total_rows
    S, steps, bs, N = job.S, job.steps, my_scenes, job.N
    rect_head, guidance, train, trajopt = job.rect_head, job.guidance, job.train, job.trajopt
    log("headline: %s, %d scenes on this rank, %d warm-up + %d timed steps" % (a.workload, my_scenes, a.warmup, a.steps))
    m = job.measure(a.steps, a.warmup, dist if world > 1 else None)
    dt, counts, div_totals = m["dt"], m["counts"], m["div_totals"]
    log("headline done: %.2f ms per step" % m["ms_per_step"])

    if trajopt:   # no denoiser in this workload: report row-iterations/s and stop
        if rank == 0:
            print(json.dumps({"metric": "traj-opt row-iterations/sec (STL forward + adjoint + Adam per row and iteration)",
                              "value": world * N * a.trajopt_iters * a.steps / dt, "unit": "row-iterations/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                              "data": "synthetic",
                              "config": {"workload": "trajopt: %d scenes x %d x 3 = %d rows/GPU, K=%d, %d Adam iterations per "
                                                     "step in one launch" % (bs, S, N, a.neighbors, a.trajopt_iters)},
                              "stl_sat_rate": acc_from_counts(counts)[0]}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    nst, k_ms, flop, achieved = m["kernel_steps"], m["kernel_ms"], m["flop"], m["achieved"]
    split_f16 = a.chain_waves in (0, 16, 2)
    split_bf16 = a.chain_waves == 32
    peak = chain_peak(a.chain_waves)
    dtype = ("f32 (MLP products formed from two f16 pieces per operand, 2^-23 per operand, 3 v_mfma_f32_16x16x32_f16 per f32 "
             "product, f32 accumulate; everything else plain f32)") if split_f16 else (
        "bf16x3 (two bf16 pieces per f32 operand, 2^-17 per operand, 3 bf16 MFMA products per f32 product, f32 accumulate; "
        "STL and rect_net in f32)") if split_bf16 else "f32"
    acc, sacc = acc_from_counts(counts)
    stl_info = m["stl_info"]
    extras = world == 1 and not a.no_extras
    # Second opinion inside the same run: the same step with both MLP chains on the exact-fp32 MFMA
    # kernels (chain_waves 8: bit-for-bit a k-ordered fmaf chain), three timed steps after one warm-up, priced against the
    # dense fp32 matrix peak.
    fp32_exact = None
    sampler_exact = None
    if extras and not train and a.chain_waves != 8:
        log("extras: the exact-fp32 leg")
        jx = mk(a.workload, a.scenes, 8)
        jx.scene, jx.ids_host = job.scene, job.ids_host
        mx = jx.measure(3, 1)
        sampler_exact = jx.sampler
        fp32_exact = {"chain_waves": 8, "steps": 3, "ms_per_step": mx["ms_per_step"], "value": mx["value"],
                      "kernel_ms": mx["kernel_ms"], "achieved": mx["achieved"], "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                      "frac": mx["achieved"] / PEAK_FP32_MATRIX_TFLOPS,
                      "note": "v_mfma_f32_16x16x4_f32 chains (exact f32), same step, same run"}
    # (which kernel the multi-step denoiser launches of this batch run on: the library's own answer)
    from pstl_diffusion_policy_amd import ffi as _ffi2
    dom_kernel = "k_chain2" if _ffi2.rollout_layout(_ffi2.make_cfg(bs, S * 3, S, a.neighbors, steps, hp, _ffi2.PSTL_FLAG_RNG,
                                                                    a.chain_waves, plan_rows=job.plan_rows))[0] == 2 else "k_chain"
    # BASELINE configs 2, 3 and 5 in the driver's own run: three timed steps each after two warm-ups,
    # same shard size, each with the roofline fraction of ITS dominant launch (the 49-step denoiser launch)
    also = None
    if extras:
        also = {}
        for wl in ("e5", "e7", "e8_train"):
            if wl == a.workload:
                continue
            log("extras: also." + wl)
            jw = mk(wl, a.scenes)
            jw.scene, jw.ids_host = job.scene, job.ids_host
            mw = jw.measure_best(3, 2, 2)
            rec = {"steps": 3, "timing": mw["timing"], "ms_per_step": mw["ms_per_step"], "value": mw["value"], "unit": "trajectories/s",
                   "stl_sat_rate": acc_from_counts(mw["counts"])[0],
                   "roofline": {"bound": "mfma", "kernel": "%s (%d reverse steps per launch)" % (dom_kernel, mw["kernel_steps"]),
                                "kernel_ms": mw["kernel_ms"], "achieved": mw["achieved"], "peak": peak, "unit": "TFLOP/s",
                                "frac": mw["achieved"] / peak}}
            if "backward" in mw:
                rec["backward"] = mw["backward"]
            also[wl] = rec
            del jw
            torch.cuda.empty_cache()
    # the headline workload at other batch sizes: best of 3 x 3 timed steps each after 2 warm-ups
    sweep = None
    if extras and not train:
        from pstl_diffusion_policy_amd import ffi as _ffi
        layout_cfg = lambda nsc: _ffi.make_cfg(nsc, S * 3, S, a.neighbors, steps, hp, _ffi.PSTL_FLAG_RNG, a.chain_waves)
        sweep = []
        for sbs in (1, 16, 128, 512):
            if sbs >= a.scenes:
                continue
            log("extras: sweep, %d scenes" % sbs)
            js = mk(a.workload, sbs)
            ms_ = js.measure_best(3, 2, 3)
            sweep.append({"rows": js.N, "ms_per_step": ms_["ms_per_step"], "value": ms_["value"], "layout": chain_layout(layout_cfg(sbs)),
                          "launches": "one HIP-graph replay per step" if js.use_graph else "eager", "timing": ms_["timing"]})
        sweep.append({"rows": N, "ms_per_step": m["ms_per_step"], "value": m["value"], "layout": chain_layout(layout_cfg(a.scenes)),
                      "launches": "one HIP-graph replay per step" if job.use_graph else "eager",
                      "timing": "the headline measurement (%d steps)" % a.steps})
    # VALU-issue roofline of the one-row-per-lane STL kernels (they are instruction-issue-bound, not HBM-bound): vector
    # instructions per row-evaluation from the committed PMC pass (SQ_INSTS_VALU / rows of that launch) x the row-evaluation
    # rate measured live in this run x 4 issue cycles per wavefront instruction / (1024 SIMDs x the clock of the PMC run)
    pmc_path = None
    for rnd in ("r6", "r5", "r4", "r3", "r2"):
        cand = os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")
        if os.path.exists(cand):
            pmc_path = cand
            break
    pj = json.load(open(pmc_path)) if pmc_path else {}
    for kind, info in stl_info.items():
        v = pj.get("stl_kernels", {}).get(kind)
        if v and v.get("rows_per_launch") and v.get("K") == a.neighbors and v.get("clock_GHz"):
            per_row = v["SQ_INSTS_VALU"] / v["rows_per_launch"]       # wavefront instructions per row-evaluation (1/64 each)
            clk = float(v["clock_GHz"]) * 1e9
            info["valu_wave_insts_per_row_eval"] = per_row
            # SIMD cycles per wavefront vector instruction, live: 1024 SIMDs x clock / (instructions per row-evaluation x the
            # run's row-evaluations/s).  What a SIMD CAN issue was measured in round 6 (profiles/r6/valu_rate.txt: 2.0 - 3.7
            # cycles by kind at 3 waves per SIMD, 1.6 - 3.1 at 5; transcendentals 5.7 / 5.3); rounds 3-5 assumed 4 for all of
            # them, which is what `valu_issue_frac_4cycle_model` still divides by (kept for comparison with those rounds).
            waves = 3 if kind == "guidance" else 5
            floor = 2.9 if kind == "guidance" else 2.3       # mix-weighted estimate for these kernels' instruction mix
            info["valu_cycles_per_inst_per_simd"] = 1024.0 * clk / (per_row * info["row_evals_per_s"])
            info["waves_per_simd"] = waves
            info["issue_floor_cycles_per_inst_estimate"] = floor
            info["frac_of_issue_floor"] = floor / info["valu_cycles_per_inst_per_simd"]
            info["valu_issue_frac_4cycle_model"] = per_row * info["row_evals_per_s"] * 4.0 / (1024.0 * clk)
            info["valu_issue_note"] = ("SQ_INSTS_VALU per row-evaluation (%s) x live row-evaluations/s against 1024 SIMDs x %.2f GHz; "
                                       "issue floor: profiles/r6/valu_rate.txt weighted by the kernels' instruction mix (an estimate)"
                                       % (os.path.relpath(pmc_path, ROOT), clk / 1e9))
            if v.get("hbm_bytes_per_row_eval"):
                info["hbm_bytes_per_row_eval_pmc"] = v["hbm_bytes_per_row_eval"]
    # HBM bytes of that launch from the committed PMC passes (FETCH_SIZE/WRITE_SIZE cannot be read inside this process);
    # only quoted when this run is the configuration those passes were collected on
    traffic = None
    is_default = (a.workload == "e7_guid" and bs == 4096 and S == 64 and a.neighbors == 2 and steps == 50
                  and a.noise == "kernel" and not a.chain_waves)
    if is_default and pj.get("summary_dominant_kernel"):
        traffic = pj["summary_dominant_kernel"]["hbm_bytes_per_launch_corrected"]
    line = None
    if rank == 0:
        line = {
            "metric": "sampled trajectories/sec (%d DDPM steps, multi_cands=%d) + STL-sat rate" % (steps, a.multi_cands),
            "value": total_rows * a.steps / dt, "unit": "trajectories/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": ("%s: %d scenes/GPU" + (" (of %d, strong scaling)" % a.total_scenes if a.scaling == "strong" else "") +
                                    " x sampling_size %d x 3 modes = %d rows/GPU, T=20, K=%d neighbours, "
                                    "diffusion_steps=%d (%d denoiser evals), multi_cands=%s, guidance=%s, RefineNet=%s, "
                                    "random-init weights (seed 1007)")
                                   % (a.workload + (" --joint" if (train and a.joint) else ""), bs, S, N, a.neighbors, steps, steps - 1,
                                      a.multi_cands if rect_head else None,
                                      "before=10,niters=1,lr=0.01" if guidance else None, rect_head),
                       "rows_per_gpu": N, "rows_whole_job": total_rows,
                       "parallelism": "scene shards x%d, no data-path collective" % world,
                       "plan_rows": job.plan_rows,
                       "chain_waves": a.chain_waves,
                       "noise": "in-kernel Philox4x32-7" if a.noise == "kernel" else "torch.randn tensors"},
            # who took part: every rank's record of the final all-gather carries its device's identity
            "ranks": {"backend": (backend + (" (= RCCL)" if backend == "nccl" else "")) if world > 1 else "single process",
                      "world_size": world, "ranks_seen": job.seen.get("ranks_seen", 1), "distinct_devices": distinct_devices(job.seen),
                      "rank0_device": job.ident_text,
                      "host_scalar_exchange_ms_per_step": m["host_scalar_exchange_ms_per_step"],
                      "host_scalar_exchange": "global sum(valid), rows over the %s" % (
                          "gloo side group" if cpu_group is not None else "process group" if world > 1 else "local values (one rank)")},
            "stl_sat_rate": acc, "scene_sat_rate": sacc, "counts": [int(v) for v in counts.tolist()],
            "diversity": None if train else diversity_from_totals(div_totals),
            "roofline": {"bound": "mfma", "kernel": "%s (denoiser MLP chain, %d reverse steps per launch)" % (dom_kernel, nst),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "peak_note": ("dense f16/bf16 MFMA peak 2516.8 TFLOP/s / 3 products per f32 product"
                                       if (split_f16 or split_bf16) else "dense f32 MFMA peak"),
                         "matrix_pipe_frac_issued": (achieved * F_STEP_ISSUED_BF16 / F_STEP_MIN / PEAK_BF16_MATRIX_TFLOPS)
                                                    if (split_f16 or split_bf16) else achieved / peak,
                         "traffic_source": "%s (2*FETCH_SIZE + WRITE_SIZE, bytes per launch)" % os.path.relpath(pmc_path, ROOT)
                                           if traffic else None,
                         "fp32_exact": fp32_exact,
                         "kernel_ms": k_ms, "flop_per_launch": flop, "stl_kernels": stl_info,
                         "whole_step_frac": ((steps - 1) * F_STEP_MIN + ((F_STEP_MIN + 7168) if rect_head else 0))
                                            * float(N) / (dt / a.steps) / 1e12 / peak,
                         "note": "algorithmic FLOP = rows x steps x 172032 (hoisted layer-1 columns not counted); achieved "
                                 "counts every f32 multiply-add once, whatever the kernel issues for it; whole_step_frac = (all "
                                 "denoiser evaluations + RefineNet + merge_net, same minimal count) / ms_per_step / peak"},
        }
        if "backward" in m:
            line["roofline"]["backward"] = m["backward"]
        if also is not None:
            line["also"] = also
        if sweep is not None:
            line["sweep"] = sweep
        if extras and a.workload == "e7_guid" and not a.chain_waves:
            log("extras: paper_metric (the README command lines through the CLI mirror)")
            line["paper_metric"] = paper_metric()
        if not a.no_cpu_baseline and world == 1:      # the CPU leg runs on rank 0 of the single-GPU run only
            line["cpu_baseline"] = cpu_baseline(a, hp, sd, guidance, rect_head, None if (train or trajopt) else job.sampler, dev,
                                                sampler_exact=sampler_exact)
        log("printing the line")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
