"""Scene-indexed driver of the HIP path: packs weights, prepares a scene batch and runs the sampling region
(reference nusc_train.py:957-1105) through libpstl_hip.so without ever materialising row-replicated tensors.

Host code only sequences launches; all arithmetic on rows happens in the HIP kernels.
"""
import ctypes
import os
import math
import warnings

import numpy as np
import torch

from . import ffi

MLP_NAMES = ("ego_encoder", "neighbor_encoder", "lane_encoder", "policy_net", "merge_net", "rect_net")


def diffusion_coeffs(steps, device=None):
    """Cosine schedule, float32, same op sequence as the reference (nusc_train.py:528-537; --cos is forced on)."""
    t = torch.linspace(0, 1, steps + 1)
    ab = torch.cos((t + 0.008) / 1.008 * np.pi / 2) ** 2
    beta = torch.clip(1 - ab[1:] / ab[:-1], 0, 0.999) * 0.2
    alpha = 1.0 - beta
    alpha_hat = torch.cumprod(alpha, dim=0)
    if device is not None:
        return beta.to(device), alpha.to(device), alpha_hat.to(device)
    return beta, alpha, alpha_hat


class PackedWeights:
    """Kernel-layout copy of a reference state_dict (keys '<net>.{0,2,4}.{weight,bias}')."""

    def __init__(self, state_dict, device):
        self.device = torch.device(device)
        L = ffi.lib()
        wp, keep = self._pointers(state_dict)
        self.has_rect = "rect_net.0.weight" in state_dict
        self.has_merge = "merge_net.0.weight" in state_dict
        self.packed = torch.empty(L.pstl_packed_weight_floats(), dtype=torch.float32, device=self.device)
        ffi.check(L.pstl_pack_weights(ctypes.byref(wp), ffi.ptr(self.packed), ffi.stream()), "pack_weights")
        so = int(L.pstl_packed_status_offset())
        self.status = self.packed[so:so + 16]
        self._tbias = {}
        self._read_status()
        del keep

    def _pointers(self, state_dict):
        """WeightPtrs of the networks `state_dict` holds completely (six tensors each) + the converted copies to keep alive."""
        keep = {}
        wp = ffi.WeightPtrs()
        for name in MLP_NAMES:
            m = ffi.Mlp3()
            for li, idx in enumerate((0, 2, 4)):
                for kind, field in (("weight", "w%d" % li), ("bias", "b%d" % li)):
                    key = "%s.%d.%s" % (name, idx, kind)
                    if key in state_dict:
                        t = state_dict[key]
                        if not isinstance(t, torch.Tensor):
                            t = torch.from_numpy(np.ascontiguousarray(t))
                        t = ffi.f32(t.detach(), self.device)
                        keep[key] = t
                        setattr(m, field, t.data_ptr())
            have = sum(1 for f, _ in m._fields_ if getattr(m, f))
            if have not in (0, 6):
                raise ValueError("state_dict holds %d of the 6 tensors of %s: a network is packed whole" % (have, name))
            setattr(wp, name, m)
        return wp, keep

    def _read_status(self):
        # Status block of the packed buffer (include/pstl_hip.h, pstl_packed_status_offset): max |w| of the weights the MLP
        # chains carry as half pieces, per network.  Reading it is a synchronisation (after which converted source blobs may go).
        wmax = [float(v) for v in self.status[:2].cpu()]
        self.chain_wmax = dict(policy_net=wmax[0], rect_net=wmax[1])
        # domain of the default (split-f16) chain arithmetic: |w| < 63.9 (NaN compares false)
        self.split_f16_ok = all(m < ffi.SPLIT_F16_WMAX for m in wmax)

    def update(self, state_dict, read_status=True):
        """Packs again, in place, the networks `state_dict` holds completely (an optimiser step changed them; the others stay):
        RefineNet training changes rect_net only -- 15 launches instead of the 75 of a full pack.  read_status=False skips the
        synchronising read of max |w|: the kernels compare the recorded maximum themselves and set the domain word, which a
        caller that checks it anyway (RectTrainer.train_step) then sees."""
        wp, keep = self._pointers(state_dict)
        ffi.check(ffi.lib().pstl_repack_weights(ctypes.byref(wp), ffi.ptr(self.packed), ffi.stream()), "repack_weights")
        if any(k.startswith("policy_net.") for k in state_dict):
            self._tbias = {}
        if read_status:
            self._read_status()
        else:
            self.chain_wmax, self.split_f16_ok = None, True
            self._keep = keep      # (no synchronisation here: the converted copies live until the next update)
        return self

    def chain_overflowed(self, clear=False):
        """True when a launch on the split-f16 arithmetic since the last clear left a non-finite state: a layer input was
        outside the half range (|x| >= 4094).  Synchronises -- call it where the caller synchronises anyway."""
        flag = self.status[2:3].view(torch.int32)
        hit = bool(flag.item() != 0)
        if clear:
            flag.zero_()
        return hit

    def tbias(self, steps):
        if steps not in self._tbias:
            tb = torch.empty(steps, ffi.HID, dtype=torch.float32, device=self.device)
            ffi.check(ffi.lib().pstl_time_bias(ffi.ptr(self.packed), int(steps), ffi.ptr(tb), ffi.stream()), "time_bias")
            self._tbias[steps] = tb
        return self._tbias[steps]


_MODE_COLUMNS = {}


def _mode_column(n, dev):
    """(0, 1, 2) repeated n times (the `highlevel` column of the reference's dense rows), built once per size and device."""
    key = (int(n), str(dev))
    if key not in _MODE_COLUMNS:
        if len(_MODE_COLUMNS) > 16:
            _MODE_COLUMNS.clear()
        _MODE_COLUMNS[key] = torch.tensor([0.0, 1.0, 2.0], device=dev).repeat(int(n)).contiguous()
    return _MODE_COLUMNS[key]


class SceneBatch:
    """Device-resident, scene-indexed inputs of one batch (schema: SURVEY.md 3.0) + per-row constants.

    scene: dict with ego_traj (bs,nt,6), neighbors (bs,K,7), neighbors_traj (bs,K,nt,7), {curr,left,right}lane_wpts
    (bs,15,3), {curr,left,right}_id (bs,1), and either stlp_modes (bs,3,6) or stlp_rows (N,6).
    """

    def __init__(self, scene, S, hp, device, global_valid_sum=None, global_rows=None, row_offset=0, dyn=None,
                 scale_in_dyn=False, plan_rows=0):
        """plan_rows: the row count the default arithmetic (chain_waves = 0) picks its denoiser kernel for (pstl_cfg.plan_rows).
        0: this batch's own rows.  A job that shards a batch passes the SAME number on every shard (shard.plan_rows: the rows of
        its largest shard), so that all its rows run through the same kernel and a row's bits do not depend on the shard that
        holds it; a single process that wants the bits of such a job passes the job's number.
        dyn: None, or a 4-float32 device tensor laid out as a pstl_dyn (include/pstl_hip.h): the kernels then read the noise
        seed from its first 8 bytes and the guidance-loss scale from its third word -- which THIS constructor writes there, on
        the device, from the lane ids (no host synchronisation) -- instead of taking them by value: the launches of a whole
        planning step can be captured in a HIP graph and replayed with new inputs (nusc_sim.py).  scale_in_dyn: the caller has
        already put the scale there (SceneBatch.loss_scale of the host copy of the lane ids): nothing is computed here."""
        dev = torch.device(device)
        self.row_offset = int(row_offset)   # global index of the first row (in-kernel noise is keyed by global row)
        self.plan_rows = int(plan_rows or 0)
        self.dyn = dyn
        # A batch that arrives in host memory (the closed-loop caller builds one scene per simulation step) crosses PCIe as
        # ONE staged copy instead of one per tensor, and the guidance-loss scale is taken from the host copy of the lane ids:
        # no device synchronisation while the batch is set up.
        keys = ["ego_traj", "neighbors", "neighbors_traj", "currlane_wpts", "leftlane_wpts", "rightlane_wpts", "curr_id",
                "left_id", "right_id", "stlp_rows" if "stlp_rows" in scene else "stlp_modes"]
        src = {k: torch.as_tensor(scene[k]) for k in keys}
        host = dev.type == "cuda" and all(not t.is_cuda for t in src.values())
        if host:
            pad = lambda t: torch.nn.functional.pad(t.reshape(-1).to(torch.float32), (0, -t.numel() % 4))   # 16-byte slots
            flat = torch.cat([pad(t) for t in src.values()]).to(dev)
            parts, o = {}, 0
            for k, t in src.items():
                parts[k] = flat[o:o + t.numel()].reshape(t.shape)
                o += t.numel() + (-t.numel() % 4)
            f = lambda k: parts[k]
            if global_valid_sum is None:
                global_valid_sum = float(sum(src[k].to(torch.float32).sum() for k in ("curr_id", "left_id", "right_id"))) * int(S)
        else:
            f = lambda k: ffi.f32(src[k], dev)
        self.hp, self.S, self.device = hp, int(S), dev
        ego = f("ego_traj")
        self.bs = ego.shape[0]
        self.ego_traj = ego                       # (bs,nt,6): the ground-truth future, only read by the ADE/FDE metric
        self.ego0 = ego[:, 0, :].contiguous()
        self.s0 = ego[:, 0, :4].contiguous()
        self.neighbors = f("neighbors")
        self.nei_traj = f("neighbors_traj")[..., :7].contiguous()
        self.K = self.nei_traj.shape[1]
        assert self.nei_traj.shape[2] == ffi.T, "this build is specialised for nt = 20"
        self.lanes = [f("%slane_wpts" % k) for k in ("curr", "left", "right")]
        assert self.lanes[0].shape[1] == ffi.NSEG, "this build is specialised for n_segs = 15"
        self.ids = [f("%s_id" % k).reshape(self.bs).contiguous() for k in ("curr", "left", "right")]
        self.rps = 3 * self.S
        self.N = self.bs * self.rps
        if "stlp_rows" in scene:
            self.stlp = f("stlp_rows").reshape(self.N, 6).contiguous()
        else:   # every sample of a (scene, mode) shares the STL parameters (reference nusc_train.py:745)
            self.stlp = f("stlp_modes").reshape(self.bs, 1, 3, 6).expand(self.bs, self.S, 3, 6).reshape(self.N, 6).contiguous()
        self.hl = _mode_column(self.bs * self.S, dev)                                                     # nusc_train.py:753
        ids3 = torch.stack(self.ids, dim=-1)                                                             # (bs,3)
        self.valid = ids3.reshape(self.bs, 1, 3).expand(self.bs, self.S, 3).reshape(self.N).contiguous()  # :751-752
        # scale of d loss / d score in the guidance loss mask_mean(relu(thres - score), valid) (nusc_train.py:23-27,619)
        rows = self.N if global_rows is None else int(global_rows)
        if dyn is not None and scale_in_dyn:
            self.grad_scale = 0.0
        elif dyn is not None and global_valid_sum is None:
            # the same float32 operations on the device, written into the parameter block the kernels read
            c = torch.clamp((ids3.sum() * float(self.S)) / float(rows), min=1e-2)
            dyn[2:3].copy_(((1.0 / c) / float(rows)).reshape(1))
            self.grad_scale = 0.0     # (the by-value argument is ignored when cfg.dyn is set)
        else:
            vsum = float(ids3.sum().item()) * self.S if global_valid_sum is None else float(global_valid_sum)
            self.grad_scale = self.loss_scale(vsum, rows)
            if dyn is not None:
                dyn[2:3].fill_(self.grad_scale)
        # prepared tables for the STL kernels
        self.nei_prep = torch.empty(self.bs, self.K, ffi.T, ffi.NEI_PREP, dtype=torch.float32, device=dev)
        self.lane_prep = torch.empty(self.bs, 3, ffi.NSEG, 4, dtype=torch.float32, device=dev)
        cfg = self.cfg(2)
        ffi.check(ffi.lib().pstl_prepare_scene(ctypes.byref(cfg), ffi.ptr(self.nei_traj), ffi.ptr(self.lanes[0]),
                                               ffi.ptr(self.lanes[1]), ffi.ptr(self.lanes[2]), ffi.ptr(self.nei_prep),
                                               ffi.ptr(self.lane_prep), ffi.stream()), "prepare_scene")

    @staticmethod
    def loss_scale(valid_sum, rows):
        """d loss / d score of mask_mean(relu(thres - score), valid) per violated valid row: (1 / clip(mean(valid), 1e-2)) / rows,
        in float32 as the reference computes it (nusc_train.py:23-27,619)."""
        mean_valid = np.float32(np.float32(valid_sum) / np.float32(rows))
        c = np.float32(max(mean_valid, np.float32(1e-2)))
        return float(np.float32(np.float32(1.0) / c) / np.float32(rows))

    def cfg(self, steps, flags=0, chain_waves=0, seed=0):
        return ffi.make_cfg(self.bs, self.rps, self.S, self.K, steps, self.hp, flags, chain_waves, seed, self.row_offset,
                            dyn=self.dyn, plan_rows=self.plan_rows)


def guidance_triggered(i, steps, g):
    """Trigger rule of the reference (nusc_train.py:589-598)."""
    if not g or not g.get("enabled", False):
        return False
    i_val = steps - 1 - i if g.get("reverse", False) else i
    if g.get("sets") is not None:
        return i_val in g["sets"]
    if g.get("freq") is not None:
        return i_val % g["freq"] == 0
    return i <= g.get("before", 1000)


class Sampler:
    def __init__(self, weights, hp, chain_waves=None):
        # chain_waves: arithmetic of the MLP chains (include/pstl_hip.h): 0 = both MLP chains on split-f16 MFMA products
        # (default), 8 = exact fp32 MFMA, 32 = split-bf16.  Callers that go through the reference's CLI surface choose with
        # PSTL_CHAIN_WAVES.
        if chain_waves is None:
            chain_waves = int(os.environ.get("PSTL_CHAIN_WAVES", "0"))
        self.w, self.hp, self.chain_waves = weights, hp, int(chain_waves)
        self.chain_fallback = None       # why the exact-fp32 kernels replaced the requested arithmetic, if they did
        if self.chain_waves in (0, 16, 2) and not weights.split_f16_ok:
            self.use_exact_fp32("a chain weight is outside the split-f16 domain |w| < %g (max |w|: policy_net %g, rect_net %g)"
                                % (ffi.SPLIT_F16_WMAX, weights.chain_wmax["policy_net"], weights.chain_wmax["rect_net"]))
        self.L = ffi.lib()
        # when set to a list, every multi-step rollout launch appends (start_event, end_event, n_steps, n_rows):
        # HIP events recorded on the launch stream, used by bench.py to time the dominant kernel live
        self.trace = None
        # when set to a dict, STL launches append (start_event, end_event, row_evaluations) under "guidance" / "score"
        self.trace_stl = None
        self.trace_bwd = None   # when set to a list, RectTrainer appends (start_event, end_event, n_rows) around pstl_refine_backward
        self.debug_buf = None   # diagnostic builds only (chain_waves 708): receives the kernel's cycle stamps

    def use_exact_fp32(self, why):
        """Switch the MLP chains to the exact-fp32 MFMA kernels (chain_waves 8: fp32's range, 0.36x the throughput)."""
        self.chain_fallback = why
        self.chain_waves = 8
        warnings.warn("pstl: MLP chains fall back to the exact-fp32 kernels (chain_waves = 8): " + why, RuntimeWarning,
                      stacklevel=3)

    def check_chain_domain(self, fallback=True):
        """Call after a region, where the caller synchronises anyway: did a split-f16 launch overflow the half range (a layer
        input |x| >= 4094)?  Returns False when all is well.  Otherwise the results of the region hold NaNs; with
        fallback=True the sampler is switched to the exact-fp32 kernels, the flag cleared and True returned -- the caller
        re-runs the batch; with fallback=False a FloatingPointError is raised."""
        if self.chain_waves not in (0, 16, 2) or not self.w.chain_overflowed(clear=True):
            return False
        why = "a layer input left the split-f16 domain |x| < 4094 (the state became non-finite)"
        if not fallback:
            raise FloatingPointError("pstl: " + why + "; re-run with chain_waves = 8")
        self.use_exact_fp32(why)
        return True

    def _stl_event(self, kind, row_evals):
        """HIP events (recorded on the launch stream) around one STL launch, kept for bench.py; None when not tracing."""
        if getattr(self, "trace_stl", None) is None:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), int(row_evals))
        ev[0].record()
        self.trace_stl.setdefault(kind, []).append(ev)
        return ev

    # ---- A1 ----
    def encode(self, sb, need_rect=True, save=False):
        """save=True (training with --joint): a fourth return value, the per-token activations the encoders' backward pass
        needs (dict tok_in (T,48), tok_h1, tok_h2 (T,256), tok_out (T,32); T = bs*(K+4), see pstl_encode_scene_saved)."""
        dev = sb.device
        feature = torch.empty(sb.bs, ffi.FEAT, dtype=torch.float32, device=dev)
        base_p = torch.empty(sb.bs, ffi.HID, dtype=torch.float32, device=dev)
        base_r = torch.empty(sb.bs, ffi.HID, dtype=torch.float32, device=dev) if (need_rect and self.w.has_rect) else None
        cfg = sb.cfg(2)
        if save:
            T = sb.bs * (sb.K + 4)
            sv = dict(tok_in=torch.empty(T, 48, dtype=torch.float32, device=dev),
                      tok_h1=torch.empty(T, ffi.HID, dtype=torch.float32, device=dev),
                      tok_h2=torch.empty(T, ffi.HID, dtype=torch.float32, device=dev),
                      tok_out=torch.empty(T, 32, dtype=torch.float32, device=dev))
            ffi.check(self.L.pstl_encode_scene_saved(ctypes.byref(cfg), ffi.ptr(self.w.packed), ffi.ptr(sb.ego0),
                                                     ffi.ptr(sb.neighbors), ffi.ptr(sb.lanes[0]), ffi.ptr(sb.lanes[1]),
                                                     ffi.ptr(sb.lanes[2]), ffi.ptr(sb.ids[0]), ffi.ptr(sb.ids[1]),
                                                     ffi.ptr(sb.ids[2]), ffi.ptr(feature), ffi.ptr(base_p), ffi.ptr(base_r),
                                                     ffi.ptr(sv["tok_in"]), ffi.ptr(sv["tok_h1"]), ffi.ptr(sv["tok_h2"]),
                                                     ffi.ptr(sv["tok_out"]), ffi.stream()), "encode_scene_saved")
            return feature, base_p, base_r, sv
        work = torch.empty(self.L.pstl_encode_scene_work_floats(ctypes.byref(cfg)), dtype=torch.float32, device=dev)
        ffi.check(self.L.pstl_encode_scene(ctypes.byref(cfg), ffi.ptr(self.w.packed), ffi.ptr(sb.ego0),
                                           ffi.ptr(sb.neighbors), ffi.ptr(sb.lanes[0]), ffi.ptr(sb.lanes[1]),
                                           ffi.ptr(sb.lanes[2]), ffi.ptr(sb.ids[0]), ffi.ptr(sb.ids[1]),
                                           ffi.ptr(sb.ids[2]), ffi.ptr(work), ffi.ptr(feature), ffi.ptr(base_p),
                                           ffi.ptr(base_r), ffi.stream()), "encode_scene")
        return feature, base_p, base_r

    # ---- A3-A5, A7 ----
    def fill_normal(self, sb, steps, step, seed, out=None):
        """The N(0,1) values the kernels draw for reverse step `step` under in-kernel noise (step == steps: x_T)."""
        if out is None:
            out = torch.empty(sb.N, ffi.CTRL, dtype=torch.float32, device=sb.device)
        cfg = sb.cfg(steps, 0, 0, seed)
        ffi.check(self.L.pstl_fill_normal(ctypes.byref(cfg), int(step), ffi.ptr(out), ffi.stream()), "fill_normal")
        return out

    def rollout(self, sb, base_policy, x, noise, steps, n_emit=0, clip=False, guidance=None, coeffs=None, seed=None):
        """x (N,40) is updated in place from x_T to x_0 (un-normalised).  noise (steps-1,N,40) supplied by the caller,
        or seed != None: the kernels draw the noise themselves (Philox keyed by seed and global row).
        Returns emit (n_emit,N,40): the last n_emit entries of the reference's normalised diff_full list."""
        dev = sb.device
        if coeffs is None:   # (kept per schedule: building it copies three host tensors to the device -- not capturable, not free)
            ck = self.__dict__.setdefault("_coeffs", {})
            if (int(steps), str(dev)) not in ck:
                ck.clear()
                ck[(int(steps), str(dev))] = diffusion_coeffs(steps, dev)
            coeffs = ck[(int(steps), str(dev))]
        beta, alpha, alpha_hat = coeffs
        # host copy of beta (sqrt(beta_i) is a by-value argument of the guidance launch); taking it from the device tensor
        # would block the host on the GPU at every rollout, so it is cached per schedule
        cache = self.__dict__.setdefault("_beta_host", {})
        key = (int(steps), beta.data_ptr())
        if key not in cache:
            cache.clear()
            cache[key] = beta.detach().cpu()
        beta_host = cache[key]
        flags = ffi.PSTL_FLAG_CLIP if clip else 0
        if guidance and guidance.get("maximize", False):
            flags |= ffi.PSTL_FLAG_MAXIMIZE
        if seed is not None:
            flags |= ffi.PSTL_FLAG_RNG
            noise = None
        cfg = sb.cfg(steps, flags, self.chain_waves, seed or 0)
        tb = self.w.tbias(steps)
        emit = torch.empty(max(n_emit, 1), sb.N, ffi.CTRL, dtype=torch.float32, device=dev)
        if guidance and guidance.get("enabled", False):
            nit = int(guidance["niters"])
            lr = float(guidance["lr"])
            neg_step = (ctypes.c_float * nit)(*[-lr / (1 - 0.9 ** (j + 1)) for j in range(nit)])
            bc2 = (ctypes.c_float * nit)(*[math.sqrt(1 - 0.999 ** (j + 1)) for j in range(nit)])
            work = torch.empty(3, sb.N, ffi.CTRL, dtype=torch.float32, device=dev) if nit > 1 else None

        def plain(hi, lo, mu_only=0):
            ev = None
            if self.trace is not None and hi > lo:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            _launch(hi, lo, mu_only)
            if ev is not None:
                ev[1].record()
                self.trace.append((ev[0], ev[1], hi - lo + 1, sb.N))

        def _launch(hi, lo, mu_only):
            ffi.check(self.L.pstl_rollout(ctypes.byref(cfg), ffi.ptr(self.w.packed), ffi.ptr(base_policy), ffi.ptr(tb),
                                          ffi.ptr(sb.stlp), ffi.ptr(sb.hl), ffi.ptr(beta), ffi.ptr(alpha),
                                          ffi.ptr(alpha_hat), ffi.ptr(noise), int(hi), int(lo), int(mu_only),
                                          ffi.ptr(x), ffi.ptr(emit) if n_emit > 0 else ffi.ptr(self.debug_buf), int(n_emit),
                                          ffi.stream()), "rollout")

        i = steps - 1
        while i >= 1:
            if not guidance_triggered(i, steps, guidance):
                lo = i
                while lo - 1 >= 1 and not guidance_triggered(lo - 1, steps, guidance):
                    lo -= 1
                plain(i, lo)
                i = lo - 1
            else:
                plain(i, i, mu_only=1)
                z = noise[steps - 1 - i] if (noise is not None and i > 1) else None
                eo = emit[n_emit - i] if (n_emit > 0 and i <= n_emit) else None
                gev = self._stl_event("guidance", sb.N * nit)
                ffi.check(self.L.pstl_guidance_step(ctypes.byref(cfg), ffi.ptr(sb.s0), ffi.ptr(sb.nei_prep),
                                                    ffi.ptr(sb.lane_prep), ffi.ptr(sb.stlp), ffi.ptr(sb.hl),
                                                    ffi.ptr(sb.valid), ctypes.c_float(sb.grad_scale), nit, neg_step, bc2,
                                                    ctypes.c_float(float(beta_host[i])), int(i), ffi.ptr(z), ffi.ptr(x),
                                                    ffi.ptr(work), ffi.ptr(eo), ffi.stream()), "guidance_step")
                if gev is not None:
                    gev[1].record()
                i -= 1
        return emit[:n_emit]

    # ---- A6, A8-A10 ----
    def score(self, sb, controls, select=False, all3=False, states=None):
        """controls (reps,N,40) physical units -> scores (reps,N) [, scores3 (3,reps,N)] [, best controls/score/idx]."""
        dev = sb.device
        src = controls if controls is not None else states
        reps = src.shape[0]
        scores = torch.empty(reps, sb.N, dtype=torch.float32, device=dev)
        s3 = torch.empty(3, reps, sb.N, dtype=torch.float32, device=dev) if all3 else None
        sel_c = torch.empty(sb.N, ffi.CTRL, dtype=torch.float32, device=dev) if select else None
        sel_s = torch.empty(sb.N, dtype=torch.float32, device=dev) if select else None
        sel_i = torch.empty(sb.N, dtype=torch.int32, device=dev) if select else None
        cfg = sb.cfg(2)
        sev = self._stl_event("score", sb.N * reps)
        ffi.check(self.L.pstl_stl_forward(ctypes.byref(cfg), ffi.ptr(sb.s0), ffi.ptr(controls), ffi.ptr(states), int(reps),
                                          ffi.ptr(sb.nei_prep), ffi.ptr(sb.lane_prep), ffi.ptr(sb.stlp), ffi.ptr(sb.hl),
                                          ffi.ptr(scores), ffi.ptr(s3), ffi.ptr(sel_c), ffi.ptr(sel_s),
                                          ffi.ptr(sel_i, torch.int32), ffi.stream()), "stl_forward")
        if sev is not None:
            sev[1].record()
        out = {"scores": scores}
        if all3:
            out["scores3"] = s3
        if select:
            out.update(sel_controls=sel_c, sel_scores=sel_s, sel_idx=sel_i)
        return out

    def score_grad(self, sb, controls, dscore=None):
        dev = sb.device
        g = torch.empty(sb.N, ffi.CTRL, dtype=torch.float32, device=dev)
        sc = torch.empty(sb.N, dtype=torch.float32, device=dev)
        cfg = sb.cfg(2)
        ffi.check(self.L.pstl_stl_backward(ctypes.byref(cfg), ffi.ptr(sb.s0), ffi.ptr(controls), ffi.ptr(sb.nei_prep),
                                           ffi.ptr(sb.lane_prep), ffi.ptr(sb.stlp), ffi.ptr(sb.hl), ffi.ptr(dscore),
                                           ffi.ptr(g), ffi.ptr(sc), ffi.stream()), "stl_backward")
        return sc, g

    def trajs(self, sb, controls):
        out = torch.empty(sb.N, ffi.T + 1, 4, dtype=torch.float32, device=sb.device)
        cfg = sb.cfg(2)
        ffi.check(self.L.pstl_generate_trajs(ctypes.byref(cfg), ffi.ptr(sb.s0), ffi.ptr(controls), ffi.ptr(out),
                                             ffi.stream()), "generate_trajs")
        return out

    # ---- A11 ----
    def refine(self, sb, base_rect, init_controls, scores, diverse=True, clip_rect=False):
        dev = sb.device
        flags = (0 if diverse else ffi.PSTL_FLAG_NO_MERGE) | (ffi.PSTL_FLAG_CLIP_RECT if clip_rect else 0)
        cfg = sb.cfg(2, flags, self.chain_waves)
        pooled = torch.empty(sb.bs, 3, cfg.n_shards, ffi.CTRL, dtype=torch.float32, device=dev) if diverse else None
        out = torch.empty(sb.N, ffi.CTRL, dtype=torch.float32, device=dev)
        ffi.check(self.L.pstl_refine(ctypes.byref(cfg), ffi.ptr(self.w.packed), ffi.ptr(base_rect), ffi.ptr(sb.stlp),
                                     ffi.ptr(sb.hl), ffi.ptr(init_controls), ffi.ptr(scores), ffi.ptr(pooled),
                                     ffi.ptr(out), ffi.stream()), "refine")
        return out

    def select_plan(self, sb, scores, controls):
        """The closed loop's choice (reference nusc_sim.py:677-683): device tensor (first w, first a, score, domain-flag bits) of
        the best lane-keeping sample of the one scene in `sb`."""
        out = torch.empty(4, dtype=torch.float32, device=sb.device)
        cfg = sb.cfg(2)
        ffi.check(self.L.pstl_select_plan(ctypes.byref(cfg), ffi.ptr(scores), ffi.ptr(controls), ffi.ptr(self.w.status[2:3]),
                                          ffi.ptr(out), ffi.stream()), "select_plan")
        return out

    def metrics(self, sb, scores, want_mask=False):
        """Integer numerators/denominators of acc and scene_acc (device tensor of 8 int64; no host sync here)."""
        counts = torch.empty(8, dtype=torch.int64, device=sb.device)
        mask = torch.empty(sb.N, dtype=torch.uint8, device=sb.device) if want_mask else None
        cfg = sb.cfg(2)
        ffi.check(self.L.pstl_reduce_metrics(ctypes.byref(cfg), ffi.ptr(scores), ffi.ptr(sb.valid),
                                             ffi.ptr(counts, torch.int64), ffi.ptr(mask, torch.uint8), ffi.stream()),
                  "reduce_metrics")
        return counts, mask

    # ---- N4: trajectory optimisation (the data-augmentation loop) ----
    def trajopt(self, sb, params, iters, lr, thres=0.01, reg_loss=10.0, work=None, first_iter=0,
                global_valid_sum=None, global_rows=None):
        """`iters` Adam iterations on params (N,40) (controls in physical units, updated IN PLACE) under the traj-opt
        loss (reference nusc_train.py:1302-1325, compute_trajopt_loss_lite :287-300) -- one launch for all of them.
        A run may be split over several calls: pass the returned `work` (Adam m, v) back with first_iter = iterations
        already done.  Returns (scores of the iterate the last step started from (N,), work)."""
        dev = sb.device
        N = sb.N
        f32 = np.float32
        vsum = float(sb.valid.sum().item()) if global_valid_sum is None else float(global_valid_sum)
        rows = N if global_rows is None else int(global_rows)
        c = f32(max(f32(f32(vsum) / f32(rows)), f32(1e-3)))                  # clip(mean(valid), 1e-3)
        grad_scale = float(f32(f32(1.0) / c) / f32(rows))
        reg_scale = float(f32(reg_loss) / f32(rows * ffi.T))
        ks = range(first_iter + 1, first_iter + iters + 1)
        neg_step = torch.tensor([-lr / (1 - 0.9 ** k) for k in ks], dtype=torch.float32, device=dev)
        bc2 = torch.tensor([math.sqrt(1 - 0.999 ** k) for k in ks], dtype=torch.float32, device=dev)
        resume = 1 if work is not None else 0
        if work is None:
            work = torch.empty(3, ffi.CTRL, N, dtype=torch.float32, device=dev)
        scores = torch.empty(N, dtype=torch.float32, device=dev)
        cfg = sb.cfg(2)
        ffi.check(self.L.pstl_trajopt(ctypes.byref(cfg), ffi.ptr(sb.s0), ffi.ptr(sb.nei_prep), ffi.ptr(sb.lane_prep),
                                      ffi.ptr(sb.stlp), ffi.ptr(sb.hl), ffi.ptr(sb.valid), ctypes.c_float(thres),
                                      ctypes.c_float(grad_scale), ctypes.c_float(reg_scale), int(iters), ffi.ptr(neg_step),
                                      ffi.ptr(bc2), resume, ffi.ptr(params), ffi.ptr(work), ffi.ptr(scores), ffi.stream()),
                  "trajopt")
        return scores, work

    # ---- --refinement: Adam over per-row mixing weights of eight control sequences ----
    REFINEMENT_LIST_IDX = (0, 50, 80, 85, 90, 95, 98)     # k_d_list[8] of the reference (nusc_train.py:1052-1055)

    def refinement(self, sb, controls, clist, iters=50, lr=0.3, thres=0.0005, trace=False):
        """controls (N,40), clist (n_list,N,40): the rollout's normalised list (entry 0 = x_T).  Returns the refined controls
        (N,40) [, d loss / d lambda per iteration (iters,N,8)]: the reference's --refinement block (nusc_train.py:1034-1071),
        all iterations in one launch."""
        dev = sb.device
        n_list = int(clist.shape[0])
        if max(self.REFINEMENT_LIST_IDX) >= n_list:
            raise IndexError("--refinement reads entry %d of the rollout's list, which has %d entries "
                             "(--diffusion_steps >= 99)" % (max(self.REFINEMENT_LIST_IDX), n_list))
        ks = range(1, iters + 1)
        neg_step = torch.tensor([-lr / (1 - 0.9 ** k) for k in ks], dtype=torch.float32, device=dev)
        bc2 = torch.tensor([math.sqrt(1 - 0.999 ** k) for k in ks], dtype=torch.float32, device=dev)
        cfg = sb.cfg(2)
        work = torch.empty(self.L.pstl_refinement_work_floats(ctypes.byref(cfg)), dtype=torch.float32, device=dev)
        out = torch.empty(sb.N, ffi.CTRL, dtype=torch.float32, device=dev)
        tr = torch.empty(iters, sb.N, 8, dtype=torch.float32, device=dev) if trace else None
        idx = (ctypes.c_int32 * 7)(*self.REFINEMENT_LIST_IDX)
        clist = clist.reshape(n_list, sb.N, ffi.CTRL)
        ffi.check(self.L.pstl_refinement(ctypes.byref(cfg), ffi.ptr(sb.s0), ffi.ptr(sb.nei_prep), ffi.ptr(sb.lane_prep),
                                         ffi.ptr(sb.stlp), ffi.ptr(sb.hl), ffi.ptr(sb.valid), ctypes.c_float(thres),
                                         ctypes.c_float(sb.grad_scale), int(iters), ffi.ptr(neg_step), ffi.ptr(bc2),
                                         ffi.ptr(controls), ffi.ptr(clist), n_list, idx, ffi.ptr(work), ffi.ptr(out),
                                         ffi.ptr(tr), ffi.stream()), "refinement")
        return (out, tr) if trace else out

    # ---- N2: post-sampling diversity metrics ----
    def diversity(self, sb, controls, scores):
        """std / hull volume / entropies / occupancy area / ADE / FDE of the final controls (N,40) (physical units):
        what the reference computes on the CPU after its timer (nusc_train.py:1107-1140).  Returns device tensors
        per_mode (bs,3,8) f64, per_scene (bs,2) f32, totals (12) f64 (see include/pstl_hip.h); no host sync."""
        dev = sb.device
        if getattr(self, "_alphas", None) is None or self._alphas.device != dev:
            self._alphas = torch.linspace(0.0, 1.0, 11).to(dev)       # utils.py:406, computed on the host as there
        per_mode = torch.empty(sb.bs, 3, 8, dtype=torch.float64, device=dev)
        per_scene = torch.empty(sb.bs, 2, dtype=torch.float32, device=dev)
        totals = torch.empty(12, dtype=torch.float64, device=dev)
        cfg = sb.cfg(2)
        ffi.check(self.L.pstl_diversity(ctypes.byref(cfg), ffi.ptr(sb.s0), ffi.ptr(sb.ego_traj),
                                        int(sb.ego_traj.shape[-1]), ffi.ptr(controls), ffi.ptr(scores), ffi.ptr(sb.valid),
                                        ffi.ptr(self._alphas), ffi.ptr(per_mode, torch.float64), ffi.ptr(per_scene),
                                        ffi.ptr(totals, torch.float64), ffi.stream()), "diversity")
        return per_mode, per_scene, totals

    # ---- A12: the timed region ----
    def sampling_region(self, sb, steps, x_T, noise, rect_head=False, multi_cands=None, refinenet=True, guidance=None,
                        n_rolls=None, diverse=True, full_list=False, coeffs=None, want_scores3=True, seed=None,
                        diversity=False, clip_rect=False, use_rect=True, refinement_iters=None, want_counts=True):
        """x_T (N,40) and noise (steps-1,N,40) supplied by the caller (parity), or seed != None: x_T and all noise are
        drawn by the kernels (x_T / noise arguments ignored)."""
        out = {}
        feature, base_p, base_r = self.encode(sb, need_rect=rect_head)
        out["feature_scene"] = feature
        if seed is not None:
            x = self.fill_normal(sb, steps, steps, seed)
        else:
            x = x_T.clone()
        mc = multi_cands if (rect_head and multi_cands is not None) else 0
        if refinement_iters and rect_head and use_rect:
            full_list = True         # --refinement mixes entries 0..98 of the list
        n_emit = steps if full_list else max(mc, 1)
        emit = self.rollout(sb, base_p, x, noise, steps, n_emit=n_emit, clip=bool(rect_head), guidance=guidance,
                            coeffs=coeffs, seed=seed)
        if full_list:
            out["controls_list"] = emit
        controls = emit[-1]
        if rect_head and use_rect:      # use_rect=False: --not_use_rect (clip / candidate list of --rect_head stay)
            if mc > 0:
                r = self.score(sb, emit[-mc:].contiguous(), select=True)
                out.update(cand_scores=r["scores"], sel_scores=r["sel_scores"], sel_idx=r["sel_idx"],
                           sel_controls=r["sel_controls"])
                controls, best = r["sel_controls"], r["sel_scores"]
            else:
                best = self.score(sb, controls.reshape(1, sb.N, ffi.CTRL))["scores"][0]
            if refinenet:
                controls = self.refine(sb, base_r, controls, best, diverse=diverse, clip_rect=clip_rect)
                out["rect_controls"] = controls
            for ri in range(n_rolls or 0):
                sc = self.score(sb, controls.reshape(1, sb.N, ffi.CTRL))["scores"][0]
                controls = self.refine(sb, base_r, controls, sc, diverse=diverse, clip_rect=clip_rect)
                out["roll%d_scores" % ri] = sc
                out["roll%d_controls" % ri] = controls
            if refinement_iters:
                controls = self.refinement(sb, controls.contiguous(), emit, iters=int(refinement_iters))
                out["refinement_controls"] = controls
        fin = self.score(sb, controls.reshape(1, sb.N, ffi.CTRL), all3=want_scores3)
        out.update(final_controls=controls, final_scores=fin["scores"][0])
        if want_counts:      # (the closed-loop caller only needs the best sample: three launches less per planning step)
            out["counts"] = self.metrics(sb, fin["scores"][0])[0]
        if diversity:   # std / hull volume / entropies / area / ADE / FDE (the reference: on the CPU, after its timer)
            pm, ps, tot = self.diversity(sb, controls, fin["scores"][0])
            out.update(div_per_mode=pm, div_per_scene=ps, div_totals=tot)
        if want_scores3:   # the three formulas before mode selection (what compute_stl_dense returns as scores_list)
            out["final_scores3"] = fin["scores3"][:, 0]
        return out


class GraphCapture:
    """`body()` -- a fixed sequence of launches on fixed device buffers -- captured ONCE in a HIP graph and replayed: for
    launch-bound callers (small batches: the closed loop's 192 rows are ~50 launches of 5-40 us).  What changes between replays
    must live in device memory the body reads: inputs in static buffers, the noise seed and the guidance-loss scale in a
    pstl_dyn block (`SceneBatch(..., dyn=...)`, ABI 4).  The library's launches go to the capturing stream like any other work
    (ffi.stream() is torch's current stream); tensors the body allocates come from the graph's private pool and stay valid --
    `out` is what the last replay produced.  Same kernels, same arguments, same order as the eager calls: identical results."""

    def __init__(self, body, warmup=2):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # eager warm-up on a side stream: allocations, function attributes, host-side caches
            for _ in range(warmup):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = body()

    def replay(self):
        self.graph.replay()
        return self.out


class DynBlock:
    """A pstl_dyn in device memory, set from the host: set(seed[, grad_scale]) queues ONE 16-byte copy on the current stream.
    The copy is asynchronous and may run long after set() returns (behind a replay of milliseconds), so every call writes a
    pinned mirror of its own -- a ring of them, each guarded by an event recorded behind its copy and waited for before the
    slot is written again.  (One mirror rewritten per call let back-to-back set() + replay() pairs run several replays on the
    LAST seed: the earlier copies read the mirror after the host had moved on.)"""
    RING = 8

    def __init__(self, device):
        self.hosts = [torch.zeros(4, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
        self.events = [None] * self.RING
        self.dev = torch.zeros(4, dtype=torch.float32, device=device)
        self.calls = 0
        self.grad_scale = 0.0

    def set(self, seed, grad_scale=None):
        seed = int(seed) & (2 ** 64 - 1)
        slot = self.calls % self.RING
        self.calls += 1
        if self.events[slot] is not None:
            self.events[slot].synchronize()       # the copy that last read this mirror has run
        if grad_scale is not None:
            self.grad_scale = float(grad_scale)
        h = self.hosts[slot].numpy()
        h[0:2].view("uint32")[:] = (seed & 0xffffffff, seed >> 32)
        h[2] = self.grad_scale
        self.dev.copy_(self.hosts[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[slot] = ev


def adam_schedule(lr, betas, n_steps, first=1):
    """(n_steps, 2) float32: -lr / (1 - beta1^t), sqrt(1 - beta2^t) for t = first ... first + n_steps - 1, computed in double
    precision with the very Python expressions of torch/optim/adam.py (_single_tensor_adam) and rounded to float32 as torch
    rounds its Python scalars: the table pstl_adam_step indexes with its device step counter."""
    b1, b2 = float(betas[0]), float(betas[1])
    out = np.empty((int(n_steps), 2), np.float32)
    for k in range(int(n_steps)):
        t = first + k
        bias_correction1 = 1 - b1 ** t
        bias_correction2 = 1 - b2 ** t
        step_size = lr / bias_correction1
        out[k, 0] = np.float32(-step_size)
        out[k, 1] = np.float32(bias_correction2 ** 0.5)
    return out


class DeviceAdam:
    """torch.optim.Adam (defaults: no weight decay, no amsgrad -- the reference's optimiser, nusc_train.py:1233) on the device
    path: the moments, the table of per-step scalars and the step counter live in device memory and ONE launch of
    pstl_adam_step updates every tensor (csrc/adam_kernels.hip; bit for bit torch's float32 update, tests/test_adam_core_hostsim.py
    and tests/test_gpu_train_step.py).  Nothing of a step is passed by value, so a training step that ends in step() can be
    captured in a HIP graph and replayed.

    params: the live parameter tensors (contiguous float32, on one device), updated IN PLACE; their autograd version counters
    are bumped so that Net.packed() / PackedWeights notice the change.  `adopt(optimizer, params)` wraps a caller's
    torch.optim.Adam (the reference's training loop builds one): its hyper-parameters are taken over -- anything but Adam's
    defaults besides lr / betas / eps is refused --, the lr of its first param group is re-read at every step (schedulers work),
    and its own step() is never called."""

    TABLE = None      # steps of scalars per table; None: as many as the scalars need to reach their limits (see table_len)
    TABLE_MAX = 1 << 18

    @classmethod
    def table_len(cls, betas):
        """Steps after which float32(1 - beta^t) is 1.0 for both betas (beta^t < 2^-25): from there on -lr / (1 - beta1^t) and
        sqrt(1 - beta2^t) no longer change, and the kernel, which clamps its index to the table's last entry, stays exact for any
        number of replays without the host touching the table (defaults: 17 330 steps, 139 KB).  Betas too close to 1 for
        TABLE_MAX entries get a table that is extended from the host instead (note_replays)."""
        if cls.TABLE:
            return int(cls.TABLE)
        n = 2
        for b in betas:
            if 0.0 < b < 1.0:
                n = max(n, int(math.ceil(-25.0 * math.log(2.0) / math.log(b))) + 2)
        return min(n, cls.TABLE_MAX)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("DeviceAdam: no parameters")
        if len(self.params) > ffi.ADAM_MAX_TENSORS:
            raise ValueError("DeviceAdam: at most %d tensors per step (include/pstl_hip.h)" % ffi.ADAM_MAX_TENSORS)
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise ValueError("DeviceAdam: parameters must be contiguous float32 tensors on one device")
        self.device, self.lr, self.betas, self.eps = dev, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.numel = [int(p.numel()) for p in self.params]
        total = sum(self.numel)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps_done = 0                 # host mirror of the device counter (eager steps and replays the caller reports)
        self._table_first, self._table_lr, self.sched = None, None, None
        self.table_steps = self.table_len(self.betas)
        self._table_final = False           # the table's last entry holds the scalars' limits: nothing to extend, ever
        self._numel_arr = (ctypes.c_int64 * len(self.numel))(*self.numel)
        self._p_arr = (ctypes.c_void_p * len(self.params))(*[p.data_ptr() for p in self.params])
        self._ensure_table()

    @classmethod
    def adopt(cls, optimizer, params):
        """The DeviceAdam behind a caller's torch.optim.Adam (created on first use, kept on the optimizer object)."""
        if isinstance(optimizer, cls):
            return optimizer
        mine = getattr(optimizer, "_pstl_device_adam", None)
        ids = [id(p) for p in params]
        if mine is None or mine._ids != ids:
            if not isinstance(optimizer, torch.optim.Adam):
                raise TypeError("train_step: the optimiser must be torch.optim.Adam (the reference's) or a DeviceAdam, not %s"
                                % type(optimizer).__name__)
            g = optimizer.param_groups[0]
            for key, want in (("weight_decay", 0), ("amsgrad", False), ("maximize", False)):
                if g.get(key, want) != want:
                    raise NotImplementedError("DeviceAdam: torch.optim.Adam(%s=%r) is not built (the reference uses the defaults)"
                                              % (key, g[key]))
            mine = cls(params, lr=g["lr"], betas=g["betas"], eps=g["eps"])
            mine._ids = ids
            mine._group = g
            optimizer._pstl_device_adam = mine
        mine.lr = float(mine._group["lr"])
        return mine

    def _ensure_table(self):
        """The table of per-step scalars covers the step about to be taken (and the lr it was built for is still the lr)."""
        inside = self.sched is not None and self._table_first <= self.steps_done + 1 < self._table_first + self.table_steps
        past_final = self.sched is not None and self._table_final and self.steps_done + 1 >= self._table_first
        if self.sched is None or self._table_lr != self.lr or not (inside or past_final):
            first = self.steps_done + 1
            tab = adam_schedule(self.lr, self.betas, self.table_steps, first)
            self._table_final = bool(tab[-1, 0] == np.float32(-self.lr) and tab[-1, 1] == np.float32(1.0))
            tab = torch.from_numpy(tab)
            if self.sched is None:
                self.sched = tab.to(self.device)
            else:
                self.sched.copy_(tab)        # in place: a captured step keeps reading this very buffer
            self._table_first, self._table_lr = first, self.lr
            self.step_dev.fill_(0)           # the counter indexes THIS table

    def note_replays(self, n=1):
        """A captured step() was replayed n times: the device counter moved, this brings the host's mirror along and -- outside
        any capture -- extends the table of per-step scalars if the next step would leave it.  With the default table (table_len:
        it ends at the scalars' limits) that never happens and calling this late costs nothing; with a table cut short (betas
        very close to 1) call it at least every `table_steps` replays, or the steps past the table run on its last entry."""
        self.steps_done += int(n)
        self._ensure_table()

    def step(self, grads):
        """One optimiser step from `grads` (one tensor per parameter, same shapes)."""
        if len(grads) != len(self.params):
            raise ValueError("DeviceAdam.step: %d gradients for %d parameters" % (len(grads), len(self.params)))
        gs = []
        for p, g in zip(self.params, grads):
            g = ffi.f32(g, self.device)
            if g.numel() != p.numel():
                raise ValueError("DeviceAdam.step: a gradient's size does not match its parameter's")
            gs.append(g)
        self._ensure_table()
        for i, p in enumerate(self.params):     # (a parameter whose storage was replaced, e.g. by load_state_dict(assign=True))
            self._p_arr[i] = p.data_ptr()
        g_arr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        ffi.check(ffi.lib().pstl_adam_step(len(gs), self._p_arr, g_arr, self._numel_arr, ffi.ptr(self.exp_avg),
                                           ffi.ptr(self.exp_avg_sq), ffi.ptr(self.sched), self.table_steps,
                                           ffi.ptr(self.step_dev, torch.int32), ctypes.c_float(1 - self.betas[0]),
                                           ctypes.c_float(self.betas[1]), ctypes.c_float(1 - self.betas[1]),
                                           ctypes.c_float(self.eps), ffi.stream()), "adam_step")
        self.steps_done += 1
        for p in self.params:      # (no arithmetic: the kernel wrote through the raw pointer; packed-weight caches key on this)
            torch.autograd.graph.increment_version(p)
        del gs


def diversity_from_totals(totals, nt=ffi.T):
    """The reference's printed diversity numbers from the 12 additive totals of pstl_diversity (sums over shards add)."""
    t = [float(v) for v in (totals.tolist() if hasattr(totals, "tolist") else totals)]
    nan = float("nan")
    nv, nm, ns = t[2], t[7], t[10]
    out = dict(std=t[0] / nv if nv else nan, vol=t[1] / nv if nv else nan, ent_s=t[3] / nm if nm else nan,
               ent_w=t[4] / (nm * nt) if nm else nan, ent_a=t[5] / (nm * nt) if nm else nan,
               area=t[6] / nm if nm else nan, ade=t[8] / ns if ns else nan, fde=t[9] / ns if ns else nan)
    out["ent_wa"] = out["ent_w"] + out["ent_a"]
    return out


def acc_from_counts(counts):
    """acc / scene_acc exactly as mask_mean computes them in float32 (nusc_train.py:23-27,332,342)."""
    c = [int(v) for v in counts.tolist()]
    f32 = np.float32
    acc = f32(f32(c[0]) / f32(c[2])) / max(f32(f32(c[1]) / f32(c[2])), f32(1e-2))
    sacc = f32(f32(c[3]) / f32(c[5])) / max(f32(f32(c[4]) / f32(c[5])), f32(1e-2))
    return float(acc), float(sacc)


class RectTrainer:
    """RefineNet training step under the STL loss (reference nusc_train.py:1400-1427,1522-1525 with
    compute_policy_loss :370-478 for --rect_head without --diverse_loss): forward with saved activations, STL adjoint,
    head/MLP backward -> gradients of the six rect_net tensors in the reference layout.  The optimiser is Adam as the
    reference builds it (torch.optim.Adam over `net.rect_net.parameters()`), run on the device path: DeviceAdam / pstl_adam_step."""

    NAMES = ("rect_net.0.weight", "rect_net.0.bias", "rect_net.2.weight", "rect_net.2.bias", "rect_net.4.weight",
             "rect_net.4.bias")
    ENCODERS = ("ego_encoder", "neighbor_encoder", "lane_encoder")
    ENCODER_IN = (6, 7, 45)

    @classmethod
    def joint_names(cls, merge):
        """Every tensor that receives a gradient under --joint (reference nusc_train.py:1230-1231: Adam over
        net.parameters(); policy_net gets none and Adam skips it)."""
        names = list(cls.NAMES)
        for e in cls.ENCODERS:
            names += ["%s.%d.%s" % (e, i, t) for i in (0, 2, 4) for t in ("weight", "bias")]
        if merge:
            names += ["merge_net.%d.%s" % (i, t) for i in (0, 2, 4) for t in ("weight", "bias")]
        return tuple(names)

    def __init__(self, sampler):
        self.sm = sampler
        self.L = sampler.L

    def loss_and_grads(self, sb, feature, base_rect, w2, w3, init_controls, prev_scores, e7=None, stl_weight=1.0, merge=None,
                       clip_rect=False, joint=None):
        """init_controls (N,40) physical units, prev_scores (N,) (both detached in the reference).  w2, w3: the live
        rect_net.2.weight / rect_net.4.weight tensors.  Returns (loss tensor, rect_controls, scores, {name: grad}).
        e7 = None: config 5, loss = mask_mean(relu(thres - score), valid), plain rect_net input.
        e7 = dict(stl_weight, diversity_weight[, diversity_scale, rect_reg_loss, detach]): the --diverse_loss objective
        loss_stl*stl_weight + loss_reg*rect_reg_loss + loss_diversity (reference nusc_train.py:442-467) with the
        merge_net architecture; self.last holds the individual terms.
        merge: rect_net sees init + merge_net max-pool (the reference: --diverse_loss without --no_arch, nusc_model.py:185);
        default = (e7 is not None).  clip_rect: --clip_rect (nusc_model.py:230-233); the interval head already keeps a
        refined control inside its bounds, so the clip is the identity up to rounding and passes the gradient through.
        joint: None, or dict(params={name: live tensor} for rect_net.0.weight and the .2/.4 weights of the three encoders,
        saved=the fourth return value of Sampler.encode(save=True)): --joint, the gradient dict also holds the encoders'
        tensors and, with merge, merge_net's (RectTrainer.joint_names)."""
        dev = sb.device
        N = sb.N
        objective_e7 = e7 is not None
        merge = objective_e7 if merge is None else bool(merge)
        # (KEEP_DH1: the encoders' backward continues from dH1 in pstl_refine_backward's work buffer; otherwise it is never written)
        cfg = sb.cfg(2, (0 if merge else ffi.PSTL_FLAG_NO_MERGE) | (ffi.PSTL_FLAG_KEEP_DH1 if joint is not None else 0),
                     self.sm.chain_waves)
        cfg_fwd = sb.cfg(2, (0 if merge else ffi.PSTL_FLAG_NO_MERGE) | (ffi.PSTL_FLAG_CLIP_RECT if clip_rect else 0),
                         self.sm.chain_waves)
        h1 = torch.empty(N, ffi.HID, dtype=torch.float32, device=dev)
        h2 = torch.empty(N, ffi.HID, dtype=torch.float32, device=dev)
        pre = torch.empty(N, ffi.CTRL, dtype=torch.float32, device=dev)
        rect = torch.empty(N, ffi.CTRL, dtype=torch.float32, device=dev)
        pooled = torch.empty(sb.bs, 3, cfg.n_shards, ffi.CTRL, dtype=torch.float32, device=dev) if merge else None
        ffi.check(self.L.pstl_refine_train_forward(ctypes.byref(cfg_fwd), ffi.ptr(self.sm.w.packed), ffi.ptr(base_rect),
                                                   ffi.ptr(sb.stlp), ffi.ptr(sb.hl), ffi.ptr(init_controls),
                                                   ffi.ptr(prev_scores), ffi.ptr(pooled), ffi.ptr(rect), ffi.ptr(h1),
                                                   ffi.ptr(h2), ffi.ptr(pre), ffi.stream()), "refine_train_forward")
        scores = self.sm.score(sb, rect.reshape(1, N, ffi.CTRL))["scores"][0]
        dscore = torch.empty(N, dtype=torch.float32, device=dev)
        parts = torch.empty(256, dtype=torch.float32, device=dev)
        stl_w = float(e7["stl_weight"]) if objective_e7 else float(stl_weight)
        ffi.check(self.L.pstl_loss_grad(ctypes.byref(cfg), ffi.ptr(scores), ffi.ptr(sb.valid),
                                        ctypes.c_float(sb.grad_scale * stl_w), ffi.ptr(dscore), ffi.ptr(parts), ffi.stream()),
                  "loss_grad")
        # loss = mean(relu(thres - score) * valid) / clip(mean(valid), 1e-2): grad_scale is exactly (1/clip)/N
        loss = parts.sum() * (sb.grad_scale * stl_w)
        dctrl_extra = None
        if objective_e7:
            groups = sb.bs * 3 * cfg.n_shards
            group_div = torch.empty(groups, dtype=torch.float32, device=dev)
            reg_out = torch.empty(2, dtype=torch.float32, device=dev)
            reg_work = torch.empty(512, dtype=torch.float64, device=dev)
            dctrl_extra = torch.empty(N, ffi.CTRL, dtype=torch.float32, device=dev)
            dscore_div = torch.empty(N, dtype=torch.float32, device=dev)
            div_w, reg_w = float(e7["diversity_weight"]), float(e7.get("rect_reg_loss", 0.0))
            ffi.check(self.L.pstl_diversity_loss(ctypes.byref(cfg), ffi.ptr(rect), ffi.ptr(init_controls), ffi.ptr(scores),
                                                 ctypes.c_float(float(e7.get("diversity_scale", 1.0))), ctypes.c_float(div_w),
                                                 int(bool(e7.get("detach", False))), ctypes.c_float(reg_w),
                                                 ffi.ptr(group_div), ffi.ptr(reg_out), ffi.ptr(reg_work, torch.float64),
                                                 ffi.ptr(dctrl_extra), ffi.ptr(dscore_div), ffi.stream()), "diversity_loss")
            dscore = dscore + dscore_div
            loss_div = -group_div.mean() * div_w
            self.last = dict(loss_stl=loss, loss_diversity=loss_div, loss_reg=reg_out[0], group_div=group_div)
            loss = loss + loss_div + reg_out[0] * reg_w
        _, dctrl = self.sm.score_grad(sb, rect, dscore=dscore)
        if dctrl_extra is not None:
            dctrl = dctrl + dctrl_extra
        work = torch.empty(self.L.pstl_train_work_floats(ctypes.byref(cfg)), dtype=torch.float32, device=dev)
        shapes = {"rect_net.0.weight": (ffi.HID, ffi.FEAT + 47), "rect_net.0.bias": (ffi.HID,),
                  "rect_net.2.weight": (ffi.HID, ffi.HID), "rect_net.2.bias": (ffi.HID,),
                  "rect_net.4.weight": (ffi.CTRL, ffi.HID), "rect_net.4.bias": (ffi.CTRL,)}
        g = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in self.NAMES}
        w2c, w3c = ffi.f32(w2.detach(), dev), ffi.f32(w3.detach(), dev)   # named: a converted copy must outlive the launch
        bev = None
        if getattr(self.sm, "trace_bwd", None) is not None:   # bench.py: HIP events around RefineNet's backward (launch stream)
            bev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), N)
            bev[0].record()
            self.sm.trace_bwd.append(bev)
        ffi.check(self.L.pstl_refine_backward(ctypes.byref(cfg), ffi.ptr(w2c),
                                              ffi.ptr(w3c), ffi.ptr(feature), ffi.ptr(sb.stlp),
                                              ffi.ptr(sb.hl), ffi.ptr(init_controls), ffi.ptr(pooled), ffi.ptr(prev_scores),
                                              ffi.ptr(h1), ffi.ptr(h2), ffi.ptr(pre), ffi.ptr(dctrl), ffi.ptr(work),
                                              ffi.ptr(g["rect_net.0.weight"]), ffi.ptr(g["rect_net.0.bias"]),
                                              ffi.ptr(g["rect_net.2.weight"]), ffi.ptr(g["rect_net.2.bias"]),
                                              ffi.ptr(g["rect_net.4.weight"]), ffi.ptr(g["rect_net.4.bias"]),
                                              ffi.stream()), "refine_backward")
        if bev is not None:
            bev[1].record()
        if joint is not None:
            g.update(self._joint_grads(sb, cfg, work, joint["params"], joint["saved"], init_controls, merge))
        return loss, rect, scores, g

    def _joint_grads(self, sb, cfg, refine_work, params, sv, init_controls, merge):
        """pstl_encoder_backward (+ pstl_merge_backward): continues from the work buffer pstl_refine_backward just filled."""
        dev = sb.device
        g = {}
        arr = {}
        keep = []   # converted copies must outlive the launches
        PtrArr = ctypes.c_void_p * 3

        def grads_of(idx, shape_of):
            ts = []
            for e, nin in zip(self.ENCODERS, self.ENCODER_IN):
                w = torch.empty(shape_of(nin), dtype=torch.float32, device=dev)
                b = torch.empty(shape_of(nin)[0], dtype=torch.float32, device=dev)
                g["%s.%d.weight" % (e, idx)] = w
                g["%s.%d.bias" % (e, idx)] = b
                ts.append((w, b))
            return PtrArr(*[t[0].data_ptr() for t in ts]), PtrArr(*[t[1].data_ptr() for t in ts])

        dw0, db0 = grads_of(0, lambda nin: (ffi.HID, nin))
        dw1, db1 = grads_of(2, lambda nin: (ffi.HID, ffi.HID))
        dw2, db2 = grads_of(4, lambda nin: (32, ffi.HID))
        for idx in (2, 4):
            ws = [ffi.f32(params["%s.%d.weight" % (e, idx)].detach(), dev) for e in self.ENCODERS]
            keep += ws
            arr[idx] = PtrArr(*[w.data_ptr() for w in ws])
        w1r = ffi.f32(params["rect_net.0.weight"].detach(), dev)
        work = torch.empty(self.L.pstl_encoder_backward_work_floats(ctypes.byref(cfg)), dtype=torch.float32, device=dev)
        dfused = torch.empty(sb.N, ffi.CTRL, dtype=torch.float32, device=dev) if merge else None
        ffi.check(self.L.pstl_encoder_backward(ctypes.byref(cfg), ffi.ptr(refine_work), ffi.ptr(w1r), arr[2], arr[4],
                                               ffi.ptr(sv["tok_in"]), ffi.ptr(sv["tok_h1"]), ffi.ptr(sv["tok_h2"]),
                                               ffi.ptr(sv["tok_out"]), ffi.ptr(work), dw0, db0, dw1, db1, dw2, db2,
                                               ffi.ptr(dfused), ffi.stream()), "encoder_backward")
        if merge:
            mw = torch.empty(self.L.pstl_merge_backward_work_floats(ctypes.byref(cfg)), dtype=torch.float32, device=dev)
            shapes = {0: (32, ffi.CTRL), 2: (32, 32), 4: (ffi.CTRL, 32)}
            for idx in (0, 2, 4):
                g["merge_net.%d.weight" % idx] = torch.empty(shapes[idx], dtype=torch.float32, device=dev)
                g["merge_net.%d.bias" % idx] = torch.empty(shapes[idx][0], dtype=torch.float32, device=dev)
            ffi.check(self.L.pstl_merge_backward(ctypes.byref(cfg), ffi.ptr(self.sm.w.packed), ffi.ptr(init_controls),
                                                 ffi.ptr(dfused), ffi.CTRL, ffi.ptr(mw),
                                                 ffi.ptr(g["merge_net.0.weight"]), ffi.ptr(g["merge_net.0.bias"]),
                                                 ffi.ptr(g["merge_net.2.weight"]), ffi.ptr(g["merge_net.2.bias"]),
                                                 ffi.ptr(g["merge_net.4.weight"]), ffi.ptr(g["merge_net.4.bias"]),
                                                 ffi.stream()), "merge_backward")
        del keep   # (same-stream caching allocator: releasing the converted copies after the launches is ordered)
        return g

    def train_step(self, sb, params, optimizer, steps, x_T=None, noise=None, seed=None, multi_cands=5, coeffs=None,
                   group=None, e7=None, stl_weight=1.0, merge=None, clip_rect=False, joint=False, domain_check="eager"):
        """One optimisation step of config 5 on one batch shard: sampling under no-grad (rollout, candidate scoring and
        selection), RefineNet forward/backward under the STL loss, gradient all-reduce over the ranks (the loss is a
        mean over the GLOBAL batch, so per-rank gradients simply add), optimizer.step() on the caller's parameters.
        `params`: dict name -> live torch Parameter/Tensor for RectTrainer.NAMES (the weights used by the kernels are
        re-packed from them by the caller after the step).
        joint (--joint): `params` holds RectTrainer.joint_names(merge) -- the encoders and merge_net are trained too.
        optimizer: the caller's torch.optim.Adam (adopted: hyper-parameters taken over, never stepped) or a DeviceAdam; the step
        itself is pstl_adam_step on the device path.
        domain_check: "eager" reads the split-f16 domain word (one 4-byte copy = one synchronisation) before the optimiser
        consumes the gradients and repeats the step on the exact kernels when it is set; "deferred" reads nothing -- for a caller
        that captures the step in a HIP graph (no synchronisation inside a capture) and calls `sampler.check_chain_domain()`
        where it synchronises anyway; a step that left the domain has then already been applied."""
        if domain_check not in ("eager", "deferred"):
            raise ValueError("domain_check: 'eager' or 'deferred'")
        import torch.distributed as dist
        sm = self.sm
        use_merge = (e7 is not None) if merge is None else bool(merge)
        names = self.joint_names(use_merge) if joint else self.NAMES
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if e7 is not None and world > 1:
            # loss_diversity is a mean over the groups of the GLOBAL batch: with equal shards every rank contributes 1/world
            # (loss_reg's mask_mean is normalised per shard; its weight --rect_reg_loss is 0 in every README command)
            e7 = dict(e7, diversity_weight=float(e7["diversity_weight"]) / world,
                      rect_reg_loss=float(e7.get("rect_reg_loss", 0.0)) / world)

        def forward_backward():
            saved = None
            if joint:
                feature, base_p, base_r, saved = sm.encode(sb, need_rect=True, save=True)
            else:
                feature, base_p, base_r = sm.encode(sb, need_rect=True)
            x = sm.fill_normal(sb, steps, steps, seed) if seed is not None else x_T.clone()
            emit = sm.rollout(sb, base_p, x, noise, steps, n_emit=max(multi_cands, 1), clip=True, coeffs=coeffs, seed=seed)
            r = sm.score(sb, emit[-multi_cands:].contiguous(), select=True)
            return self.loss_and_grads(sb, feature, base_r, params["rect_net.2.weight"], params["rect_net.4.weight"],
                                       r["sel_controls"], r["sel_scores"], e7=e7, stl_weight=stl_weight, merge=merge,
                                       clip_rect=clip_rect, joint=dict(params=params, saved=saved) if joint else None)

        loss, rect, scores, g = forward_backward()
        # Domain of the split-f16 chains (rollout AND the saved RefineNet forward), read BEFORE the gradients reach the
        # optimiser: a layer input beyond the half range leaves plausible-looking garbage, not NaNs.  One 4-byte copy; the
        # step is then repeated on the exact-fp32 kernels (same noise: supplied, or the same Philox seed) and the sampler stays
        # there -- the caller sees `self.sm.chain_fallback` and a RuntimeWarning.
        # With several ranks the decision is the job's, not the rank's: the flags are MAX-reduced first, so that
        # every rank repeats the step and switches arithmetic together -- gradients of two arithmetics are never mixed, and
        # the shard split does not change which kernels ran.
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if domain_check == "deferred" and multi:
            raise ValueError("train_step: the deferred domain check is for single-process (graph-captured) steps")
        hit = domain_check == "eager" and sm.chain_waves in (0, 16, 2) and sm.w.chain_overflowed(clear=True)
        if multi:
            # EVERY rank joins this all-reduce, whatever arithmetic it is on (a rank already on the exact kernels used
            # to skip it, and its peers' flag all-reduce then paired with its gradient all-reduce): word 0 = "a split-f16 launch of
            # mine left the domain", word 1 = "I am on the exact kernels already".  Either one anywhere moves the whole job to the
            # exact kernels for this step and the rest of the run: gradients of two arithmetics are never mixed.
            flag = torch.tensor([1 if hit else 0, 0 if sm.chain_waves in (0, 16, 2) else 1], dtype=torch.int32)
            if dist.get_backend(group) != "gloo":
                flag = flag.to(sb.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            f = [int(v) for v in flag.tolist()]
            hit = sm.chain_waves in (0, 16, 2) and bool(f[0] or f[1])
        if hit:
            sm.use_exact_fp32("a layer input left the split-f16 domain |x| < 4094 (the state became non-finite)" +
                              (" on some rank" if multi else ""))
            loss, rect, scores, g = forward_backward()
        if multi:
            flat = torch.cat([g[k].reshape(-1) for k in names] + [loss.reshape(1)])
            if dist.get_backend(group) == "gloo":       # host tensors (ranks sharing a device in the tests); RCCL: in place
                host = flat.cpu()
                dist.all_reduce(host, group=group)
                flat = host.to(flat.device)
            else:
                dist.all_reduce(flat, group=group)      # 145 704 gradients (--joint: up to 387 056) + the loss: one all-reduce
            o = 0
            for k in names:
                n = g[k].numel()
                g[k] = flat[o:o + n].reshape(g[k].shape)
                o += n
            loss = flat[o]
        # the optimiser step on the device path: one launch of pstl_adam_step over the trained tensors -- a
        # caller's torch.optim.Adam is adopted (its hyper-parameters and lr), never stepped; the caller re-packs afterwards
        for k in names:      # (what loss.backward() leaves behind in the reference's loop: the gradients stay inspectable)
            params[k].grad = g[k]
        DeviceAdam.adopt(optimizer, [params[k] for k in names]).step([g[k] for k in names])
        return loss, scores
