"""Host-side mirror of the reference model surface for the hot path (reference: nusc_model.py).

`Net` keeps the reference constructor, sub-module names and therefore `state_dict()` keys
({ego,neighbor,lane}_encoder / policy_net / merge_net / rect_net .{0,2,4}.{weight,bias}), so reference checkpoints
load unchanged (`load_state_dict(torch.load(...), strict=not args.rect_head)`, reference nusc_train.py:1215).
The torch modules only HOLD the parameters; `encode_feat`, `forward` (diffusion branch) and `rect_forward` run the
HIP kernels of libpstl_hip.so.  Inference only (the reference runs this path under no_grad / net.eval()).

Scope: the diffusion branch (`--diffusion`, multi-sample rows).  The VAE / BC / gt_data_training branches of the
reference Net.forward (nusc_model.py:128-154) are baselines outside the hot path and raise NotImplementedError.
"""
import ctypes
import os
import warnings

import torch
import torch.nn as nn

from . import ffi
from .engine import PackedWeights


def build_relu_nn(input_dim, output_dim, hiddens):
    """Linear -> ReLU -> ... -> Linear, same layer indices as the reference builder (utils.py:91-101)."""
    dims = [input_dim] + list(hiddens) + [output_dim]
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class SceneFeature(torch.Tensor):
    """The (N,224) feature tensor the reference passes around, carrying the per-scene layer-1 partials the kernels use."""
    @staticmethod
    def wrap(dense, feature_scene, base_policy, base_rect, rows_per_scene):
        t = dense.as_subclass(SceneFeature)
        t.pstl = dict(feature_scene=feature_scene, base_policy=base_policy, base_rect=base_rect,
                      rows_per_scene=rows_per_scene)
        return t


class Net(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        if not getattr(args, "diffusion", False):
            raise NotImplementedError("only the --diffusion model family is on the MI355X hot path")
        if list(args.hiddens) != [256, 256] or args.nt != 20 or args.n_segs != 15:
            raise NotImplementedError("libpstl_hip is specialised for hiddens=[256,256], nt=20, n_segs=15")
        self.output_dim = args.nt * 2
        self.feat_dim = feat_dim = 32
        self.stlp_dim = stlp_dim = 6
        self.time_dim = 32
        self.ego_encoder = build_relu_nn(6, feat_dim, args.hiddens)
        self.neighbor_encoder = build_relu_nn(7, feat_dim, args.hiddens)
        self.lane_encoder = build_relu_nn(args.n_segs * 3, feat_dim, args.hiddens)
        latent_dim = args.nt * 2 + self.time_dim + 1 + stlp_dim
        if getattr(args, "use_init_hint", False):
            raise NotImplementedError("--use_init_hint is a VAE/BC baseline option")
        self.policy_net = build_relu_nn(latent_dim + feat_dim * 7, args.nt * 2, args.hiddens)
        if getattr(args, "rect_head", False):
            if getattr(args, "diverse_loss", False):
                if not getattr(args, "no_arch", False) and getattr(args, "diverse_fuse_type", "add") != "add":
                    raise NotImplementedError("only diverse_fuse_type='add' (the reference default)")
                self.merge_net = build_relu_nn(args.nt * 2, args.nt * 2, [32, 32])
            if list(getattr(args, "rect_hiddens", [256, 256])) != [256, 256]:
                raise NotImplementedError("rect_hiddens must be [256,256]")
            self.rect_net = build_relu_nn(latent_dim - self.time_dim + feat_dim * 7, args.nt * 2, [256, 256])
        self._packed = None
        self._packed_key = None
        # arithmetic of the MLP chains (include/pstl_hip.h, cfg.chain_waves): None = PSTL_CHAIN_WAVES or the default
        # (split-f16); set to 8 (exact fp32 MFMA) when a call left the split-f16 domain (check_domain)
        self.chain_waves = None
        # Domain guard of the split-f16 chains (|layer input| < 4094; outside it the kernels leave plausible garbage, not NaNs,
        # and set a sticky status word).  "eager": every forward / rect_forward / diffusion_rollout call reads that word before
        # it returns (a 4-byte copy: one device synchronisation per call) and, when it is set, warns, repeats the call on the
        # exact-fp32 kernels and stays there.  "deferred": the caller promises to call net.check_domain() where it
        # synchronises anyway and to discard / repeat the work since the last check when it returns True (run_sampling_test
        # does that around its timed region).
        self.domain_check = "eager"

    # ---- kernel-layout weights, re-packed whenever a parameter changed (load_state_dict, optimiser step) ----
    def packed(self):
        # (called a dozen times per batch by the mirror, so the key is cheap -- ~15 us: a direct walk over the six Sequentials'
        # Linear layers instead of nn.Module.parameters() -- and complete: the identity of every storage (a replaced Parameter
        # or sub-module, `p.data = t`, load_state_dict(assign=True), .cuda()) and every version counter (in-place changes:
        # optimiser steps, load_state_dict's copy_) enter it.)
        ps, key = self._pack_key()
        if self._packed is None or key != self._packed_key:
            if not ps[0].is_cuda:
                raise RuntimeError("Net must be on the GPU (net.cuda()) before the HIP path can run")
            old = self._packed_key
            if self._packed is not None and old[0] == key[0] and [n for n, _ in old[1]] == [n for n, _ in key[1]]:
                # the same networks on the same device: only those a parameter of which changed are packed again, in place
                # (an optimiser over rect_net.parameters() touches one of the six)
                changed = {n for (n, sub), (_, sub0) in zip(key[1], old[1]) if sub != sub0}
                self._packed.update({k: v for k, v in self.state_dict().items() if k.split(".")[0] in changed})
            else:
                self._packed = PackedWeights({k: v for k, v in self.state_dict().items()}, ps[0].device)
            self._packed_key = key
        return self._packed

    def _pack_key(self):
        """(parameters in parameters() order, key): the key holds, per network, the version counter and the storage of every
        parameter -- whatever way a parameter changes, its network's entry changes."""
        ps, nets = [], []
        for name in ("ego_encoder", "neighbor_encoder", "lane_encoder", "policy_net", "merge_net", "rect_net"):
            seq = self._modules.get(name)
            if seq is None:
                continue
            mine = [p for layer in seq._modules.values() for p in layer._parameters.values() if p is not None]
            ps += mine
            nets.append((name, (tuple(p._version for p in mine), tuple(p.data_ptr() for p in mine))))
        return ps, (ps[0].device, tuple(nets))

    def chain_arith(self):
        """cfg.chain_waves for this net's weights: the requested arithmetic, or the exact-fp32 kernels when a chain weight is
        outside the split-f16 domain (|w| < 63.9; PackedWeights reads the maximum the packer recorded)."""
        cw = self.chain_waves if self.chain_waves is not None else int(os.environ.get("PSTL_CHAIN_WAVES", "0"))
        pw = self.packed()
        if cw in (0, 16, 2) and not pw.split_f16_ok:
            if not getattr(pw, "_warned", False):
                pw._warned = True
                warnings.warn("pstl: a chain weight is outside the split-f16 domain |w| < %g (policy_net %g, rect_net %g): "
                              "the MLP chains run on the exact-fp32 kernels (chain_waves = 8)"
                              % (ffi.SPLIT_F16_WMAX, pw.chain_wmax["policy_net"], pw.chain_wmax["rect_net"]), RuntimeWarning)
            return 8
        return cw

    def check_domain(self, fallback=True):
        """Did a split-f16 launch since the last check see a layer input outside the half range?  False when all is well.
        Otherwise every result since the last check is undefined: with fallback=True a RuntimeWarning is issued, the net is
        switched to the exact-fp32 kernels for good (chain_waves = 8), the flag cleared and True returned -- the caller repeats
        the work; with fallback=False a FloatingPointError is raised.  Synchronises (one 4-byte copy)."""
        if self.chain_arith() not in (0, 16, 2) or not self.packed().chain_overflowed(clear=True):
            return False
        why = "a layer input left the split-f16 domain |x| < 4094"
        if not fallback:
            raise FloatingPointError("pstl: " + why + "; set net.chain_waves = 8 (exact-fp32 kernels)")
        warnings.warn("pstl: " + why + "; the call is repeated, and later ones run, on the exact-fp32 kernels "
                      "(chain_waves = 8)", RuntimeWarning, stacklevel=3)
        self.chain_waves = 8
        return True

    def _guarded(self, launch):
        """launch(chain_waves) -> result, under the domain guard (see __init__)."""
        out = launch(self.chain_arith())
        if self.domain_check == "eager" and self.check_domain():
            out = launch(8)
        return out

    def hparams(self):
        a = self.args
        return dict(nt=a.nt, dt=a.dt, mul_w_max=a.mul_w_max, mul_a_max=a.mul_a_max,
                    smoothing_factor=a.smoothing_factor, stl_nn_thres=a.stl_nn_thres, ego_L=a.ego_L, ego_W=a.ego_W,
                    refined_nL=a.refined_nL, refined_nW=a.refined_nW, n_segs=a.n_segs, n_shards=a.n_shards,
                    norm_stl=bool(getattr(a, "norm_stl", False)))

    def _encode(self, nn_input):
        pw = self.packed()
        dev = pw.device
        ego0 = ffi.f32(nn_input["ego_traj"][:, 0], dev)
        bs = ego0.shape[0]
        nei = ffi.f32(nn_input["neighbors"], dev)
        lanes = [ffi.f32(nn_input["%slane_wpts" % k], dev) for k in ("curr", "left", "right")]
        ids = [ffi.f32(nn_input["%s_id" % k].reshape(bs), dev) for k in ("curr", "left", "right")]
        feature = torch.empty(bs, ffi.FEAT, dtype=torch.float32, device=dev)
        base_p = torch.empty(bs, ffi.HID, dtype=torch.float32, device=dev)
        base_r = torch.empty(bs, ffi.HID, dtype=torch.float32, device=dev) if pw.has_rect else None
        cfg = ffi.make_cfg(bs, 1, 1, nei.shape[1], 2, self.hparams())
        work = torch.empty(ffi.lib().pstl_encode_scene_work_floats(ctypes.byref(cfg)), dtype=torch.float32, device=dev)
        ffi.check(ffi.lib().pstl_encode_scene(ctypes.byref(cfg), ffi.ptr(pw.packed), ffi.ptr(ego0), ffi.ptr(nei),
                                              ffi.ptr(lanes[0]), ffi.ptr(lanes[1]), ffi.ptr(lanes[2]), ffi.ptr(ids[0]),
                                              ffi.ptr(ids[1]), ffi.ptr(ids[2]), ffi.ptr(work), ffi.ptr(feature),
                                              ffi.ptr(base_p), ffi.ptr(base_r), ffi.stream()), "encode_scene")
        return feature, base_p, base_r

    def encode_feat(self, nn_input, ext=None):
        """(bs,224) scene feature (reference nusc_model.py:55-95)."""
        return self._encode(nn_input)[0]

    def scene_feature(self, nn_input, n_rep):
        """Dense (bs*n_rep,224) feature as the reference returns it (nusc_model.py:104-110), tagged with the per-scene
        layer-1 partials so that later calls do not recompute or re-read it."""
        f, bp, br = self._encode(nn_input)
        bs, k = f.shape
        dense = f.reshape(bs, 1, k).expand(bs, n_rep, k).reshape(-1, k)
        return SceneFeature.wrap(dense, f, bp, br, n_rep)

    def _bases_of(self, feature, nn_input=None):
        if isinstance(feature, SceneFeature) and hasattr(feature, "pstl"):
            return feature.pstl
        raise ValueError("pass the feature object returned by this Net (forward(get_feature=True) / scene_feature)")

    def forward(self, nn_input, ext=None, get_feature=False, prev_feature=None, sample=False, n_randoms=None):
        """One denoiser evaluation: predicted noise (N,nt,2) = policy_net([feature|x|pe(t)|hl|stlp]) + x
        (reference nusc_model.py:97-180, diffusion branch with multi-sample rows)."""
        a = self.args
        if getattr(a, "gt_data_training", False):
            raise NotImplementedError("gt_data_training (e4_ddpm_mono) rows are outside the hot path")
        if n_randoms is None:
            n_randoms = a.n_randoms
        feature = prev_feature if prev_feature is not None else self.scene_feature(nn_input, n_randoms * 3)
        info = self._bases_of(feature)
        pw = self.packed()
        dev = pw.device
        x_in = ffi.f32(ext["noise"], dev)
        N = x_in.shape[0]
        t = int(ext["timestep"].reshape(-1)[0].item())
        steps = max(int(a.diffusion_steps), t + 1)
        stlp = ffi.f32(nn_input["stlp_dense"][:, 0], dev)
        hl = ffi.f32(ext["highlevel"].reshape(N), dev)
        rps = info["rows_per_scene"]
        from .engine import diffusion_coeffs
        beta, alpha, alpha_hat = diffusion_coeffs(steps, dev)

        def launch(chain_waves):
            x = x_in.clone()
            cfg = ffi.make_cfg(N // rps, rps, max(rps // 3, 1), 1, steps, self.hparams(), chain_waves=chain_waves)
            ffi.check(ffi.lib().pstl_rollout(ctypes.byref(cfg), ffi.ptr(pw.packed), ffi.ptr(info["base_policy"]),
                                             ffi.ptr(pw.tbias(steps)), ffi.ptr(stlp), ffi.ptr(hl), ffi.ptr(beta),
                                             ffi.ptr(alpha), ffi.ptr(alpha_hat), ffi.ptr(None), t, t, 2, ffi.ptr(x),
                                             ffi.ptr(None), 0, ffi.stream()), "rollout(eps)")
            return x

        x = self._guarded(launch)
        controls = x.reshape(N, a.nt, 2)
        return (controls, feature) if get_feature else controls

    def rect_forward(self, feature, highlevel, stlp_dense_feat, init_controls, scores, extras=None):
        """RefineNet head (reference nusc_model.py:182-235) with --interval (forced on by --rect_head)."""
        a = self.args
        info = self._bases_of(feature)
        pw = self.packed()
        dev = pw.device
        N = init_controls.shape[0]
        rps = info["rows_per_scene"]
        diverse = bool(getattr(a, "diverse_loss", False)) and not getattr(a, "no_arch", False)
        flags = (0 if diverse else ffi.PSTL_FLAG_NO_MERGE) | (ffi.PSTL_FLAG_CLIP_RECT if getattr(a, "clip_rect", False) else 0)
        n_shards = int(self.hparams().get("n_shards", 4))
        if diverse and rps // 3 != a.n_randoms:
            raise ValueError("merge_net pooling needs sampling_size == n_randoms (reference nusc_model.py:187-196)")
        init = ffi.f32(init_controls.reshape(N, -1), dev)
        pooled = torch.empty(N // rps, 3, n_shards, ffi.CTRL, dtype=torch.float32, device=dev) if diverse else None
        out = torch.empty(N, ffi.CTRL, dtype=torch.float32, device=dev)
        # converted copies get names: every argument is evaluated before the asynchronous launch, and an unnamed copy
        # would be freed (and its block handed to the next copy) before the kernel reads it
        stlp_c, hl_c, sc_c = ffi.f32(stlp_dense_feat, dev), ffi.f32(highlevel.reshape(N), dev), ffi.f32(scores.reshape(N), dev)

        def launch(chain_waves):
            cfg = ffi.make_cfg(N // rps, rps, rps // 3, 1, 2, self.hparams(), flags, chain_waves=chain_waves)
            ffi.check(ffi.lib().pstl_refine(ctypes.byref(cfg), ffi.ptr(pw.packed), ffi.ptr(info["base_rect"]),
                                            ffi.ptr(stlp_c), ffi.ptr(hl_c), ffi.ptr(init), ffi.ptr(sc_c), ffi.ptr(pooled),
                                            ffi.ptr(out), ffi.stream()), "refine")
            return out

        out = self._guarded(launch)
        return out.reshape(N, a.nt, 2)


def init_state_dict(seed=1007, rect_head=True, diverse_loss=True):
    """Random-init weights exactly as `torch.manual_seed(seed); Net(args)` gives them in the reference (same layer
    construction order, hence the same RNG consumption).  Used for synthetic benchmarks: real checkpoints are not
    available offline."""
    import types
    args = types.SimpleNamespace(diffusion=True, hiddens=[256, 256], nt=20, n_segs=15, rect_head=rect_head,
                                 diverse_loss=diverse_loss, no_arch=False, diverse_fuse_type="add",
                                 rect_hiddens=[256, 256], use_init_hint=False)
    rng = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = Net(args)
    torch.random.set_rng_state(rng)
    return {k: v.detach().clone() for k, v in net.state_dict().items()}
