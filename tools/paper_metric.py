#!/usr/bin/env python3
"""The paper's own metric through the drop-in surface: md["time"] of run_sampling_test -- the wall time of
the region reference nusc_train.py:957-1105 per batch, with a device sync on both sides -- for the README command lines
"Ours" and "Ours+guidance" (reference README.md:114,120) at the reference's defaults (-b 128 = 24 576 rows, 100 diffusion
steps, 8 neighbours), through `python -m pstl_diffusion_policy_amd.nusc_train` 's main() on synthetic scenes with random-init
weights (no nuScenes cache or checkpoints offline).  GPU only.
    python tools/paper_metric.py [--batches 8] [--kernel_noise]
Prints one JSON object: per configuration the median / min / first-batch time and the trajectories per second it implies,
beside the paper's Table-I numbers (different GPU, real data, no device sync in the reference's timer: context, not target)."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PAPER = {"ours": 0.174, "ours_guidance": 0.786}     # docs/resources/table_1.png, "Time (s)" per batch of 24 576 rows

COMMON = ["-e", "e7_ours", "--diffusion", "--stl_weight", "0.0", "--load_stlp", "--rect_head", "--flex", "--diverse_loss",
          "--test", "-P", "e7_ours", "--run_sampling_test", "--skip_nusc_load", "--viz_correct", "--allow_random_init"]
CONFIGS = {
    "ours": COMMON + ["--multi_cands", "5"],
    "ours_guidance": COMMON + ["--multi_cands", "10", "--guidance", "--guidance_before", "10", "--guidance_niters", "1",
                               "--guidance_lr", "0.01", "--n_rolls", "3", "--other"],
}


def main():
    import contextlib
    import io
    args = sys.argv[1:]
    batches = int(args[args.index("--batches") + 1]) if "--batches" in args else 8
    extra = (["--kernel_noise"] if "--kernel_noise" in args else []) + (["--no_graph"] if "--no_graph" in args else [])
    from pstl_diffusion_policy_amd import nusc_train as nt
    out = {"rows_per_batch": 128 * 64 * 3, "batches": batches, "noise": "in-kernel Philox" if extra else "torch.randn_like per step"}
    for name, argv in CONFIGS.items():
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            md = nt.main(argv + ["--n_trials", str(batches - 1)] + extra)
        t = md.hist["time"]
        steady = t[1:] if len(t) > 1 else t
        med = statistics.median(steady)
        out[name] = {"argv": " ".join(argv + extra), "time_s_median": med, "time_s_min": min(steady), "time_s_first_batch": t[0],
                     "trajectories_per_s": out["rows_per_batch"] / med, "acc": md("acc"), "scene_acc": md("scene_acc"),
                     "paper_time_s": PAPER[name], "paper_hardware": "unspecified NVIDIA GPU (reference README.md:35)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
