#!/bin/bash
# README command lines of the reference through the CLI mirror (synthetic scenes), one line of output each
base="python -m pstl_diffusion_policy_amd.nusc_train --batch_size 8 --n_trials 1 --diffusion_steps 20 --n_neighbors 3 --allow_random_init"
run() { echo "== $*"; $base "$@" 2>&1 | grep -E "^###|Error|error|Traceback|trajopt batch|epoch" | tail -2; }
run -e e5_ddpm --diffusion --stl_weight 0.0 --load_stlp --flex --run_sampling_test --test -P e5_ddpm --skip_nusc_load --viz_correct
run -e e5_ddpm --diffusion --stl_weight 0.0 --load_stlp --flex --run_sampling_test --test -P e5_ddpm --skip_nusc_load --viz_correct --guidance
run -e e7_ours --diffusion --stl_weight 0.0 --load_stlp --rect_head --flex --diverse_loss --multi_cands 5 --test -P e7_ours --run_sampling_test --skip_nusc_load --viz_correct
run -e e7_ours --diffusion --stl_weight 0.0 --load_stlp --rect_head --flex --diverse_loss --multi_cands 10 --test -P e7_ours --run_sampling_test --viz_correct --guidance --guidance_before 10 --guidance_niters 1 --guidance_lr 0.01 --n_rolls 3 --other --skip_nusc_load
run -e e7_ours --diffusion --stl_weight 0.0 --load_stlp --rect_head --flex --diverse_loss --multi_cands 5 --run_sampling_test --test -P e7_ours --skip_nusc_load --viz_correct --no_refinenet
run -e e8_ours_ablation --diffusion --stl_weight 1.0 --load_stlp --rect_head --flex --diversity_weight 0.0 --multi_cands 5 --run_sampling_test --test -P e8_ours_ablation --skip_nusc_load --viz_correct
run -e e4 --diffusion --load_stlp --flex --run_sampling_test --test --norm_stl --time_profile
run -e e7_train --diffusion --stl_weight 0.0 --load_stlp -P e5_ddpm --rect_head --flex --diverse_loss --multi_cands 5 --epochs 1 --print_freq 1
run -e e8_train --diffusion --stl_weight 1.0 --load_stlp --load_tj --rect_head --flex -P e5_ddpm --diversity_weight 0.0 --n_shards 4 --interval --multi_cands 5 --diff_full --epochs 1 --print_freq 1
run -e tj --trajopt_only --traj_opt_iters 30 --load_stlp
run -e e7_joint --diffusion --stl_weight 1.0 --load_stlp -P e5_ddpm --rect_head --flex --diverse_loss --multi_cands 5 --epochs 1 --print_freq 1 --joint
run -e e7_refinement --diffusion --stl_weight 0.0 --load_stlp --rect_head --flex --diverse_loss --multi_cands 5 --test -P e7_ours --run_sampling_test --skip_nusc_load --viz_correct --refinement --diffusion_steps 100
