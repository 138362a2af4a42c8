import cProfile, pstats, io, sys, contextlib
sys.path.insert(0, "/root/repo")
from pstl_diffusion_policy_amd import nusc_train as nt
argv = ["-e", "e7_ours", "--diffusion", "--stl_weight", "0.0", "--load_stlp", "--rect_head", "--flex", "--diverse_loss",
        "--test", "-P", "e7_ours", "--run_sampling_test", "--skip_nusc_load", "--viz_correct", "--allow_random_init",
        "--multi_cands", "5", "--kernel_noise", "--n_trials", "40"]
buf = io.StringIO()
pr = cProfile.Profile()
with contextlib.redirect_stdout(buf):
    pr.enable()
    md = nt.main(argv)
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
