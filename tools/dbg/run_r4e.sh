cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/r4e
mkdir -p $o
cd /tmp && export TMPDIR=/tmp
cat > /tmp/cl.py <<'P'
import sys
sys.path.insert(0, sys.argv[1])
from pstl_diffusion_policy_amd.nusc_model import init_state_dict
from pstl_diffusion_policy_amd.nusc_sim import closed_loop
recs = closed_loop(init_state_dict(1007), n_sim_steps=30, K=8, S=64, diffusion_steps=100, multi_cands=5, guidance=True, guidance_before=10, guidance_lr=0.04, seed=1, verbose=False, graph=True)
print(sorted(r["latency_s"] for r in recs[3:])[13])
P
rocprofv3 --kernel-trace --output-format csv -d $o/tr -o run -- python3 /tmp/cl.py $GRAFT_REPO_ROOT > $o/cl_rocprof.txt 2>&1
find $o/tr -name "*kernel_trace.csv" -exec cp {} $o/closed_loop_graph_kernel_trace.csv \;
rm -rf $o/tr
cd $GRAFT_REPO_ROOT
python3 tools/dbg/trace_gaps.py $o/closed_loop_graph_kernel_trace.csv 20 > $o/closed_loop_graph_gaps.txt 2>&1
cat $o/closed_loop_graph_gaps.txt; tail -2 $o/cl_rocprof.txt
