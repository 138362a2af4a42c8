#!/bin/bash
# A/B timing of build variants of one translation unit ON the GPU box (nothing is built here and shipped):
#   tools/dbg/ab.sh "<bench args>" name1:unit:"<extra hipcc flags>" name2:unit:"<flags>" ...      unit = mlp | stl | div
# Each variant is compiled into /tmp/pstl_variants/ (the other objects are the in-tree ones), then bench.py is run
# against it through tools/dbg/with_lib.py.  "base" = the in-tree library.
root=$(cd "$(dirname "$0")/../.." && pwd)
c=$root/pstl_diffusion_policy_amd/csrc
out=/tmp/pstl_variants
mkdir -p $out
bargs=$1; shift
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stl_kernels']; print('$1: step %.2f ms, chain %.3f ms (frac %.3f), guidance %.3f, score %.3f, sat %.6f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], s.get('guidance',{}).get('ms_per_step',0), s.get('score',{}).get('ms_per_step',0), d['stl_sat_rate']))"; }
cd $root
python3 bench.py --no_cpu_baseline $bargs 2>/dev/null | tail -1 | line base
for v in "$@"; do
  n=${v%%:*}; rest=${v#*:}; unit=${rest%%:*}; flags=${rest#*:}
  objs="$c/stl_kernels.o $c/mlp_kernels.o $c/train_kernels.o $c/diversity_kernels.o $c/stl_program.o"
  if [ "$unit" = "mlp" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xarch_device -mllvm=-misched-prera-direction=topdown $flags -c $c/mlp_kernels.hip -o $out/v_$n.o || { echo "$n: build failed"; continue; }
    objs=${objs/$c\/mlp_kernels.o/$out\/v_$n.o}
  elif [ "$unit" = "div" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -c $c/diversity_kernels.hip -o $out/v_$n.o || { echo "$n: build failed"; continue; }
    objs=${objs/$c\/diversity_kernels.o/$out\/v_$n.o}
  else
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xarch_device -mllvm=-misched=gcn-iterative-ilp $flags -c $c/stl_kernels.hip -o $out/v_$n.o || { echo "$n: build failed"; continue; }
    objs=${objs/$c\/stl_kernels.o/$out\/v_$n.o}
  fi
  hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out/libpstl_$n.so
  python3 tools/dbg/with_lib.py $out/libpstl_$n.so bench.py --no_cpu_baseline $bargs 2>/dev/null | tail -1 | line $n
done
