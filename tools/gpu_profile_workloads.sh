#!/bin/bash
# rocprofv3 kernel-trace stats of the non-headline workloads (lock-step launches only).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_wl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in tshape go2 go2rough; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o bench -- python3 $R/bench.py --workload $w --steps 300 --warmup 50 --sub-batches 0 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  echo "$w done"
done
