#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> <counter> [<counter> ...]   -- one rocprofv3 PMC pass of a short default bench run
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$TAG -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-batches 0 ${BENCH_ARGS} > $O/pmc_$TAG.json 2> $O/pmc_$TAG.err
