#!/bin/bash
# sweep of the wave-priority policies (RSR_PRIO_MODE) on one box: tools/prio_sweep.sh <workload> <envs> <modes...>
wl=$1; n=$2; shift 2
arms="librsrmjx_base.so"
for m in "$@"; do arms="$arms librsrmjx.so@RSR_PRIO_MODE=$m"; done
python3 tools/ab_bench.py --workload $wl --envs $n --rounds 2 $arms | grep "M env"
