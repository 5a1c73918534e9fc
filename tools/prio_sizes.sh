#!/bin/bash
# the default wave priority schedule against "off" over batch sizes: tools/prio_sizes.sh <workload> <sizes...>
wl=$1; shift
for n in "$@"; do
  echo "== $wl $n"
  python3 tools/ab_bench.py --workload $wl --envs $n --rounds 2 librsrmjx.so@RSR_PRIO_MODE=0 librsrmjx.so librsrmjx.so@RSR_PRIO_MODE=1 librsrmjx.so@RSR_PRIO_MODE=2 | grep "M env"
done
