#!/bin/bash
# whole-env units against the default split over small batches: tools/whole_sizes.sh <workload> <sizes...>
wl=$1; shift
for n in "$@"; do
  echo "== $wl $n"
  python3 tools/ab_bench.py --workload $wl --envs $n --rounds 2 librsrmjx.so librsrmjx.so@RSR_WHOLE_ENVS=$n librsrmjx.so@RSR_WHOLE_ENVS=$((n/2)) librsrmjx.so@RSR_WHOLE_ENVS=0 | grep "M env"
done
