#!/bin/bash
# After `gpurun -- 'ROUND=roundN bash tools/gpu_round_evidence.sh'`: copies the summaries gpurun merged back under gpurun_out/final/
# into profiles/ (tracked) and regenerates tests/golden/parity_envelopes.json from the measured statistics.  The kernel sources of the
# tree must be the ones the evidence ran on: the envelope file is stamped with bench.csrc_sha16() of the tree, and the bench lines
# carry the hash they ran on -- the script refuses when the two differ.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
O=gpurun_out/final
ran=$(python3 -c "import json;print(json.loads(open('$O/profiles/${ROUND:-round3}_bench_cube.json').read().strip().splitlines()[-1])['config']['csrc_sha16'])")
tree=$(python3 -c "import bench;print(bench.csrc_sha16())")
if [ "$ran" != "$tree" ]; then echo "evidence ran on kernel sources $ran, the tree is $tree: not applied"; exit 1; fi
cp $O/profiles/* profiles/
python3 tools/make_parity_envelopes.py $O/parity_stats.json tests/golden/parity_envelopes.json
mkdir -p profiles/data && cp $O/parity_stats.json profiles/data/${ROUND:-round3}_parity_stats.json
echo "applied evidence of kernel sources $tree"
