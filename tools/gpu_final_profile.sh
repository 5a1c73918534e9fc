#!/bin/bash
# Round-end evidence on the GPU box for the workloads: rocprofv3 kernel-trace stats of a bench run, then the PMC passes
# (each in its own run with --kernel-trace only, as the pool requires), summarised into profiles/<round>_*_<workload>.csv with
# the hash of the kernel sources (tools/rocpd_summary.py), then the plain bench lines.
# usage: ROUND=round3 tools/gpu_final_profile.sh [workloads...]      (default: cube tshape go2 go2rough handstand)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-round3}
O=$R/gpurun_out/final
P=$R/gpurun_out/final/profiles
mkdir -p $O $P
WL=${@:-cube tshape go2 go2rough handstand}
cd /tmp && export TMPDIR=/tmp
db() { find $1 -name "*.db" | head -1; }
for w in $WL; do
  rm -rf $O/trace_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o bench -- python3 $R/bench.py --workload $w --steps 300 --warmup 50 --sub-batches 0 --no-cpu-baseline > $P/${ROUND}_bench_${w}_under_rocprofv3.json 2> $O/trace_$w.err
  python3 $R/tools/rocpd_summary.py kernels $(db $O/trace_$w) $P/${ROUND}_kernel_stats_$w.csv
  rm -rf $O/trace_$w                  # (raw rocprofv3 output is tens of MB per run: only the summaries travel back)
  echo "$w trace done"
  for pass in "inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
              "cyc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
              "fetch FETCH_SIZE" "write WRITE_SIZE"; do
    set -- $pass; tag=$1; shift
    rm -rf $O/pmc_${tag}_$w
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_${tag}_$w -o pmc -- python3 $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --sub-batches 0 > $O/pmc_${tag}_$w.json 2> $O/pmc_${tag}_$w.err
    python3 $R/tools/rocpd_summary.py pmc $(db $O/pmc_${tag}_$w) $P/${ROUND}_pmc_${tag}_$w.csv
    rm -rf $O/pmc_${tag}_$w
  done
  echo "$w pmc done"
done
# the plain bench lines quote the PMC summaries of THIS run (bench.py binds them to the hash of the kernel sources)
cp $P/${ROUND}_pmc_*.csv $R/profiles/
cd $R
for w in $WL; do timeout -k 10 200 python3 bench.py --workload $w --steps 300 --warmup 50 > $P/${ROUND}_bench_$w.json 2> $O/bench_$w.err; echo "$w bench done"; done
# the headline under the driver's protocol (20 timed steps after 5 warm-up steps: the transient after reset), three times
for i in 1 2 3; do timeout -k 10 100 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-batches 0 2>/dev/null; done > $P/${ROUND}_bench_cube_20_5.jsonl
ls $P
