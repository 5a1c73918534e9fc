#!/bin/bash
# Round-end measurement on the GPU box: kernel-trace stats of the default bench, then PMC passes (each in its own run,
# with --kernel-trace only, as the pool requires).  Outputs under gpurun_out/final/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 $R/bench.py --steps 300 --warmup 50 --sub-batches 0 > $O/bench_cube_profiled.json 2> $O/bench_cube_profiled.err
echo "trace done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/pmc_inst -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-batches 0 > $O/pmc_inst.json 2> $O/pmc_inst.err
echo "pmc inst done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d $O/pmc_cyc -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-batches 0 > $O/pmc_cyc.json 2> $O/pmc_cyc.err
echo "pmc cycles done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-batches 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "pmc fetch done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub-batches 0 > $O/pmc_write.json 2> $O/pmc_write.err
echo "pmc write done"
cd $R
for w in cube tshape go2 go2rough; do timeout -k 10 200 python3 bench.py --workload $w --steps 300 --warmup 50 > $O/bench_$w.json 2> $O/bench_$w.err; done
echo "other workloads done"
ls -R $O | head -50
