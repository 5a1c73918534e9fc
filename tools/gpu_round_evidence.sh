#!/bin/bash
# Everything the round's DESIGN / README numbers rest on, in one GPU call: parity statistics on the final kernels (-> the envelopes
# the tests enforce), rocprofv3 kernel stats + PMC passes + bench lines of the workloads (tools/gpu_final_profile.sh), stage shares,
# wave timeline, per-launch step times.  Outputs: gpurun_out/final/.   usage: ROUND=round3 tools/gpu_round_evidence.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-round3}
export ROUND
O=$R/gpurun_out/final
mkdir -p $O/profiles
cd $R
timeout -k 10 600 python3 tools/gpu_parity_stats.py --json $O/parity_stats.json 2>&1 | grep -v amdgpu > $O/profiles/${ROUND}_parity_stats.log; echo "parity stats rc=$?"
bash $R/tools/gpu_final_profile.sh > $O/final_profile.log 2>&1; echo "final_profile rc=$?"
python3 tools/gpu_stage_profile.py --rebuild 2>&1 | grep -v amdgpu > $O/profiles/${ROUND}_stage_shares_cube.log; echo "stage cube done"
python3 tools/gpu_stage_profile.py --go2 2>&1 | grep -v amdgpu > $O/profiles/${ROUND}_stage_shares_go2.log; echo "stage go2 done"
python3 tools/gpu_wave_timeline.py --envs 8192 --units 4 --whole -1 --rebuild 2>&1 | grep -v amdgpu > $O/profiles/${ROUND}_wave_timeline_cube.log; echo "timeline done"
python3 tools/gpu_wave_timeline.py --envs 8192 --units 4 --whole 0 2>&1 | grep -v amdgpu > $O/profiles/${ROUND}_wave_timeline_cube_all_split.log; echo "timeline all-split done"
python3 tools/gpu_step_times.py 2>&1 | grep -v amdgpu > $O/profiles/${ROUND}_step_times_cube.log; echo "step times done"
ls $O/profiles | head -80
