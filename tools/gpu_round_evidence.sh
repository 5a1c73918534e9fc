#!/bin/bash
# Everything the round's DESIGN / README numbers rest on, in one GPU call: rocprofv3 kernel stats + PMC passes + bench lines of
# all four workloads (tools/gpu_final_profile.sh), stage shares, wave timeline, parity statistics.  Outputs: gpurun_out/final/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O/profiles
bash $R/tools/gpu_final_profile.sh > $O/final_profile.log 2>&1; echo "final_profile rc=$?"
cd $R
python3 tools/gpu_stage_profile.py --rebuild 2>&1 | grep -v amdgpu > $O/profiles/round2_stage_shares_cube.log; echo "stage cube done"
python3 tools/gpu_stage_profile.py --go2 2>&1 | grep -v amdgpu > $O/profiles/round2_stage_shares_go2.log; echo "stage go2 done"
python3 tools/gpu_wave_timeline.py --envs 8192 --units 4 --rebuild 2>&1 | grep -v amdgpu > $O/profiles/round2_wave_timeline_cube_units4.log; echo "timeline units4 done"
python3 tools/gpu_wave_timeline.py --envs 8192 --units 1 2>&1 | grep -v amdgpu > $O/profiles/round2_wave_timeline_cube_units1.log; echo "timeline units1 done"
python3 tools/gpu_units_ab.py 2>&1 | grep -v amdgpu > $O/profiles/round2_units_ab_cube.log; echo "units ab done"
ls $O/profiles | head -60
