#!/bin/bash
# the issue-rate microbenchmark (now with the packed instructions, partly empty EXEC masks and the SDWA compare + carry block of the mask build) plain and under the SQ counters:
# is SQ_INSTS_VALU one count per wave-instruction for every instruction kind the WFA rungs use?
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 minigraph_amd/tools/valu_rate.hip -o /tmp/valu_rate 2> $out/r05q_valu_build.err
timeout 120 /tmp/valu_rate > $out/r05q_valu_rate_exec.txt 2>&1
rm -rf $out/prof_r05q_valu
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/prof_r05q_valu -- /tmp/valu_rate > $out/r05q_valu_rate_under_pmc.txt 2> $out/r05q_valu_pmc.err
C=$(find $out/prof_r05q_valu -name "*counter_collection.csv" | head -1)
[ -n "$C" ] && python minigraph_amd/tools/prof_summary.py --calib "$C" > $out/r05q_valu_calibration.txt 2>&1
rm -rf $out/prof_r05q_valu
grep -n "mask\|k_rate<9>\|k_rate<4>\|kernel" $out/r05q_valu_rate_exec.txt $out/r05q_valu_calibration.txt | cut -c1-250 | head -30
cut -c1-200 $out/r05q_valu_calibration.txt | awk 'NR<4 || /k_rate/' | awk '{print}' | tail -45
