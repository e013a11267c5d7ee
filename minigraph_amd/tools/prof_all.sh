#!/bin/bash
# The evidence set of a round, all from ONE tree (run on the GPU box through gpurun):  prof_all.sh <tag>
#   <tag>_kernel_stats_isolated.txt / _pipelined.txt   rocprofv3 --kernel-trace of the bench command (one chunk in flight, everything on one stream / the real pipeline)
#   <tag>_kernel_stats_devchain.txt                    the same, isolated, with graph chaining + gap list on the device
#   <tag>_pmc_fetch.txt / _pmc_write.txt / _pmc.json   FETCH_SIZE and WRITE_SIZE passes (separate runs, no tracing)
#   <tag>_sq_counters.txt                              two SQ counter passes
# kernel-trace and --pmc are never combined in one run.
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ulimit -c 0
WDA=${PROF_WORKDIR:+--workdir $PROF_WORKDIR}   # PROF_WORKDIR=<dir>: the synthetic workload (graph, reads, graph image) is generated once and reused by every run below
B="python bench.py --steps 1 --warmup 1 --no-cpu --resident-steps 0 --one-placement --no-asm --no-small --no-file-out --no-rank-share $WDA"
out=gpurun_out
parts=${PROF_PARTS:-iso pipe dev pmc sq}   # PROF_PARTS="iso pmc": a subset (each part is one or two bench runs under rocprofv3, about 70 s each)
has() { case " $parts " in *" $1 "*) return 0;; esac; return 1; }
has iso && {
MGA_PIPE=1 MGA_WFA_SIDE=0 bash minigraph_amd/tools/prof_trace.sh ${tag}_iso MGA_PIPE=1 MGA_WFA_SIDE=0 -- --steps 1 --warmup 1 --no-cpu --resident-steps 0 --one-placement --no-asm --no-small --no-file-out --no-rank-share $WDA > /dev/null 2>&1
mv $out/${tag}_iso_kernel_stats.txt $out/${tag}_kernel_stats_isolated.txt
}
has pipe && {
bash minigraph_amd/tools/prof_trace.sh ${tag}_pipe -- --steps 2 --warmup 1 --no-cpu --resident-steps 0 --one-placement --no-asm --no-small --no-file-out --no-rank-share $WDA > /dev/null 2>&1
mv $out/${tag}_pipe_kernel_stats.txt $out/${tag}_kernel_stats_pipelined.txt
}
has dev && {
bash minigraph_amd/tools/prof_trace.sh ${tag}_dev MGA_PIPE=1 MGA_WFA_SIDE=0 -- --steps 1 --warmup 1 --no-cpu --resident-steps 0 --one-placement --no-asm --no-small --no-file-out --no-rank-share --threads 8 $WDA > /dev/null 2>&1
mv $out/${tag}_dev_kernel_stats.txt $out/${tag}_kernel_stats_devchain_isolated.txt
}
has pmc && {
for c in FETCH_SIZE WRITE_SIZE; do
	rm -rf $out/prof_${tag}_$c
	rocprofv3 --pmc $c -d $out/prof_${tag}_$c -o pmc -- $B > /dev/null 2> $out/${tag}_pmc_$c.err
done
F=$(find $out/prof_${tag}_FETCH_SIZE -name "*.db" | head -1); W=$(find $out/prof_${tag}_WRITE_SIZE -name "*.db" | head -1)
python minigraph_amd/tools/prof_summary.py --pmc "$F" "FETCH_SIZE pass of: $B" > $out/${tag}_pmc_fetch.txt
python minigraph_amd/tools/prof_summary.py --pmc "$W" "WRITE_SIZE pass of: $B" > $out/${tag}_pmc_write.txt
python minigraph_amd/tools/prof_summary.py --pmc-json "$F" "$W" "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, no tracing) of $B on MI355X, $tag: 3 Gbp graph, 125000 reads per step; KB summed over launches, uncorrected (the accesses of the WFA kernels are byte / dword wide, not the 16 B per lane streams whose FETCH_SIZE reads half: MI355X_MICROARCH.md, HBM)" > $out/${tag}_pmc.json
}
has sq && {
rm -rf $out/prof_${tag}_sq1 $out/prof_${tag}_sq2
MGA_PIPE=1 MGA_WFA_SIDE=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $out/prof_${tag}_sq1 -o pmc -- $B > /dev/null 2> $out/${tag}_sq1.err
# PROF_SQ_LIGHT=1: the first counter set only (the instruction counts behind roofline.valu_busy); the second set and the device-chaining pass are skipped
[ "${PROF_SQ_LIGHT:-0}" = 1 ] || MGA_PIPE=1 MGA_WFA_SIDE=0 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU -d $out/prof_${tag}_sq2 -o pmc -- $B > /dev/null 2> $out/${tag}_sq2.err
# (third pass: the same first counter set with graph chaining + gap list on the device -- k_gchain_p1 / p2 / p3, k_plan -- listed behind the others: a kernel's first pass wins)
rm -rf $out/prof_${tag}_sq3
[ "${PROF_SQ_LIGHT:-0}" = 1 ] || MGA_PIPE=1 MGA_WFA_SIDE=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $out/prof_${tag}_sq3 -o pmc -- $B --threads 8 > /dev/null 2> $out/${tag}_sq3.err
S1=$(find $out/prof_${tag}_sq1 -name "*.db" | head -1); S2=$(find $out/prof_${tag}_sq2 -name "*.db" 2>/dev/null | head -1); S3=$(find $out/prof_${tag}_sq3 -name "*.db" 2>/dev/null | head -1)
python minigraph_amd/tools/prof_summary.py --sq "$S1" "$S2" "$S3" "MGA_PIPE=1 MGA_WFA_SIDE=0 $B (durations are inflated by the counter collection)" > $out/${tag}_sq_counters.txt 2> $out/${tag}_sq_summary.err
}
# the pipelined trace keeps its timestamps: union of kernel intervals against the sum of durations over the second half of the run (the steps after the warm-up)
T=$(find $out/prof_${tag}_pipe -name "*kernel_trace.csv" 2>/dev/null | head -1)
[ -n "$T" ] && python minigraph_amd/tools/trace_overlap.py "$T" --window 0.5,1.0 --title "pipelined bench step of $tag, steps after the warm-up" > $out/${tag}_pipe_overlap.txt 2>&1
rm -rf $out/prof_${tag}_* $out/prof_${tag}   # the raw rocprofv3 outputs (hundreds of MB) stay on the GPU box: gpurun copies at most 64 MiB back
ls -la $out/${tag}_*
