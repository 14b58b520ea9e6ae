#!/bin/bash
# copies the summaries of tools/measure_r03.sh (gpurun_out/measure3/) into profiles/ (tracked)
S=gpurun_out/measure3; D=profiles
cp $S/bench_default.json $D/r03_bench_default.json
cp $S/r03_*_kernel_stats.csv $D/
cp $S/r03_pmc.json $D/r03_pmc.json
cp $S/latency_single_frame.txt $D/r03_latency_single_frame.txt
cp $S/wvb_phases.txt $D/r03_stage_b_phase_split.txt
ls -la $D | grep r03
