#!/bin/bash
# copies the summaries of tools/measure_r04.sh (gpurun_out/measure4/) into profiles/ (tracked)
S=gpurun_out/measure4; D=profiles
cp $S/bench_default.json $D/r04_bench_default.json
cp $S/r04_*_kernel_stats.csv $D/
cp $S/r04_pmc.json $D/r04_pmc.json
cp $S/latency_single_frame.txt $D/r04_latency_single_frame.txt
cp $S/wvb_phases.txt $D/r04_stage_b_phase_split.txt
ls -la $D | grep r04
