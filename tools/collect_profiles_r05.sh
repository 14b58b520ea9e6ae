#!/bin/bash
# copies the summaries of tools/measure_r05.sh (gpurun_out/measure5/) into profiles/ (tracked)
S=gpurun_out/measure5; D=profiles
cp $S/bench_default.json $D/r05_bench_default.json
cp $S/r05_*_kernel_stats.csv $D/
cp $S/r05_pmc.json $D/r05_pmc.json
cp $S/latency_single_frame.txt $D/r05_latency_single_frame.txt
ls -la $D | grep r05
