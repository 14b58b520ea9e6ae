#!/bin/bash
# copies the summaries of tools/measure_r06.sh (gpurun_out/measure6/) into profiles/ (tracked)
S=gpurun_out/measure6; D=profiles
cp $S/bench_line.json $D/r06_bench_line.json          # the ONE line of stdout (what the driver parses)
cp $S/bench_also.json $D/r06_bench_default.json       # the complete records of the same run (the side file)
cp $S/r06_*_kernel_stats.csv $D/
cp $S/r06_pmc.json $D/r06_pmc.json
cp $S/latency_single_frame.txt $D/r06_latency_single_frame.txt
[ -f $S/ffp15_tail_matrix.txt ] && cp $S/ffp15_tail_matrix.txt $D/r06_ffp15_tail_matrix.txt
ls -la $D | grep r06
