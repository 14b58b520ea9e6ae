#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3ffp; mkdir -p $O
timeout 600 python bench.py --workload ffp15 --also none --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_ffp15.json 2> $O/bench_ffp15.err
cut -c1-220 $O/bench_ffp15.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --workload ffp15 --also none --steps 3 --warmup 1 --no-cpu-baseline > $O/st.json 2> $O/st.err
f=$(find $O/st -name "*kernel_stats.csv" | head -1); cp $f $O/ffp15_kernel_stats.csv; rm -rf $O/st
