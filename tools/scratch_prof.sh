R=$PWD; O=$R/gpurun_out/prof_ffp; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
FD_FS_TAIL=1 FD_BATCH_THREADS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --workload ffp15 --also none --steps 3 --warmup 1 --no-cpu-baseline --no-probe --full-out $O/full.json > $O/out.json 2> $O/err.txt
f=$(find $O/st -name "*kernel_stats.csv" | head -1); cp $f $O/stats.csv; rm -rf $O/st
