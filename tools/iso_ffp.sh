# usage: iso.sh <tag> [env...]: isolated kernel stats of ffp15, prints the prefilter lines
tag=$1; shift
env "$@" FD_BENCH_FFP_SLOTS=1 FD_BATCH_THREADS=1 FD_BATCH_STREAMS=1 timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o p -- python bench.py --workload ffp15 --also none --no-cpu-baseline --no-probe --steps 2 --warmup 1 > /dev/null 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); echo "== $tag"; grep prefilter_group $f | cut -d, -f1-4 | cut -c40-200
