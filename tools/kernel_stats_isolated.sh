# isolated kernel durations of the headline workload: one call in flight, host stages inline (no overlap between kernels)
R=$PWD; O=$R/gpurun_out/st1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cascade -- python $R/bench.py --workload cascade --also none --steps 6 --warmup 2 --frames-per-step 128 --no-cpu-baseline > $O/cascade.json 2> $O/cascade.err
f=$(find $O/cascade -name "*kernel_stats.csv" | head -1); cp $f $O/cascade_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/cascade_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
calls=max(int(r["Calls"]) for r in rows if "prefilter" in r["Name"])
print("isolated: total kernel ms", tot/1e6, "per call us", tot/1e3/calls)
for r in rows[:8]: print("  %-60s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
