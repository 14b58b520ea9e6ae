#!/bin/bash
# isolated kernel durations of the headline workload (one call in flight, host stages inline); usage: iso_stats.sh <tag> [ENV=VAL ...]
R=$PWD; TAG=$1; shift; O=$R/gpurun_out/iso_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
env "$@" FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python $R/bench.py --workload cascade --also none --steps 6 --warmup 2 --frames-per-step 128 --no-cpu-baseline --no-probe > $O/bench.json 2> $O/err.txt
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/t
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
calls=max(int(r["Calls"]) for r in rows if "prefilter" in r["Name"])
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("$TAG: per call us %.1f (%d calls)" % (tot/1e3/calls, calls))
for r in rows[:10]:
    n=r["Name"].replace("(anonymous namespace)::","")[:36]
    print("  %-36s x%.2f avg %7.1f min %7.1f max %7.1f  per-call %7.1f"%(n,int(r["Calls"])/calls,float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3,float(r["TotalDurationNs"])/1e3/calls))
PY
