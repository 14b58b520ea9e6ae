R=$PWD; O=$R/gpurun_out/st; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for wl in cascade ffp15; do
  S=10; FP="--frames-per-step 128"; [ $wl = ffp15 ] && S=3 && FP=""
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$wl -- python $R/bench.py --workload $wl --also none --steps $S --warmup 2 $FP --no-cpu-baseline > $O/$wl.json 2> $O/$wl.err
  f=$(find $O/$wl -name "*kernel_stats.csv" | head -1); cp $f $O/${wl}_kernel_stats.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("$O/${wl}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("$wl total kernel ms", tot/1e6)
for r in rows[:14]: print("  %-70s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
done
