#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3c3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_cascade_hardening.py tests/test_gpu_host_apps.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for wl in cascade_late cascade_group; do
  timeout 600 python bench.py --workload $wl --also none --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  echo "== $wl"; cut -c1-260 $O/bench_$wl.json; tail -3 $O/bench_$wl.err
done
cd /tmp; export TMPDIR=/tmp
for wl in cascade_late cascade_group; do
  FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_$wl -- python $R/bench.py --workload $wl --also none --steps 3 --warmup 2 --frames-per-step 128 --no-cpu-baseline > $O/iso_$wl.json 2> $O/iso_$wl.err
  f=$(find $O/iso_$wl -name "*kernel_stats.csv" | head -1); cp $f $O/iso_${wl}_kernel_stats.csv; rm -rf $O/iso_$wl
done
