#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3c5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_filters.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for mode in new old; do
  E=""; [ $mode = old ] && E="FD_PYR_FUSED=0"
  env $E timeout 300 python bench.py --workload cascade --also none --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$mode.json 2> $O/bench_$mode.err
  echo "== $mode"; cut -c1-200 $O/bench_$mode.json
done
cd /tmp; export TMPDIR=/tmp
FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso -- python $R/bench.py --workload cascade --also none --steps 6 --warmup 2 --frames-per-step 128 --no-cpu-baseline > $O/iso.json 2> $O/iso.err
f=$(find $O/iso -name "*kernel_stats.csv" | head -1); cp $f $O/iso_kernel_stats.csv; rm -rf $O/iso
