#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cascade_hardening.py tests/test_gpu_parity.py -m gpu -q --maxfail=10 -x -k "not hog and not hist and not sdm and not fhog and not whi" > $O/pytest1.log 2>&1
echo "pytest1 rc=$?" >> $O/pytest1.log
tail -5 $O/pytest1.log
FD_HIP_LIB=$PWD/featuredetection_amd/alt/libfd_hip_wvbprof.so timeout 300 python tools/wvb_phases.py 64 2>&1 | grep phase
cd /tmp; export TMPDIR=/tmp
for mode in new; do
  E=""; [ $mode = old ] && E="FD_WVM_STAGEB=old"
  env $E FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_$mode -- python $R/bench.py --workload cascade --also none --steps 6 --warmup 2 --frames-per-step 128 --no-cpu-baseline > $O/iso_$mode.json 2> $O/iso_$mode.err
  f=$(find $O/iso_$mode -name "*kernel_stats.csv" | head -1); cp $f $O/iso_${mode}_kernel_stats.csv
  env $E timeout 300 python $R/bench.py --workload cascade --also none --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$mode.json 2> $O/bench_$mode.err
  echo "== $mode bench"; cut -c1-200 $O/bench_$mode.json
done
rm -rf $O/iso_new $O/iso_old
