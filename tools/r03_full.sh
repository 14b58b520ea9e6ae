#!/bin/bash
# full GPU suite + the measurements of tools/r03_call2.sh
R=$PWD; O=$R/gpurun_out/r3full; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_all.log 2>&1
echo "pytest_all rc=$?" >> $O/pytest_all.log
tail -15 $O/pytest_all.log
FD_HIP_LIB=$PWD/featuredetection_amd/alt/libfd_hip_wvbprof.so timeout 300 python tools/wvb_phases.py 64 2>&1 | grep phase
cd /tmp; export TMPDIR=/tmp
FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_new -- python $R/bench.py --workload cascade --also none --steps 6 --warmup 2 --frames-per-step 128 --no-cpu-baseline > $O/iso_new.json 2> $O/iso_new.err
f=$(find $O/iso_new -name "*kernel_stats.csv" | head -1); cp $f $O/iso_new_kernel_stats.csv; rm -rf $O/iso_new
timeout 600 python $R/bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
cut -c1-300 $O/bench_default.json
