#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3c6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cascade_hardening.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python bench.py --workload cascade --also none --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cascade.json 2> $O/bench_cascade.err
cut -c1-200 $O/bench_cascade.json
timeout 600 python bench.py --workload ffp15 --also none --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_ffp15.json 2> $O/bench_ffp15.err
cut -c1-220 $O/bench_ffp15.json
cd /tmp; export TMPDIR=/tmp
FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso -- python $R/bench.py --workload cascade --also none --steps 6 --warmup 2 --frames-per-step 128 --no-cpu-baseline > $O/iso.json 2> $O/iso.err
f=$(find $O/iso -name "*kernel_stats.csv" | head -1); cp $f $O/iso_kernel_stats.csv; rm -rf $O/iso
