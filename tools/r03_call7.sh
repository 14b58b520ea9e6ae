#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3c7; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_filters.py tests/test_gpu_robustness.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for mode in new old; do
  E=""; [ $mode = old ] && E="FD_PYR_CHAIN=0"
  env $E timeout 300 python bench.py --workload cascade --also none --steps 10 --warmup 3 --no-cpu-baseline --no-probe > $O/bench_$mode.json 2> $O/bench_$mode.err
  echo "== $mode"; cut -c1-200 $O/bench_$mode.json
done
python tools/latency_probe.py 500 2>&1 | tail -1
bash tools/measure_r03.sh stats > $O/measure_stats.log 2>&1
bash tools/measure_r03.sh pmc > $O/measure_pmc.log 2>&1
tail -3 $O/measure_pmc.log | cut -c1-300
