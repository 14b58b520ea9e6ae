#!/bin/bash
# Round-1 measurement batch (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats and the
# FETCH_SIZE / WRITE_SIZE passes the roofline `traffic` figures come from.  Outputs under gpurun_out/measure/.
R=$PWD
O=$R/gpurun_out/measure
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
for wl in hog_svm wvm sdm; do
  timeout 240 $B --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- $B --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
done
timeout 200 $B --workload wvm --size 1920x1080 --no-cpu-baseline > $O/bench_wvm_1080p.json 2> $O/bench_wvm_1080p.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wvm_1080p -- $B --workload wvm --size 1920x1080 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
timeout 200 $B --workload ffp15 --steps 5 --warmup 2 > $O/bench_ffp15.json 2> $O/bench_ffp15.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ffp15 -- $B --workload ffp15 --steps 3 --warmup 1 > /dev/null 2>&1
timeout 200 $B --workload rvm --size 1920x1080 --no-cpu-baseline > $O/bench_rvm_1080p.json 2> $O/bench_rvm_1080p.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rvm_1080p -- $B --workload rvm --size 1920x1080 --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
timeout 200 $B --workload aggregated --steps 20 --warmup 3 > $O/bench_aggregated.json 2> $O/bench_aggregated.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_aggregated -- $B --workload aggregated --steps 10 --warmup 3 > /dev/null 2>&1
for wl in hog_svm wvm; do
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$wl -- $B --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$wl -- $B --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_fetch_hog_svm $O/pmc_write_hog_svm k_svm_rbf_mfma hog_svm $O/r01_pmc_traffic.json
python $R/tools/pmc_traffic.py $O/pmc_fetch_wvm $O/pmc_write_wvm k_wvm_cascade wvm $O/r01_pmc_traffic.json
cat $O/bench_*.json
