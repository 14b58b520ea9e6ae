#!/bin/bash
# Round-5 measurement batch (run on the GPU box from the repo root): rocprofv3 kernel stats per workload, the --pmc passes (each counter
# group in its own run, kernel trace only) behind the roofline figures of the cascade workloads, then the driver's bench line (which
# reads the fresh profiles/r05_pmc.json) and the single-frame latency.  Outputs under gpurun_out/measure5/;
# tools/collect_profiles_r05.sh copies the summaries into profiles/.
# usage: tools/measure_r05.sh [stats|pmc|bench|all]   (default all)
R=$PWD
O=$R/gpurun_out/measure5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
WHAT=${1:-all}
HEAD=$(cat $R/.git_head 2>/dev/null || echo unknown)
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  for wl in cascade cascade_late cascade_group hog_svm ffp15 sdm; do
    S=10; [ $wl = ffp15 ] && S=3
    FP=""; case $wl in cascade*) FP="--frames-per-step 128";; esac
    # cascade: ONE call in flight and the host stages inline, so that a kernel's duration is its own (with six calls in flight the
    # kernels of different calls share the CUs); bench.py's live figure (kernel_probe) is taken the same way
    ENVX=""; case $wl in cascade*) ENVX="env FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0";; esac
    [ $wl = sdm ] && ENVX="env FD_BENCH_SDM_INFLIGHT=1"
    [ $wl = ffp15 ] && ENVX="env FD_BENCH_FFP_SLOTS=1"
    timeout 300 $ENVX rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- $B --workload $wl --also none --steps $S --warmup 2 $FP --no-cpu-baseline --no-probe > $O/stats_$wl.json 2> $O/stats_$wl.err
    f=$(find $O/stats_$wl -name "*kernel_stats.csv" | head -1)
    # (the frame generator of ffp15 / config5 runs torch convolutions once at set-up: not part of the workload)
    [ -n "$f" ] && grep -v "naive_conv\|miopen\|MIOpen\|at::native\|elementwise_kernel\|vectorized_elementwise\|reduce_kernel\|index_elementwise\|CatArrayBatchedCopy\|philox\|distribution" $f > $O/r05_${wl}_kernel_stats.csv
    rm -rf $O/stats_$wl
  done
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  [ -f $O/r05_pmc.json ] || cp $R/profiles/r05_pmc.json $O/r05_pmc.json 2>/dev/null   # (a partial re-run keeps the other workloads' entries)
  for wl in ${PMC_WLS:-cascade cascade_late cascade_group sdm}; do
    S=3; FP="--frames-per-step 64"
    [ $wl = sdm ] && FP="--frames-per-step 4"
    CMD="$B --workload $wl --also none --steps $S --warmup 1 $FP --no-cpu-baseline --no-probe"
    i=0
    for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
               "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
               "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_${wl}_$i -- $CMD > /dev/null 2> $O/pmc_${wl}_$i.err
    done
    K=k_wv,k_resize,k_pyrdown,k_frames_to_gray,k_svm_u8,k_fs_oe
    [ $wl = sdm ] && K=k_sdm_descriptors,k_sdm
    python $R/tools/pmc_summary.py $wl $O/r05_pmc.json $K "rocprofv3 --kernel-trace --pmc <group> (4 separate passes) -- python bench.py --workload $wl --also none --steps $S --warmup 1 $FP --no-cpu-baseline --no-probe; git head $HEAD" $O/pmc_${wl}_1 $O/pmc_${wl}_2 $O/pmc_${wl}_3 $O/pmc_${wl}_4 > $O/pmc_$wl.summary 2>&1
    rm -rf $O/pmc_${wl}_1 $O/pmc_${wl}_2 $O/pmc_${wl}_3 $O/pmc_${wl}_4
  done
  [ -f $O/r05_pmc.json ] && cp $O/r05_pmc.json $R/profiles/r05_pmc.json   # the bench line below cites it
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  ( cd $R && timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err )
  timeout 300 python $R/tools/latency_probe.py 1000 > $O/latency_single_frame.txt 2>&1
fi
ls $O | head -60
python3 - <<PY
import json
try:
    r = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("SUMMARY", json.dumps(r.get("summary")))
except Exception as e:
    print("no bench line:", e)
PY
tail -2 $O/latency_single_frame.txt 2>/dev/null
