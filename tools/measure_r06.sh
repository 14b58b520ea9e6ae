#!/bin/bash
# Round-6 measurement batch (run on the GPU box from the repo root): for ALL SIX workloads of the bench line, at the git head of the
# snapshot (.git_head), (1) rocprofv3 kernel stats with the workload's launches ISOLATED (one call / frame / batch in flight, host
# stages inline, one stream and one host thread for the 15-detector batch) so that a kernel's average duration is its own and
# bench.py's roofline.kernel_ms can be recomputed from the CSV; (2) the --pmc passes (each counter group in its own run, kernel trace
# only) behind roofline.traffic and roofline_issue; (3) the driver's command, whose line reads the fresh profiles/r06_pmc.json, and the
# single-frame latency.  Outputs under gpurun_out/measure6/; tools/collect_profiles_r06.sh copies the summaries into profiles/.
# usage: tools/measure_r06.sh [stats|pmc|bench|all|tail]   (default all; tail: the host-stages / device-tail x hardware-queues table);  WLS="..." restricts the workloads
R=$PWD
O=$R/gpurun_out/measure6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
WHAT=${1:-all}
HEAD=$(cat $R/.git_head 2>/dev/null || echo unknown)
WLS=${WLS:-cascade cascade_late cascade_group hog_svm ffp15 sdm}
iso_env() {   # one unit of work in flight: every kernel runs alone
  case $1 in
    cascade*) echo "env FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0";;
    hog_svm)  echo "env FD_BENCH_HOG_INFLIGHT=1";;
    ffp15)    echo "env FD_BENCH_FFP_SLOTS=1 FD_BATCH_THREADS=1 FD_BATCH_STREAMS=1";;
    sdm)      echo "env FD_BENCH_SDM_INFLIGHT=1";;
  esac
}
NOISE="naive_conv\|miopen\|MIOpen\|at::native\|elementwise_kernel\|vectorized_elementwise\|reduce_kernel\|index_elementwise\|CatArrayBatchedCopy\|philox\|distribution"
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
  for wl in $WLS; do
    S=10; [ $wl = ffp15 ] && S=3
    FP=""; case $wl in cascade*) FP="--frames-per-step 128";; esac
    timeout 300 $(iso_env $wl) rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- $B --workload $wl --also none --steps $S --warmup 2 $FP --no-cpu-baseline --no-probe --full-out $O/stats_$wl.full.json > $O/stats_$wl.json 2> $O/stats_$wl.err
    f=$(find $O/stats_$wl -name "*kernel_stats.csv" | head -1)
    # (the frame generators of ffp15 / sdm run torch kernels once at set-up: not part of the workload)
    [ -n "$f" ] && grep -v "$NOISE" $f > $O/r06_${wl}_kernel_stats.csv
    rm -rf $O/stats_$wl $O/stats_$wl.full.json
  done
fi
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  [ -z "$KEEP_PMC" ] && rm -f $O/r06_pmc.json   # KEEP_PMC=1 WLS=sdm: add one workload to an existing file
  for wl in $WLS; do
    S=3; FP="--frames-per-step 64"
    [ $wl = sdm ] && FP="--frames-per-step 4"
    [ $wl = hog_svm ] && FP="--frames-per-step 8"
    [ $wl = ffp15 ] && { S=2; FP="--frames-per-step 4"; }
    CMD="$B --workload $wl --also none --steps $S --warmup 1 $FP --no-cpu-baseline --no-probe --full-out /tmp/pmc_full.json"
    i=0
    for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
               "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
               "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      PX=""; [ $wl = sdm ] && PX="env FD_BENCH_SDM_NBATCH=1 FD_BENCH_SDM_HOSTGEN=1"   # (rocprofv3 --pmc segfaults inside the torch crop generator: host-made crops, one batch)
      timeout 400 $(iso_env $wl) $PX rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_${wl}_$i -- $CMD > /dev/null 2> $O/pmc_${wl}_$i.err
    done
    case $wl in
      sdm)     K=k_sdm_descriptors,k_sdm;;
      hog_svm) K=k_hog_svm_fused,k_hsf_finish,k_gradbin,k_pyrdown,k_resize,k_bgr2gray;;
      *)       K=k_wv,k_resize,k_pyrdown,k_frames_to_gray,k_bgr2gray,k_svm_u8,k_fs_oe;;
    esac
    python $R/tools/pmc_summary.py $wl $O/r06_pmc.json $K "rocprofv3 --kernel-trace --pmc <group> (4 separate passes) -- $(iso_env $wl | sed 's/^env //') python bench.py --workload $wl --also none --steps $S --warmup 1 $FP --no-cpu-baseline --no-probe; git head $HEAD" $O/pmc_${wl}_1 $O/pmc_${wl}_2 $O/pmc_${wl}_3 $O/pmc_${wl}_4 > $O/pmc_$wl.summary 2>&1
    rm -rf $O/pmc_${wl}_1 $O/pmc_${wl}_2 $O/pmc_${wl}_3 $O/pmc_${wl}_4
  done
  [ -f $O/r06_pmc.json ] && cp $O/r06_pmc.json $R/profiles/r06_pmc.json   # the bench line below cites it
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  ( cd $R && timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $O/bench_also.json > $O/bench_line.json 2> $O/bench_default.err )
  timeout 300 python $R/tools/latency_probe.py 1000 > $O/latency_single_frame.txt 2>&1
fi
if [ "$WHAT" = tail ]; then
  # DESIGN.md section 6's table: the 15-detector batch with its stages 2-3 on the host / on the device (k_fs_oe_big) against the HIP
  # runtime's hardware queues, and the headline beside it (one process has one GPU_MAX_HW_QUEUES)
  T=$O/ffp15_tail_matrix.txt; : > $T
  run() { wl=$1; shift; env "$@" $B --workload $wl --also none --steps 20 --warmup 3 --no-cpu-baseline --no-probe --full-out /tmp/tail_full.json | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('%-8s %-95s %8.1f Mpatches/s %7.2f ms/step' % ('$wl', '$*', r['value'], r['ms_per_step']))" >> $T; }
  for Q in 4 6 8; do
    run cascade GPU_MAX_HW_QUEUES=$Q
    run ffp15 GPU_MAX_HW_QUEUES=$Q FD_FS_TAIL=0 FD_BATCH_THREADS=8
    run ffp15 GPU_MAX_HW_QUEUES=$Q FD_FS_TAIL=1 FD_BATCH_THREADS=1
    run ffp15 GPU_MAX_HW_QUEUES=$Q FD_FS_TAIL=1 FD_BATCH_THREADS=2
  done
  echo "git head $HEAD" >> $T
  cat $T
fi
ls $O | head -60
wc -c $O/bench_line.json 2>/dev/null
python3 - <<PY
import json
try:
    r = json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
    print("SUMMARY", json.dumps(r.get("summary")))
except Exception as e:
    print("no bench line:", e)
PY
tail -2 $O/latency_single_frame.txt 2>/dev/null
