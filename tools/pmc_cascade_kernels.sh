# counters of the pyramid kernels of the headline workload (isolated: one call in flight)
R=$PWD; O=$R/gpurun_out/pmcp; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --workload cascade --also none --steps 2 --warmup 1 --frames-per-step 64 --no-cpu-baseline"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  FD_BENCH_SLOTS=1 FD_FRAMES_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- $CMD > /dev/null 2> $O/p$i.err
done
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for d in ("$O/p1","$O/p2"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            for key in ("k_resize_tiled","k_pyrdown_tiled","k_frames_to_gray","k_wvm_prefilter","k_wvm_deepB","k_svm_u8"):
                if key in k:
                    a=acc[key][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,cs in acc.items():
    v={c:s/n for c,(s,n) in cs.items()}
    gui=v.get("GRBM_GUI_ACTIVE",0)/8
    print(k, "launches", cs["SQ_WAVES"][1], "cycles/XCD %.0f"%gui, "waves %.0f"%v.get("SQ_WAVES",0))
    print("   VALU %.3g  SALU %.3g  LDS %.3g  VMEM_RD %.3g VMEM_WR %.3g" % tuple(v.get(c,0) for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR")))
    if gui: print("   valu issue frac %.3f  lds-active frac %.3f  bank-conflict/idx-active %.3f" % (v.get("SQ_ACTIVE_INST_VALU",0)*4/(1024*gui), v.get("SQ_LDS_IDX_ACTIVE",0)/(256*gui) if gui else 0, v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_LDS_IDX_ACTIVE",1),1)))
    wc=v.get("SQ_WAVE_CYCLES",1)
    print("   wave time: valu %.2f lds %.2f wait_any %.2f wait_inst %.2f" % (v.get("SQ_ACTIVE_INST_VALU",0)/wc, v.get("SQ_ACTIVE_INST_LDS",0)/wc, v.get("SQ_WAIT_ANY",0)/wc, v.get("SQ_WAIT_INST_ANY",0)/wc))
PY
