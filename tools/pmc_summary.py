"""Per-launch counter figures of the kernels of one bench.py workload from rocprofv3 --pmc passes -> profiles/r02_pmc.json.

usage: pmc_summary.py <workload> <out.json> <kernel-substring> <source-note> <pass_dir> [<pass_dir> ...]
Every pass directory holds one `rocprofv3 --kernel-trace --pmc <counters>` run of the same bench command (counters are
collected in separate passes, never together with other trace domains).  Values are averaged over the launches of every kernel
whose name contains the substring; FETCH_SIZE is doubled (gfx950: 128-byte requests are tallied at 64 bytes,
MI355X_MICROARCH.md section HBM) and both size counters are converted from KiB to bytes.  SQ_* cycle counters are in
quad-cycles (same guide), SQ_INSTS_* in wave instructions."""
import csv
import glob
import json
import re
import sys

workload, out, substr, note = sys.argv[1:5]
dirs = sys.argv[5:]
subs = substr.split(",")   # kernels matching ANY are listed; the k_all(...) aggregate covers those matching the FIRST
substr = subs[0]


def wanted(k):
    return any(x in k for x in subs)


def short(name):
    m = re.match(r"(?:void\s+)?(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name


acc = {}   # kernel -> counter -> [sum, n]
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not wanted(k):
                continue
            a = acc.setdefault(short(k), {}).setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    # kernel durations of the same passes (kernel trace)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not wanted(k):
                continue
            a = acc.setdefault(short(k), {}).setdefault("_dur_ns", [0.0, 0])
            a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            a[1] += 1
if not acc:
    raise SystemExit("no samples for kernels matching %r in %s" % (substr, dirs))
kernels = {}
for k, cs in acc.items():
    rec = {}
    for c, (s, n) in cs.items():
        rec[c] = s / n
        rec.setdefault("launches", n)
    o = dict(launches=rec.get("launches", 0))
    if "_dur_ns" in rec:
        o["kernel_ms_profiled"] = rec["_dur_ns"] / 1e6
    if "FETCH_SIZE" in rec or "WRITE_SIZE" in rec:
        o["fetch_bytes"] = 2 * rec.get("FETCH_SIZE", 0.0) * 1024
        o["write_bytes"] = rec.get("WRITE_SIZE", 0.0) * 1024
        o["hbm_bytes"] = o["fetch_bytes"] + o["write_bytes"]
    for c, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_SALU", "salu_insts"), ("SQ_INSTS_LDS", "lds_insts"), ("SQ_INSTS_VMEM_RD", "vmem_rd_insts"),
                   ("SQ_INSTS_MFMA", "mfma_insts"), ("SQ_INSTS_VALU_MFMA_MOPS_I8", "mfma_mops_i8"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu_quadcycles"),
                   ("SQ_ACTIVE_INST_LDS", "active_inst_lds_quadcycles"), ("SQ_WAVE_CYCLES", "wave_quadcycles"), ("SQ_BUSY_CYCLES", "sq_busy_cycles"),
                   ("SQ_WAVES", "waves"), ("GRBM_GUI_ACTIVE", "gui_active_cycles"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles"),
                   ("SQ_LDS_IDX_ACTIVE", "lds_idx_active_cycles"), ("SQ_WAIT_INST_ANY", "wait_inst_any_quadcycles"), ("SQ_WAIT_ANY", "wait_any_quadcycles"),
                   ("SQ_ACTIVE_INST_ANY", "active_inst_any_quadcycles"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles")):
        if c in rec:
            o[key] = rec[c]
    if rec.get("GRBM_GUI_ACTIVE") and "SQ_INSTS_VALU" in rec:
        o["valu_issue_frac"] = 4.0 * rec["SQ_INSTS_VALU"] / (128.0 * rec["GRBM_GUI_ACTIVE"])
    if rec.get("SQ_LDS_IDX_ACTIVE"):
        o["lds_bank_conflict_frac"] = rec.get("SQ_LDS_BANK_CONFLICT", 0.0) / rec["SQ_LDS_IDX_ACTIVE"]
    kernels[k] = o
# workload aggregate over the matched kernels: per-"frame" figures = sum over kernels of (per-launch value x launches) / launches of the
# most frequent kernel would be ambiguous, so the aggregate is the plain sum of per-launch averages (one launch of each kernel)
agg = {}
for kn, o in kernels.items():
    if substr not in kn:
        continue
    for key, v in o.items():
        if key not in ("launches", "valu_issue_frac", "lds_bank_conflict_frac"):
            agg[key] = agg.get(key, 0.0) + v
agg["kernel_ms"] = agg.pop("kernel_ms_profiled", None)
# time-weighted totals over ALL launches of the matched kernels (counter passes serialise the dispatches): the fraction of the
# VALU issue slots that were used while these kernels ran = 4 cycles x SQ_INSTS_VALU / (128 SIMDs per XCD x GRBM_GUI_ACTIVE summed
# over the 8 XCDs), and the same for LDS instructions per cycle
tot, cnt = {}, {}
for kn, cs in acc.items():
    if substr not in kn:
        continue
    for c, (sv, n) in cs.items():
        tot[c] = tot.get(c, 0.0) + sv
        cnt[c] = cnt.get(c, 0) + n
if tot.get("GRBM_GUI_ACTIVE"):
    if "SQ_INSTS_VALU" in tot:
        agg["valu_issue_frac"] = 4.0 * tot["SQ_INSTS_VALU"] / (128.0 * tot["GRBM_GUI_ACTIVE"])
    if "SQ_INSTS_LDS" in tot:
        agg["lds_insts_per_simd_cycle"] = tot["SQ_INSTS_LDS"] / (128.0 * tot["GRBM_GUI_ACTIVE"])
    if "_dur_ns" in tot:   # durations come from every pass, the counter from one: compare per-launch averages
        agg["clock_ghz"] = (tot["GRBM_GUI_ACTIVE"] / cnt["GRBM_GUI_ACTIVE"]) / 8.0 / (tot["_dur_ns"] / cnt["_dur_ns"])
if tot.get("SQ_WAVE_CYCLES"):
    for c, key in (("SQ_ACTIVE_INST_VALU", "wave_time_valu"), ("SQ_ACTIVE_INST_LDS", "wave_time_lds"), ("SQ_WAIT_ANY", "wave_time_waitcnt"),
                   ("SQ_WAIT_INST_ANY", "wave_time_issue_stall"), ("SQ_ACTIVE_INST_ANY", "wave_time_active")):
        if c in tot:
            agg[key] = tot[c] / tot["SQ_WAVE_CYCLES"]
try:
    allrec = json.load(open(out))
except Exception:
    allrec = {}
kernels["k_all(" + substr + ")"] = agg
allrec[workload] = dict(source=note, kernels=kernels)
json.dump(allrec, open(out, "w"), indent=1)
print(json.dumps(allrec[workload], indent=1))
