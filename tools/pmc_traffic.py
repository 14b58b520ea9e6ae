"""Per-launch HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <fetch_dir> <write_dir> <kernel-substring> <workload> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide
coalesced reads (MI355X_MICROARCH.md, HBM section), so it is doubled.  WRITE_SIZE is taken as reported."""
import csv, glob, json, sys


def avg(d, counter, kernel):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s samples for %s in %s" % (counter, kernel, d))
    return sum(vals) / len(vals), len(vals)


fetch_dir, write_dir, kernel, workload, out = sys.argv[1:6]
f, nf = avg(fetch_dir, "FETCH_SIZE", kernel)
w, nw = avg(write_dir, "WRITE_SIZE", kernel)
rec = dict(kernel=kernel, launches=nf, fetch_size_kib_raw=f, write_size_kib_raw=w, fetch_bytes=2 * f * 1024, write_bytes=w * 1024,
           hbm_bytes_per_launch=2 * f * 1024 + w * 1024,
           note="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --workload %s --steps 5 --warmup 2 "
                "--no-cpu-baseline`; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)" % workload)
try:
    allrec = json.load(open(out))
except Exception:
    allrec = {}
allrec[workload] = rec
json.dump(allrec, open(out, "w"), indent=1)
print(json.dumps(rec))
