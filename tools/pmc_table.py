"""Per-kernel HBM-side traffic of one matching step from a tools/pmc.sh run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes):
bytes per launch (FETCH_SIZE doubled: gfx950 tallies 128-B read requests at 64 B; both counters are KiB and include Infinity-Cache
hits), the launch's duration in that (serialised) run and the resulting GB/s against the 8 TB/s HBM3E peak.
usage: pmc_table.py <pmc dir> <out.md>"""
import csv, glob, os, re, sys, collections
d, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ns": 0.0, "nf": 0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = max(glob.glob(f"{d}/{c}/*/*counter_collection.csv"), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        a = agg[name]
        a[c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            a["nf"] += 1
            a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        else:
            a["n"] += 1
rows = []
for name, a in agg.items():
    if not a["nf"] or not a["n"] or not name.startswith("k_"):
        continue
    rd, wr, us = 2 * a["FETCH_SIZE"] / a["nf"] * 1024, a["WRITE_SIZE"] / a["n"] * 1024, a["ns"] / a["nf"] / 1e3
    rows.append((us * a["nf"], name, a["nf"], rd, wr, us))
rows.sort(reverse=True)
with open(out, "w") as f:
    f.write("# HBM-side traffic per kernel (rocprofv3 PMC, cfg2 step, averages over all launches of the run)\n\n")
    f.write("FETCH_SIZE x2 (gfx950 correction) and WRITE_SIZE, KiB -> bytes; both include Infinity-Cache hits.  Durations are from the PMC run\n"
            "(kernels serialised by the profiler).  Peak: 8000 GB/s.\n\n")
    f.write("| kernel | launches | read MB | written MB | us | GB/s | of peak |\n|---|---|---|---|---|---|---|\n")
    for _, name, n, rd, wr, us in rows:
        gbs = (rd + wr) / us / 1e3
        f.write(f"| `{name}` | {n} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | {us:.1f} | {gbs:.0f} | {gbs / 8000:.2f} |\n")
print(open(out).read())
