"""Per-launch average of a PMC counter per kernel from a rocprofv3 --pmc run.  usage: pmc_parse.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt in name:
            a = agg[(r["Counter_Name"], name[:70], r.get("Grid_Size", ""))]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for (c, n, g), (k, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{c:12s} launches {k:5d}  avg/launch {v / k / 1e6:10.3f} M  grid {g:>9s}  {n}")
