import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 23
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms/step", tot / 1e6 / steps)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{float(r['TotalDurationNs'])/1e3/steps:8.1f} us/step {float(r['Percentage']):5.1f}% calls/step {float(r['Calls'])/steps:5.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:80]}")
