"""One step of a rocprofv3 kernel trace as a timeline.  usage: steptrace.py <dir> <step_index> [marker_kernel=k_cloud_offsets]
A step starts at each dispatch of the marker kernel (the first kernel of the captured step)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
idx = int(sys.argv[2])
marker = sys.argv[3] if len(sys.argv) > 3 else "k_cloud_offsets"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[3]]
print(len(starts), "steps; step durations (us):", [round((rows[starts[i + 1]][0] - rows[starts[i]][0]) / 1e3) for i in range(len(starts) - 1)][:60])
a, b = starts[idx], starts[idx + 1]
t0 = rows[a][0]
qs = sorted(set(r[2] for r in rows[a:b]))
busy = {q: 0 for q in qs}
for s, e, q, n in rows[a:b]:
    busy[q] += e - s
    name = n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{qs.index(q)}  {name}")
print("busy per queue (us):", {qs.index(q): round(v / 1e3) for q, v in busy.items()}, "step", (rows[b][0] - t0) / 1e3)
