"""Per-call durations of k_arp_inorm_bwd inside a step, from a rocprofv3 kernel trace: python tools/arp_calls.py <trace dir> [step]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_assemble_batch" in r["Kernel_Name"]]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 10
acc = {}
for s in range(step, min(step + 6, len(starts) - 1)):
    seg = rows[starts[s]:starts[s + 1]]
    t0 = int(seg[0]["Start_Timestamp"])
    j = 0
    for r in seg:
        if "k_arp_inorm_bwd" in r["Kernel_Name"]:
            acc.setdefault(j, []).append(((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // 256))
            j += 1
    acc.setdefault("wall", []).append((int(rows[starts[s + 1]]["Start_Timestamp"]) - t0) / 1e3)
for k, v in acc.items():
    if k == "wall":
        print("step wall us", round(sum(v) / len(v), 1))
    else:
        print("call", k, "clouds", v[0][2], "start", round(sum(x[0] for x in v) / len(v), 1), "dur", round(sum(x[1] for x in v) / len(v), 1))
