"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel stats and the launch chain of one BA solve.
usage: python scripts/prof_summary.py <dir> <prefix> [ba_call_index]"""
import csv, sys
d, pre = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 300
rows = list(csv.DictReader(open(f"{d}/{pre}_kernel_stats.csv")))
for r in rows:
    print(r["Name"][:48].ljust(48), r["Calls"].rjust(7), r["TotalDurationNs"].rjust(12), r["AverageNs"][:9].rjust(10), r["Percentage"][:5])
tr = list(csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(tr) if "k_bam_eval" in r["Kernel_Name"]]
# a solve starts at every 5th eval launch (max_iterations = 5)
if len(starts) > which * 5:
    i0 = starts[which * 5]
    q = tr[i0].get("Queue_Id")
    seq = []
    for r in tr[i0:]:
        if r.get("Queue_Id") != q:
            continue
        seq.append(r)
        if "k_bam_finish" in r["Kernel_Name"]:
            break
    t0 = int(seq[0]["Start_Timestamp"]); prev = t0; busy = 0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); busy += e - s
        print(r["Kernel_Name"].split("(")[0][-22:].ljust(24), "gap %6d  dur %6d" % (s - prev, e - s))
        prev = e
    print("solve total ns", prev - t0, "busy", busy)
