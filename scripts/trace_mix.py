"""What the kernels of the batched leg cost IN THE MIX, from a rocprofv3 --kernel-trace CSV: per kernel name the launches, the average and total
duration, and - per queue (= stream of one combiner) - how much of the span the queue had a kernel running and how long a kernel waited between
the end of its predecessor on the same queue and its own start (dispatch gap).  usage: python scripts/trace_mix.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "0")))
rows.sort()
t0 = min(s for s, e, n, q in rows if "k_lk_batch" in n)
rows = [x for x in rows if x[0] >= t0]
t1 = max(e for s, e, n, q in rows)
span = (t1 - t0) / 1e3
print("span %.1f ms, %d launches" % (span / 1e3, len(rows)))
byname = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in rows:
    byname[n][0] += 1; byname[n][1] += (e - s) / 1e3
print("%-34s %8s %10s %10s" % ("kernel", "launches", "avg us", "total ms"))
for n, (c, t) in sorted(byname.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %8d %10.1f %10.1f" % (n[-34:], c, t / c, t / 1e3))
byq = collections.defaultdict(list)
for s, e, n, q in rows:
    byq[q].append((s, e, n))
print("%-8s %8s %9s %9s %12s  kernels" % ("queue", "launches", "busy %", "gap<50us", "avg gap us"))
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, n in lst) / 1e3
    gaps = [(lst[i][0] - lst[i - 1][1]) / 1e3 for i in range(1, len(lst))]
    small = [g for g in gaps if 0 <= g < 50]
    names = collections.Counter(n[-22:] for s, e, n in lst).most_common(3)
    print("%-8s %8d %9.1f %9d %12.2f  %s" % (q, len(lst), 100 * busy / span, len(small), sum(small) / max(1, len(small)), names))
