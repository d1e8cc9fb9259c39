"""GPU busy fraction of a run from a rocprofv3 --kernel-trace CSV: union of all kernel intervals over the span from the first
batched launch to the last kernel. usage: python scripts/gpu_busy.py <kernel_trace.csv> [name of the first kernel of the region]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
first = sys.argv[2] if len(sys.argv) > 2 else "k_lk_batch"
rows.sort()
t0 = min(s for s, e, n in rows if first in n)
iv = [(s, e) for s, e, n in rows if s >= t0]
t1 = max(e for s, e in iv)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
# average number of kernels in flight
area = sum(e - s for s, e in iv)
print("span %.3f s, busy (>= 1 kernel running) %.1f %%, kernels in flight on average %.2f, launches %d" % ((t1 - t0) / 1e9, 100.0 * busy / (t1 - t0), area / (t1 - t0), len(iv)))
