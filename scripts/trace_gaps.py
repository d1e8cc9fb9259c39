"""Back-end stream timeline of one bench step from a rocprofv3 --kernel-trace CSV: per PnP call the gaps copy -> hypotheses ->
refit and the host turnaround until the next launch on that stream. usage: python scripts/trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["n"] = r["Kernel_Name"].split("(")[0].replace("pmv::", "")
byq = collections.defaultdict(list)
for r in rows: byq[r["Queue_Id"]].append(r)
for q, L in byq.items():
    L.sort(key=lambda r: r["s"])
    names = collections.Counter(r["n"] for r in L)
    if "k_pnp_hyp" not in names: continue
    print("queue", q, dict(names.most_common(8)))
    gaps = collections.defaultdict(list)
    for a, b in zip(L, L[1:]):
        gaps[(a["n"], b["n"])].append((b["s"] - a["e"]) / 1e3)
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print("  %-28s -> %-28s n=%5d  median %7.1f us  mean %7.1f  total %8.1f ms" % (k[0], k[1], len(v), st.median(v), st.mean(v), sum(v) / 1e3))
    busy = sum(r["e"] - r["s"] for r in L) / 1e6
    span = (L[-1]["e"] - L[0]["s"]) / 1e6
    print("  busy %.1f ms of %.1f ms span (%.0f %%)" % (busy, span, 100 * busy / span))
