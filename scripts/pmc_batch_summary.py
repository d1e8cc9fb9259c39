"""Per-kernel, per-wavefront averages of one rocprofv3 --pmc pass (run on the GPU box; the raw CSVs of a batched run exceed the
copy-back limit). usage: python scripts/pmc_batch_summary.py <dir with p_counter_collection.csv / p_kernel_trace.csv> <out.txt>"""
import csv, collections, sys
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f"{d}/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("pmv::", "").replace("void ", "").split("<")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(float); n2 = collections.Counter()
for r in csv.DictReader(open(f"{d}/p_kernel_trace.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("pmv::", "").replace("void ", "").split("<")[0]
    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; n2[k] += 1
with open(out, "w") as f:
    for k in sorted(acc, key=lambda k: -dur[k]):
        n = len(cnt[k]); w = acc[k].get("SQ_WAVES", 0.0)
        line = "%-28s launches %6d avg_us %8.1f " % (k, n, dur[k] / max(1, n2[k]))
        if w:
            line += "waves/launch %8.0f " % (w / n) + " ".join("%s/wave=%.5g" % (c, v / w) for c, v in sorted(acc[k].items()) if c != "SQ_WAVES")
        else:
            line += " ".join("%s/launch=%.5g" % (c, v / n) for c, v in sorted(acc[k].items()))
        f.write(line + "\n"); print(line)
