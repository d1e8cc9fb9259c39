"""Per-kernel SQ counters (MFMA use, LDS bank conflicts) from one rocprofv3 --pmc pass:
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
              SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d <dir> -o s --output-format csv -- python3 bench.py ...
usage: python scripts/pmc_sq.py <counter_collection.csv> <out csv>"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("pmv::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
names = ["SQ_INSTS_VALU_MFMA_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU"]
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches," + ",".join(n + "_per_launch" for n in names) + ",mfma_busy_over_wave_cycles,lds_bank_conflict_over_lds_active\n")
    for k in sorted(acc):
        n = max(1, len(cnt[k]))
        v = [acc[k].get(x, 0.0) / n for x in names]
        mf = v[1] / v[2] if v[2] else 0.0
        lds = v[4] / v[5] if v[5] else 0.0
        row = [k, str(n)] + ["%.1f" % x for x in v] + ["%.4f" % mf, "%.4f" % lds]
        f.write(",".join(row) + "\n")
        print(k.ljust(26), "n=%5d" % n, "mfma_f64/launch %9.1f" % v[0], "mfma_busy/wave_cyc %.4f" % mf, "lds_conflict/lds_active %.3f" % lds, "valu/launch %.0f" % v[6])
