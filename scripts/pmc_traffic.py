"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see
/opt/skills/guides/MI355X_MICROARCH.md 'HBM' and 'rocprofv3 PMC slots'):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py ... --frames 300
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py ... --frames 300
FETCH_SIZE / WRITE_SIZE are in KiB (TCC_EA0 request counters x 64 B / 1024); on gfx950 FETCH_SIZE counts 128-B requests as
64 B, so it is doubled. Writes profiles/pmc_traffic.json + a per-kernel CSV summary.
usage: python scripts/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out json> <out csv> <note> [bench config, default 1]"""
import csv, json, sys, collections
fetch_csv, write_csv, out_json, out_csv, note = sys.argv[1:6]
config = int(sys.argv[6]) if len(sys.argv) > 6 else 1


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("pmv::", "").replace("void ", "").split("<")[0]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


f = per_kernel(fetch_csv, "FETCH_SIZE")
w = per_kernel(write_csv, "WRITE_SIZE")
rows = []
kern = {}
for k in sorted(set(f) | set(w)):
    nf, sf = f.get(k, [0, 0.0]); nw, sw = w.get(k, [0, 0.0])
    fetch_b = 2.0 * 1024.0 * sf / max(nf, 1)      # gfx950 correction x2
    write_b = 1024.0 * sw / max(nw, 1)
    rows.append((k, nf, round(fetch_b, 1), round(write_b, 1), round(fetch_b + write_b, 1)))
    kern[k] = dict(launches=nf, fetch_bytes_per_launch=round(fetch_b, 1), write_bytes_per_launch=round(write_b, 1),
                   hbm_bytes_per_launch=round(fetch_b + write_b, 1))
json.dump(dict(source=note, config=config, unit="bytes per launch; FETCH_SIZE (KiB) x 2 (gfx950) + WRITE_SIZE (KiB)", kernels=kern), open(out_json, "w"), indent=1)
with open(out_csv, "w") as fo:
    fo.write("kernel,launches,fetch_bytes_per_launch(x2 corrected),write_bytes_per_launch,hbm_bytes_per_launch\n")
    for r in rows:
        fo.write(",".join(str(x) for x in r) + "\n")
for r in rows:
    print(r)
