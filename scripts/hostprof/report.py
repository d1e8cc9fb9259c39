"""usage: python scripts/hostprof/report.py <hostprof output> [top N]: CPU samples per library and per function (nm on in-tree libraries)"""
import bisect, collections, os, subprocess, sys
path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
maps, pcs, chains = [], [], []
for line in open(path):
    if line.startswith("map "):
        p = line.split()
        lo, hi = (int(x, 16) for x in p[1].split("-"))
        maps.append((lo, hi, int(p[3], 16), p[6] if len(p) > 6 else "[anon]"))
    elif not line.startswith("samples"):
        q = [int(x, 16) for x in line.split()]
        pcs.append(q[0]); chains.append(q[1:])
maps.sort()
syms, segs = {}, {}
def file_to_vaddr(lib, a):   # nm prints virtual addresses, /proc/self/maps gives file offsets: translate through the PT_LOAD headers
    if lib not in segs:
        L = []
        try:
            for l in subprocess.run(["readelf", "-lW", lib], capture_output=True, text=True).stdout.splitlines():
                q = l.split()
                if q and q[0] == "LOAD": L.append((int(q[1], 16), int(q[2], 16), int(q[4], 16)))   # offset, vaddr, filesz
        except Exception: pass
        segs[lib] = L
    for o, v, sz in segs[lib]:
        if o <= a < o + sz: return a - o + v
    return a
def table(lib):
    if lib not in syms:
        t = []
        if os.path.exists(lib):
            try:
                out = subprocess.run(["nm", "-C", "--defined-only", "-n", lib], capture_output=True, text=True).stdout
                out += subprocess.run(["nm", "-C", "-D", "--defined-only", "-n", lib], capture_output=True, text=True).stdout
                for l in out.splitlines():
                    q = l.split(None, 2)
                    if len(q) == 3 and q[1] in "tTwW": t.append((int(q[0], 16), q[2]))
            except Exception: pass
        t.sort(); syms[lib] = t
    return syms[lib]
by_lib, by_fn, by_chain = collections.Counter(), collections.Counter(), collections.Counter()
def resolve(pc):
    i = bisect.bisect_right(maps, (pc, 1 << 62, 0, "")) - 1
    if i < 0 or not (maps[i][0] <= pc < maps[i][1]): return "?", "?"
    lo, hi, off, lib = maps[i]
    t = table(lib); a = file_to_vaddr(lib, pc - lo + off)
    j = bisect.bisect_right(t, (a, "\xff")) - 1
    return os.path.basename(lib), (t[j][1][:110] if j >= 0 else "?")
for pc, ch in zip(pcs, chains):
    lib, fn = resolve(pc)
    by_lib[lib] += 1
    by_fn[(lib, fn)] += 1
    if ch and lib.startswith("libc"):   # who calls into libc: first frames outside libc
        names = [resolve(c) for c in ch]
        outer = [f"{l}:{f[:60]}" for l, f in names if not l.startswith("libc")][:3]
        by_chain[(fn[:40], " <- ".join(outer))] += 1
n = len(pcs)
print(f"{n} samples")
for k, v in by_lib.most_common(12): print(f"  {100 * v / n:5.1f}%  {k}")
print("functions:")
for (lib, fn), v in by_fn.most_common(top): print(f"  {100 * v / n:5.1f}%  {lib}: {fn}")
# hottest raw addresses inside libc with the exported symbols on both sides (libc's own functions are not in its dynamic symbol table)
raw = collections.Counter()
for pc in pcs:
    i = bisect.bisect_right(maps, (pc, 1 << 62, 0, "")) - 1
    if i >= 0 and maps[i][0] <= pc < maps[i][1] and os.path.basename(maps[i][3]).startswith("libc"):
        raw[(maps[i][3], file_to_vaddr(maps[i][3], pc - maps[i][0] + maps[i][2]))] += 1
if raw:
    print("hottest libc addresses:")
    for (lib, a), v in raw.most_common(12):
        t = table(lib); j = bisect.bisect_right(t, (a, "\xff")) - 1
        prev = f"{t[j][1]}+{a - t[j][0]:#x}" if j >= 0 else "?"
        nxt = f"{t[j + 1][1]}-{t[j + 1][0] - a:#x}" if j + 1 < len(t) else "?"
        print(f"  {100 * v / n:5.1f}%  {a:#x}  after {prev}, before {nxt}")
if by_chain:
    print("callers of libc samples:")
    for (fn, ch), v in by_chain.most_common(top): print(f"  {100 * v / n:5.1f}%  {fn} <- {ch}")
