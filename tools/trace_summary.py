"""Summarise a rocprofv3 kernel trace of bench.py: time per kernel family and per (kernel, grid) of the LAST graph replay."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last replay = the last contiguous run of kernels: find the Adam kernel launches and take the span between the last two
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = (adam[-2] + 1, adam[-1] + 1) if len(adam) >= 2 else (0, len(rows))
step = rows[lo:hi]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]
fam = collections.defaultdict(lambda: [0, 0.0])
shape = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = short(r["Kernel_Name"])
    fam[k][0] += 1; fam[k][1] += d
    g = (k, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("LDS_Block_Size", ""))
    shape[g][0] += 1; shape[g][1] += d
    tot += d
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
print("last replay: %d kernels, sum of durations %.2f ms, span %.2f ms" % (len(step), tot / 1e3, span / 1e3))
print("\n== by kernel ==")
for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us %5d x %7.1f  %s" % (t, n, t / n, k))
print("\n== by kernel + grid ==")
for g, (n, t) in sorted(shape.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%8.1f us %4d x %7.1f  %s grid(%s,%s,%s) lds %s" % (t, n, t / n, g[0], g[1], g[2], g[3], g[4]))
