"""MFMA-busy per kernel family of one eager step: joins a `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES` pass with the kernel
trace of the graph-replayed run of the same workload.
  python tools/pmc_family.py <counter_collection.csv> <kernel_stats.csv> <out.csv>"""
import collections, csv, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:90]


pmc, stats, out = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(pmc)):
    acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(stats)):
    d = dur[short(r["Name"])]
    d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
rows = []
for k, d in acc.items():
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", [])
    if not busy or not dur[k][0]:
        continue
    per = sum(busy) / len(busy)
    us = dur[k][1] / dur[k][0] / 1e3
    # SQ_VALU_MFMA_BUSY_CYCLES sums the matrix-pipe busy cycles of the chip's 1024 SIMDs; 2.4 GHz shader clock
    rows.append((k, len(busy), per, us, per / (us * 1e-6 * 2.4e9 * 1024) if us else 0.0, dur[k][1] / 1e3))
rows.sort(key=lambda r: -r[5])
with open(out, "w") as f:
    f.write("kernel,launches (one eager step),SQ_VALU_MFMA_BUSY_CYCLES per launch,avg_us per launch (graph run),mfma_busy_frac,total_us (graph run)\n")
    for r in rows:
        f.write('"%s",%d,%.4g,%.1f,%.3f,%.0f\n' % r)
print(open(out).read()[:2500])
