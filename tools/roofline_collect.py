"""Turn gpurun_out/roofline/* (tools/roofline_profile.sh) into the committed profiles/<round tag>_* files."""
import collections, csv, json, os, re, shutil
R = "gpurun_out/roofline"
TAG = os.environ.get("SDHIP_PROFILE_TAG", "r03")
PD = os.environ.get("SDHIP_PROFILE_DIR", "gpurun_out/profiles")     # merged back by gpurun; copy into profiles/ to commit
os.makedirs(PD, exist_ok=True)


def kstats(path, out, top=40):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(out, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows[:top]:
            f.write('"%s",%s,%s,%s,%s\n' % (r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
    return rows


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:80]


rows = kstats(R + "/stats/r_kernel_stats.csv", PD + "/%s_roofline_kernel_stats.csv" % TAG)
kstats(R + "/step/r_kernel_stats.csv", PD + "/%s_bench_graph_kernel_stats.csv" % TAG, 60)
shutil.copy(R + "/step_summary.txt", PD + "/%s_step_summary.txt" % TAG)
DOM = "conv_band_kernel" if any("conv_band_kernel" in r["Name"] for r in rows) else "conv_fast_kernel"
conv = [r for r in rows if DOM in r["Name"]][0]
avg = lambda acc, pat, name: (lambda v: (sum(v) / len(v), len(v)))([x for k, d in acc.items() if pat in k for x in d.get(name, [])])
fetch, nf = avg(counters(R + "/fetch/r_counter_collection.csv"), DOM, "FETCH_SIZE")
write, nw = avg(counters(R + "/write/r_counter_collection.csv"), DOM, "WRITE_SIZE")
m1 = counters(R + "/mfma/r_counter_collection.csv")
busy, _ = avg(m1, DOM, "SQ_VALU_MFMA_BUSY_CYCLES")
sqbusy, _ = avg(m1, DOM, "SQ_BUSY_CYCLES")
line = [l for l in open(R + "/stats.log") if l.startswith("{")][-1]
roof = json.loads(line)["roofline"]
# FETCH_SIZE / WRITE_SIZE are reported in KiB.  FETCH_SIZE counts read requests x 64 bytes: a kernel that reads whole 128-byte
# lines (conv_fast: 128-byte LDS rows) must be doubled (MI355X_MICROARCH.md §HBM); the 64-channel band kernel fetches every
# pixel as two 64-byte halves in different chunks — 64-byte requests, counted exactly (calibrated against the 32-channel
# layer and the halo arithmetic: profiles/r02_band_traffic_calibration.txt)
fetch_scale = 1.0 if DOM == "conv_band_kernel" else 2.0
hbm = fetch_scale * fetch * 1024 + write * 1024
dur_cycles = float(conv["AverageNs"]) * 2.4            # 2.4 GHz shader clock
# SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1024 SIMDs, the cycles a SIMD's matrix pipe was busy
mfma_frac = busy / (dur_cycles * 1024.0)
out = {"kernel": conv["Name"][:160], "avg_ns_rocprof": float(conv["AverageNs"]), "calls": int(conv["Calls"]),
       "ms_per_launch_bench": roof["ms_per_launch"], "achieved_tflops_bench": roof["achieved"],
       "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write, "launches_fetch": nf, "launches_write": nw,
       "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 2 * 8 * 256 * 512 * 64 * 2 + 64 * 64 * 25 * 2,
       "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy, "SQ_BUSY_CYCLES_per_launch": sqbusy, "mfma_busy_frac": round(mfma_frac, 4),
       "fetch_scale": fetch_scale,
       "note": "hbm = fetch_scale*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes, bench.py --roofline-only; fetch_scale 1 for the "
               "64-byte-request pattern of the band kernel (profiles/r02_band_traffic_calibration.txt), 2 for whole-line readers; mfma_busy_frac = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles at 2.4 GHz x 1024 SIMDs)"}
json.dump(out, open(PD + "/%s_roofline_traffic.json" % TAG, "w"), indent=1)
print(json.dumps(out, indent=1))

# per-kernel-family table of one eager step: duration (kernel trace of the graph run), MFMA busy, HBM bytes
sm = counters(R + "/step_mfma/r_counter_collection.csv")
sf = counters(R + "/step_fetch/r_counter_collection.csv")
sw = counters(R + "/step_write/r_counter_collection.csv")
fam = collections.defaultdict(lambda: dict(n=0, busy=0.0, sq=0.0, fetch=0.0, write=0.0))
for k, d in sm.items():
    f = fam[short(k)]
    f["n"] += len(d.get("SQ_BUSY_CYCLES", [])); f["busy"] += sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", [])); f["sq"] += sum(d.get("SQ_BUSY_CYCLES", []))
for k, d in sf.items():
    fam[short(k)]["fetch"] += sum(d.get("FETCH_SIZE", []))
for k, d in sw.items():
    fam[short(k)]["write"] += sum(d.get("WRITE_SIZE", []))
# average duration per launch of each family, from the kernel trace of the graph-replayed run
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(R + "/step/r_kernel_stats.csv")):
    d = dur[short(r["Name"])]
    d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
with open(PD + "/%s_step_counters.csv" % TAG, "w") as f:
    f.write("kernel,launches (3 eager steps),SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,hbm_MB (fetch_scale*FETCH+WRITE),"
            "hbm_MB_per_launch,avg_us_per_launch (graph run),GB/s,frac_of_8TB/s,fetch_scale\n")
    fs = lambda k: 1.0 if "conv_band_kernel<5, 64" in k else 2.0      # 64-byte read requests: r02_band_traffic_calibration.txt
    for k, v in sorted(fam.items(), key=lambda kv: -(fs(kv[0]) * kv[1]["fetch"] + kv[1]["write"])):
        mb = (fs(k) * v["fetch"] + v["write"]) * 1024 / 1e6
        n = max(v["n"], 1)
        us = dur[k][1] / dur[k][0] / 1e3 if dur[k][0] else 0.0
        gbs = (mb / n) / us * 1e3 if us else 0.0
        f.write('"%s",%d,%.4g,%.4g,%.1f,%.2f,%.1f,%.0f,%.3f,%.0f\n' % (k, v["n"], v["busy"], v["sq"], mb, mb / n, us, gbs, gbs / 8000.0, fs(k)))
print(open(PD + "/%s_step_counters.csv" % TAG).read()[:3000])
