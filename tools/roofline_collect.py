"""Turn gpurun_out/roofline/* (tools/roofline_profile.sh) into the committed profiles/r01_roofline_* files."""
import csv, json, os, re, sys
R = "gpurun_out/roofline"
os.makedirs("profiles", exist_ok=True)
def kstats(path, out, top=25):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(out, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for r in rows[:top]:
            f.write('"%s",%s,%s,%s,%s\n' % (r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
    return rows
rows = kstats(R + "/stats/r_kernel_stats.csv", "profiles/r01_roofline_kernel_stats.csv")
kstats(R + "/step/r_kernel_stats.csv", "profiles/r01_bench_graph_kernel_stats.csv", 40)
conv = [r for r in rows if "conv_fast_kernel" in r["Name"]][0]
def pmc(path, name):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == name and "conv_fast_kernel" in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
fetch, nf = pmc(R + "/fetch/r_counter_collection.csv", "FETCH_SIZE")
write, nw = pmc(R + "/write/r_counter_collection.csv", "WRITE_SIZE")
line = [l for l in open(R + "/stats.log") if l.startswith("{")][-1]
roof = json.loads(line)["roofline"]
# FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950 -> x2
hbm = 2.0 * fetch * 1024 + write * 1024
out = {"kernel": conv["Name"][:160], "avg_ns_rocprof": float(conv["AverageNs"]), "calls": int(conv["Calls"]),
       "ms_per_launch_bench": roof["ms_per_launch"], "achieved_tflops_bench": roof["achieved"],
       "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write, "launches_fetch": nf, "launches_write": nw,
       "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 2 * 8 * 256 * 512 * 64 * 2 + 64 * 64 * 25 * 2,
       "note": "hbm = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes, bench.py --roofline-only"}
json.dump(out, open("profiles/r01_roofline_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
