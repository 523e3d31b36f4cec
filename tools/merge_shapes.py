"""Join gpurun_out/shapes.json with a rocprofv3 kernel trace CSV: GPU time per conv layer shape."""
import sys, csv, json, collections
log = json.load(open(sys.argv[1]))
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
fw = [r for r in rows if "conv_fwd_kernel" in r["Kernel_Name"] or "conv_fast_kernel" in r["Kernel_Name"]]
wg = [r for r in rows if "conv_wgrad_kernel" in r["Kernel_Name"] or "wgrad_fast_kernel" in r["Kernel_Name"]]
nf = sum(1 for d in log if d["k"] == "fwd"); nw = len(log) - nf
fw, wg = fw[-nf:], wg[-nw:]
agg = collections.OrderedDict()
fi = wi = 0
for d in log:
    if d["k"] == "fwd": r = fw[fi]; fi += 1
    else: r = wg[wi]; wi += 1
    key = (d["k"], d["B"] * d["Do"], d["Ho"], d["Wo"], d["Cin"], d["Cout"], "%dx%d" % (d["kh"], d["kw"]), d["kd"], d["stride"], d["dil"], d["pro"], d.get("stats", 0))
    a = agg.setdefault(key, [0, 0.0, r["Kernel_Name"][:60]])
    a[0] += 1; a[1] += dur(r)
tot = sum(a[1] for a in agg.values())
print("total conv us %.0f  (fwd-kernel %.0f, wgrad %.0f)" % (tot, sum(a[1] for k, a in agg.items() if k[0] == "fwd"), sum(a[1] for k, a in agg.items() if k[0] == "wgrad")))
print("kind  Bimg Ho Wo Cin Cout k kd s d pro stats | n  total_us  us/launch  ideal_us(mfma 2.5PF)  ideal_us(hbm 8TB/s)")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    k, B, Ho, Wo, Ci, Co, kk, kd, s, d, pro, st = key
    kh, kw = [int(v) for v in kk.split("x")]
    fl = 2.0 * B * Ho * Wo * Ci * Co * kh * kw * kd
    by = 2.0 * B * Ho * Wo * (Ci * s * s + Co)
    print("%-5s %4d %3d %3d %4d %4d %s %d %d %d %d %d | %3d %8.0f %8.1f %8.1f %8.1f" % (k, B, Ho, Wo, Ci, Co, kk, kd, s, d, pro, st, a[0], a[1], a[1] / a[0], fl / 2.5e15 * 1e6, by / 8e12 * 1e6))
