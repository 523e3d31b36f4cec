#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the persistent 5x5 kernel on two access patterns with known byte counts (run on the GPU box
# from the repo root): 64->64 channels (the halo is fetched as 64-byte half pixels, 128 bytes apart) and 32->32 (whole
# 64-byte pixels, contiguous).  Calibrates the "x2 on gfx950" rule of MI355X_MICROARCH.md §HBM for the first pattern.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/band_traffic
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r -- python tools/gpu_bandbench.py --dbg > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r -- python tools/gpu_bandbench.py --dbg > $OUT/write.log 2>&1
python - <<'PY'
import csv, collections
acc = collections.defaultdict(list)
for kind in ("fetch", "write"):
    for r in csv.DictReader(open("gpurun_out/band_traffic/%s/r_counter_collection.csv" % kind)):
        if "conv_band_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("conv_band_kernel")[1][:12], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-14s %-10s %8.1f MiB per launch (raw counter, KiB/1024; n=%d)" % (k[0], k[1], sum(v) / len(v) / 1024, len(v)))
print("known bytes at B=8 256x512: 64->64 input 128 MiB (x1.41 halo = 180), output 128 MiB; 32->32 input 64 MiB (x1.41 = 90), output 64 MiB")
PY
