#!/bin/bash
# stall / issue counters of the dominant 5x5 kernel (bench.py --roofline-only), run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_conv5
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python bench.py --roofline-only > $OUT/trace.log 2>&1
cat $OUT/trace/*kernel_stats.csv 2>/dev/null | head -6; find $OUT/trace -name '*kernel_stats.csv' | head -2 | xargs -r head -6
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o r -- python bench.py --roofline-only > $OUT/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_conv5/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "conv_fast_kernel" in r["Kernel_Name"] or "conv_band_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-36s avg per launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
