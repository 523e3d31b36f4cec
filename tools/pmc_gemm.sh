#!/bin/bash
# counters of the streaming 1x1 GEMM on one shape (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
ARGS="$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python tools/gpu_gemm_one.py $ARGS > $OUT/stats.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -o r -- python tools/gpu_gemm_one.py $ARGS > $OUT/$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_gemm/*/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm1x1" in r["Name"]:
            print("stats:", r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"])
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_gemm/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm1x1" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-36s avg per launch %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
