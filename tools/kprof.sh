#!/bin/bash
# kernel-time table of any python command on the GPU box:  bash tools/kprof.sh <grep pattern> <script.py> [args...]
pat="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kprof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python "$@" > $OUT/run.log 2>&1
python - "$pat" <<'PY'
import csv, sys, re
pat = sys.argv[1]
rows = list(csv.DictReader(open("gpurun_out/kprof/r_kernel_stats.csv")))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "").split("(")[0]
    if re.search(pat, n):
        print("%-60s calls %5s  avg %9.1f us  min %9.1f  max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
