#!/bin/bash
# kernel trace of three graph replays of the benchmark step (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/step_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary "$@" > $OUT/run.log 2>&1
tail -2 $OUT/run.log | cut -c1-300
python tools/trace_summary.py $OUT/r_kernel_trace.csv > $OUT/summary.txt
head -60 $OUT/summary.txt
