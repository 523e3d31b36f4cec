#!/bin/bash
# Profile of the bench's dominant-kernel microbenchmark (run on the GPU box from the repo root):
#   1. rocprofv3 --kernel-trace --stats  -> per-kernel average duration (must agree with bench.py's roofline.ms_per_launch)
#   2. rocprofv3 --pmc FETCH_SIZE        -> HBM read bytes   (separate pass; x2 on gfx950, MI355X_MICROARCH.md §HBM)
#   3. rocprofv3 --pmc WRITE_SIZE        -> HBM write bytes  (separate pass)
# plus the kernel statistics of one whole bench run.  Outputs under gpurun_out/roofline/; tools/roofline_collect.py
# turns them into profiles/r02_* (committed).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/roofline
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python bench.py --roofline-only > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r -- python bench.py --roofline-only > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r -- python bench.py --roofline-only > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma -o r -- python bench.py --roofline-only > $OUT/mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/mfma2 -o r -- python bench.py --roofline-only > $OUT/mfma2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step -o r -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/step.log 2>&1
python tools/trace_summary.py $OUT/step/r_kernel_trace.csv > $OUT/step_summary.txt
# MFMA-busy of the kernels of a whole (eager) step: conv_fast / wgrad_fast families
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/step_mfma -o r -- python bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $OUT/step_mfma.log 2>&1
# HBM bytes of the BatchNorm elementwise kernels of a whole (eager) step
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/step_fetch -o r -- python bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $OUT/step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/step_write -o r -- python bench.py --steps 1 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-secondary > $OUT/step_write.log 2>&1
python tools/roofline_collect.py
# the secondary configurations' steps (dsnet = BASELINE config 2 as literally named, PSMNet(192) = config 3)
for m in dsnet psmnet; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_$m -o r -- python bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/step_$m.log 2>&1
  python tools/trace_summary.py $OUT/step_$m/r_kernel_trace.csv > ${SDHIP_PROFILE_DIR:-gpurun_out/profiles}/${SDHIP_PROFILE_TAG:-r03}_${m}_step_summary.txt
done
# the per-GPU workloads of BASELINE configs 4 (PSMNet(192) 960x512 B=4) and 5 (minidsnetExt aspp=2 hanet=1 19 classes, 1024x512 B=4)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_psm4 -o r -- python bench.py --model psmnet --batch 4 --height 512 --width 960 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/step_psm4.log 2>&1
python tools/trace_summary.py $OUT/step_psm4/r_kernel_trace.csv > ${SDHIP_PROFILE_DIR:-gpurun_out/profiles}/${SDHIP_PROFILE_TAG:-r03}_psmnet_cfg4_step_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_cfg5 -o r -- python bench.py --model minidsnetExt_cfg5 --batch 4 --height 512 --width 1024 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $OUT/step_cfg5.log 2>&1
python tools/trace_summary.py $OUT/step_cfg5/r_kernel_trace.csv > ${SDHIP_PROFILE_DIR:-gpurun_out/profiles}/${SDHIP_PROFILE_TAG:-r03}_cfg5_step_summary.txt
tail -1 $OUT/stats.log
