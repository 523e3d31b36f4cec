#!/bin/bash
# sweep host-side heuristics of the conv kernels on the real training step (diagnostic env overrides):  bash tools/gpu_tune.sh [model]
M=${1:-minidsnetExt}
run() { echo -n "[$*] "; env "$@" timeout -k 10 200 python bench.py --model $M --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | grep -o 'ms_per_step": [0-9.]*'; }
run X=0
run SDHIP_TUNE_KSOFT_SMALL=53
run SDHIP_TUNE_KSOFT_SMALL=40
run SDHIP_TUNE_KSOFT_SMALL=160
run SDHIP_TUNE_KSOFT_BIG=160
run SDHIP_TUNE_BIG=256
run SDHIP_TUNE_BIG=1024
run SDHIP_TUNE_BIG=2048
run SDHIP_TUNE_SPLIT=512
run SDHIP_TUNE_SPLIT=2048
run SDHIP_TUNE_ATOMIC_TBS=0.6
run SDHIP_TUNE_ATOMIC_TBS=2.6
run SDHIP_TUNE_ATOMIC_TBS=0.3
run SDHIP_TUNE_S2_SMALL=0
run X=0
