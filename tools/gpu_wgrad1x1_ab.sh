#!/bin/bash
# A/B of the chunk-packed 1x1 weight gradient against the unpacked kernel: kernel-only durations from rocprofv3 traces.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/w1_new -o w -- python3 $R/tools/gpu_wgrad1x1.py > $R/gpurun_out/w1_new.log 2>&1
export SDHIP_WGRAD_NO_PACK=1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/w1_old -o w -- python3 $R/tools/gpu_wgrad1x1.py > $R/gpurun_out/w1_old.log 2>&1
echo done
