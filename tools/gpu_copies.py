"""Which ATen copy / elementwise kernels does one training step still launch?  (torch profiler, eager step)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, traceback, collections
import bench
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch
model = bench.build_model(torch.bfloat16, "minidsnetExt")
step = TrainStep(model, dtype=torch.bfloat16, use_graph=False)
batch = synthetic_batch(8, 256, 512)
for _ in range(2): step(*batch)
torch.cuda.synchronize()
orig = ops.nhwc_view
hits = collections.Counter()
def spy(x):
    y, ld = orig(x)
    if y.data_ptr() != x.data_ptr():
        fr = traceback.extract_stack(limit=6)
        hits[(tuple(x.shape), tuple(x.stride()), " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-5:-1]))] += 1
    return y, ld
ops.nhwc_view = spy
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(*batch)
    torch.cuda.synchronize()
print("nhwc_view copies:")
for k, v in hits.most_common(20): print("  ", v, k)
rows = [e for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
print("aten ops by device time:")
for e in rows[:18]: print("   %-40s n=%4d  device %.0f us" % (e.key, e.count, e.device_time_total))
