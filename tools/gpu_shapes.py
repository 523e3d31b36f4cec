"""Log every conv launch (forward / data-grad / weight-grad) of ONE eager training step in call order.

Run under `rocprofv3 --kernel-trace`: the k-th conv_fwd_kernel / conv_wgrad_kernel dispatch of the last step is the
k-th entry of gpurun_out/shapes.json, so tools/merge_shapes.py can attribute GPU time to layer shapes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pmt_learning_for_semantic_segmentation_and_disparity_amd import ops
from pmt_learning_for_semantic_segmentation_and_disparity_amd.train import TrainStep, synthetic_batch

model_name = sys.argv[1] if len(sys.argv) > 1 else "minidsnetExt"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dtype = torch.bfloat16
model = bench.build_model(dtype, model_name)
loss_fn = None
if model_name == "psmnet":
    loss_fn = lambda outs, seg, disp: ops.mean_l1_loss(outs, disp[:, 0])
step = TrainStep(model, dtype=dtype, use_graph=False, loss_fn=loss_fn)
batch = synthetic_batch(B, 256, 512)
for _ in range(2):
    step(*batch)
torch.cuda.synchronize()
log = []
orig = ops.call
FWD = "B H W Cin ldx Ho Wo Cout ldy kh kw stride dil pad_t pad_l D Do kd sd pad_d in_relu groups act accumulate dtype".split()
WG = "B H W Cin ldx Ho Wo Cout lddy kh kw stride dil pad_t pad_l D Do kd sd pad_d in_relu groups prezeroed dtype".split()
def spy(name, *args):
    if name == "sdhip_conv2d_fwd":
        d = dict(zip(FWD, [int(v) for v in args[9:9 + len(FWD)]]))
        d["k"] = "fwd"; d["pro"] = int(bool(args[4].value)); d["stats"] = int(bool(args[6].value)); d["bias"] = int(bool(args[3].value))
        log.append(d)
    elif name == "sdhip_conv2d_wgrad":
        d = dict(zip(WG, [int(v) for v in args[6:6 + len(WG)]]))
        d["k"] = "wgrad"; d["pro"] = int(bool(args[4].value))
        log.append(d)
    return orig(name, *args)
ops.call = spy
step(*batch)
torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(log, open("gpurun_out/shapes.json", "w"))
print("logged", len(log), "conv launches")
