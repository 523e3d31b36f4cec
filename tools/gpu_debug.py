import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pmt_learning_for_semantic_segmentation_and_disparity_amd as pkg
from pmt_learning_for_semantic_segmentation_and_disparity_amd.nn import SpatialCorrelationSampler
maps = open('/proc/self/maps').read()
print(sorted({l.split()[-1] for l in maps.splitlines() if 'amdhip' in l or 'libsdhip' in l or 'hsa-runtime' in l}))
a = torch.zeros(1, 4, 1, 4); b = torch.zeros(1, 4, 1, 4)
a[0, 0, 0] = torch.tensor([1., 2, 3, 4]); b[0, 0, 0] = torch.tensor([10., 20, 30, 40])
y = SpatialCorrelationSampler(1, (1, 3))(a.cuda(), b.cuda())
torch.cuda.synchronize()
print(y.shape, y.stride(), y.cpu().flatten().tolist())
