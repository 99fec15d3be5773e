import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_depth_opticalflow_egomotion_amd import ops
B, ci, co, H, W = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 128, 128, 64, 208))]
dev = torch.device("cuda:0")
x = torch.randn(B, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
for _ in range(6): ops.wino_conv3x3(x, w, 1)
torch.cuda.synchronize()
