"""fp32 vs --amp bf16: loss over N synthetic steps with identical seeds / data (the opt-in
mixed-precision mode).  python tools/amp_curve.py [steps=200]  -> stdout + gpurun_out/amp_curve.json
(profiles/r03_amp_curve.txt was made with this script when it still lived in the untracked scratch/ directory)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unsupervised_depth_opticalflow_egomotion_amd import convs, synthetic
from unsupervised_depth_opticalflow_egomotion_amd.models import get_model
from unsupervised_depth_opticalflow_egomotion_amd.train_step import make_cfg, make_optimizer, train_step
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
NB = 25
BATCHES = [[torch.from_numpy(a) for a in synthetic.make_triplet_batch(4, 256, 832, 3, seed=1234 + i)] for i in range(NB)]
def run(dt):
    convs.set_compute_dtype(dt)
    cfg = make_cfg(num_scales=3, img_hw=(256, 832), mode="geom")
    torch.manual_seed(1234)
    model = get_model("geom")(cfg).to(dev).train()
    opt = make_optimizer(model, cfg.lr)
    out = []
    t0 = time.time()
    for it in range(N):
        inputs = [a.to(dev) for a in BATCHES[it % NB]]
        loss, lp, _ = train_step(model, opt, inputs, cfg)
        out.append(float(loss))
    torch.cuda.synchronize()
    return np.array(out), time.time() - t0
a, ta = run(None)
b, tb = run(torch.bfloat16)
k = max(N // 10, 1)
print("steps %d | fp32: first %.5f last-%d mean %.5f (%.1f s) | bf16: first %.5f last-%d mean %.5f (%.1f s)" % (N, a[0], k, a[-k:].mean(), ta, b[0], k, b[-k:].mean(), tb))
print("relative difference of the last-%d mean: %.4f ; of the first step: %.2e ; max over the run of |bf16-fp32|/fp32: %.4f" % (
    k, abs(b[-k:].mean() - a[-k:].mean()) / a[-k:].mean(), abs(b[0] - a[0]) / a[0], float(np.max(np.abs(b - a) / a))))
json.dump({"fp32": a.tolist(), "bf16": b.tolist()}, open("gpurun_out/amp_curve.json", "w"))
