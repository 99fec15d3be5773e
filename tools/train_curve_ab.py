"""Loss curve of N synthetic fp32 training steps (identical seeds / data) under two environments, one child process each:

    python tools/train_curve_ab.py 200 DFE_PLANECONV_MAX_HW=0 DFE_PLANECONV_MAX_HW=208

Used in round 4 to show that moving PWC decoder levels 6 / 5 (and PoseCNN's refinement convolutions) from MIOpen to the
small-plane MFMA kernels leaves training where it was (profiles/r04_planeconv_curve.txt): the two curves differ by what two
MIOpen runs of the same step differ by (its split-K weight gradients use float atomics)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, json, time
sys.path.insert(0, %r)
import numpy as np, torch
from unsupervised_depth_opticalflow_egomotion_amd import synthetic
from unsupervised_depth_opticalflow_egomotion_amd.models import get_model
from unsupervised_depth_opticalflow_egomotion_amd.train_step import make_cfg, make_optimizer, train_step
N = int(sys.argv[1]); dev = torch.device("cuda:0"); NB = 25
B = [[torch.from_numpy(a) for a in synthetic.make_triplet_batch(4, 256, 832, 3, seed=1234 + i)] for i in range(NB)]
cfg = make_cfg(num_scales=3, img_hw=(256, 832), mode="geom")
torch.manual_seed(1234)
model = get_model("geom")(cfg).to(dev).train()
opt = make_optimizer(model, cfg.lr)
out = []
for it in range(N):
    loss, lp, _ = train_step(model, opt, [a.to(dev) for a in B[it %% NB]], cfg)
    out.append(float(loss))
print("CURVE " + json.dumps(out))
''' % ROOT


def run(n, assignment):
    env = dict(os.environ)
    k, v = assignment.split("=", 1)
    env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, str(n)], env=env, capture_output=True, text=True, check=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("CURVE ")][-1]
    return json.loads(line[6:])


def main():
    import numpy as np
    n = int(sys.argv[1]); ea, eb = sys.argv[2], sys.argv[3]
    a, a2, b = np.array(run(n, ea)), np.array(run(n, ea)), np.array(run(n, eb))
    k = max(n // 10, 1)
    rel = lambda u, v: float(np.max(np.abs(u - v) / np.abs(v)))
    print("steps %d | A = %s: first %.6f, mean of the last %d %.6f | B = %s: first %.6f, mean of the last %d %.6f" % (
        n, ea, a[0], k, a[-k:].mean(), eb, b[0], k, b[-k:].mean()))
    print("max over the run of |B - A| / A: %.2e (first step %.2e, last-%d mean %.2e); the same between two runs of A: %.2e (last-%d mean %.2e)" % (
        rel(b, a), abs(b[0] - a[0]) / a[0], k, abs(b[-k:].mean() - a[-k:].mean()) / a[-k:].mean(), rel(a2, a), k,
        abs(a2[-k:].mean() - a[-k:].mean()) / a[-k:].mean()))
    print("every 20th step  A: " + " ".join("%.5f" % v for v in a[::20]))
    print("every 20th step  B: " + " ".join("%.5f" % v for v in b[::20]))


if __name__ == "__main__":
    main()
