"""Parity report of the fused loss stack against the oracle: flip counts and largest errors per case.

  python tools/parity_report.py > gpurun_out/r03_parity_report.txt          # on an MI355X box

The numbers quoted as "measured headroom" next to the tolerances of tests/test_hip_loss_stack.py come from here.
Per case: mask decisions that differ / pixels inside the noise floors, the largest relative error over the loss
rows, the largest element-wise gradient error relative to the gradient's scale (disparities, flows) and of the pose
gradient.  Conditioned cases compare with the oracle as the tests do; `raw` cases use unconditioned poses and compare
with the oracle twice: cos / sin correctly rounded (what the device computes) and the host's libm (what the reference
does on this machine)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from unsupervised_depth_opticalflow_egomotion_amd import synthetic  # noqa: E402

synthetic.CONDITION_POSE = True
from oracle import loss_stack_oracle as O  # noqa: E402
from tests import _margins as M  # noqa: E402
from tests import test_hip_loss_stack as TS  # noqa: E402

N = TS.N


def errors(inp, ac, S, depth_terms=False):
    lp_h, mk_h, tot_h, (dh, ph, fbh, ffh) = TS.run_hip(inp, ac, S, None, depth_terms)
    lp_o, mk_o, tot_o, (do, po, fbo, ffo) = TS.run_oracle(inp, ac, S, None, depth_terms)
    margins = M.geom_margins(inp, ac, S)
    within = M.within_counts({k: margins[k] for k in TS.MASKS})
    nflip = sum(int((N(mk_h[k][s]) != N(mk_o[k][s])).sum()) for k in TS.MASKS for s in range(S))
    npx = sum(N(mk_h[k][s]).size for k in TS.MASKS for s in range(S))
    loss_err = max(float(np.abs(N(lp_h[k]) - N(lp_o[k])).max() / max(np.abs(N(lp_o[k])).max(), 1e-12)) for k in lp_h)

    def g(a, b):
        a, b = N(a.grad), N(b.grad)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
    gd = max(g(dh[f][s], do[f][s]) for f in range(3) for s in range(S))
    gf = max(max(g(fbh[s], fbo[s]), g(ffh[s], ffo[s])) for s in range(S))
    gp = g(ph, po)
    return nflip, npx, sum(within.values()), loss_err, gd, gf, gp


def main():
    print("fused loss stack vs oracle (device: HIP kernels through the C ABI; oracle on this host's CPU)")
    print("%-46s %6s %9s %7s %10s %10s %10s %10s" % ("case", "flips", "decisions", "inside", "loss rel", "gdisp", "gflow", "gpose"))
    cases = [("strict %s ac=%d" % (sh, ac), synthetic.make_loss_stack_inputs(*sh, 3, seed=sd), ac, 3, False)
             for sh, sd in TS.STRICT.items() for ac in (0, 1)]
    cases += [("B=4 256x832 (configs[2]) ac=%d" % ac, synthetic.make_loss_stack_inputs(4, 256, 832, 3, seed=1234), ac, 3, False) for ac in (0, 1)]
    cases += [("B=4 256x832 + depth terms ac=0", synthetic.make_loss_stack_inputs(4, 256, 832, 3, seed=1234), 0, 3, True)]
    cases += [("B=2 375x1242 S=6 (configs[4]) ac=%d" % ac, synthetic.make_loss_stack_inputs(2, 375, 1242, 6, seed=55, num_flow_scales=6), ac, 6, False) for ac in (0, 1)]
    for name, inp, ac, S, dt in cases:
        print("%-46s %6d %9d %7d %10.2e %10.2e %10.2e %10.2e" % ((name,) + errors(inp, bool(ac), S, dt)))
    print()
    print("unconditioned poses: device vs oracle with correctly rounded cos/sin and exp (cr, exp cr), cos/sin only (cr), and with this host's libm (host)")
    for kind, shape, seed in (("sigma", (2, 128, 448), 12), ("sigma", (4, 256, 832), 16), ("posecnn", (1, 256, 832), 15)):
        for ac in (0, 1):
            inp = TS.raw_pose_inputs(kind, shape, seed)
            with O.trig("cr"), O.occ_exp("cr"):      # round 6: the canonical function of the occlusion softmax as well
                r = errors(inp, bool(ac), 3)
            print("%-46s %6d %9d %7d %10.2e %10.2e %10.2e %10.2e" % (("raw %s %s ac=%d  [cr, exp cr]" % (kind, shape, ac),) + r))
            with O.trig("cr"):
                r = errors(inp, bool(ac), 3)
            print("%-46s %6d %9d %7d %10.2e %10.2e %10.2e %10.2e" % (("raw %s %s ac=%d  [cr]" % (kind, shape, ac),) + r))
            r = errors(inp, bool(ac), 3)
            print("%-46s %6d %9d %7d %10.2e %10.2e %10.2e %10.2e" % (("raw %s %s ac=%d  [host]" % (kind, shape, ac),) + r))


if __name__ == "__main__":
    main()
