"""Decision-margin bookkeeping for the bit-exact mask contract.

A mask is a threshold on fp32 arithmetic.  The HIP kernels evaluate the reference's expressions in the reference's
association order on identical fp32 inputs, so their masks are bit-identical to the oracle's except where a
transcendental (the softmax's ``exp``: <= 1 ulp in both libraries) feeds the decision.  TAU is the noise floor per
mask family: a HIP mask may differ from the oracle's ONLY at pixels whose margin |lhs - rhs| is below it, and the
seeds the parity tests use are chosen so that no pixel is (``within == 0``) -- for them the masks must be equal."""
import numpy as np
import torch

from oracle import loss_stack_oracle as O

# occ: |w - 0.48| of the softmax weight w = 1 - e_i / (e_l + e_r).  exp differs by <= 1-2 ulp between the host
# library (SLEEF / MKL) and ocml, which moves w by <= 1e-7 (w ~ 0.5, ulp 3e-8): 2e-7 is the noise floor.
# Everything else (validity, dynamic, texture, inverse_warp2 validity) is correctly rounded IEEE arithmetic
# (+, -, *, /, sqrt, fma) in the reference's order on identical inputs: tolerance 0, they must be EQUAL.
TAU = dict(occ=2e-7, valid=0.0, dyna=0.0, texture=0.0, valid_to=0.0)


def family(name):
    for f in ("valid_to", "valid", "occ", "dyna", "texture"):
        if name.startswith(f):
            return f
    raise KeyError(name)


def geom_margins(inp, ac, S):
    """Per-pixel decision margins of the oracle for a ``synthetic.LossStackInputs`` (numpy -> dict of lists)."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()   # noqa: E731
    m = O.GeomLossOracle(num_scales=S, align_corners=ac)
    mg = m.decision_margins(T(inp.imgs[0]), T(inp.imgs[1]), T(inp.imgs[2]), [T(a) for a in inp.disps[1]],
                            T(inp.pose), [T(a) for a in inp.flows_bwd], [T(a) for a in inp.flows_fwd], T(inp.K))
    return {k: [t.numpy() for t in v] for k, v in mg.items()}


def within_counts(margins):
    """{mask: number of pixels (all scales) whose margin is below the family's TAU}."""
    return {k: int(sum((a < TAU[family(k)]).sum() for a in v)) for k, v in margins.items()}


def min_margins(margins):
    return {k: np.array([float(a.min()) for a in v]) for k, v in margins.items()}
