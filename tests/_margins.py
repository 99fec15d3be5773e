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


def geom_margins(inp, ac, S, signed=False):
    """Per-pixel decision margins of the oracle for a ``synthetic.LossStackInputs`` (numpy -> dict of lists)."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()   # noqa: E731
    m = O.GeomLossOracle(num_scales=S, align_corners=ac)
    mg = m.decision_margins(T(inp.imgs[0]), T(inp.imgs[1]), T(inp.imgs[2]), [T(a) for a in inp.disps[1]],
                            T(inp.pose), [T(a) for a in inp.flows_bwd], [T(a) for a in inp.flows_fwd], T(inp.K), signed=signed)
    return {k: [t.numpy() for t in v] for k, v in mg.items()}


POSE_FED = ("dyna_bwd", "dyna_fwd", "texture_bwd", "texture_fwd", "valid_to_l", "valid_to_r")   # decisions fed by R


def trig_floor(inp, ac, S):
    """Noise floor of the pose-fed decisions under a <= 1-ulp libm ("bit-exact given R").

    The reference's cos / sin are a vendor libm's (<= 1 ulp); the kernels use the correctly rounded values.  Each of
    the six values (cos / sin of rx, ry, rz) of a pose may therefore sit one fp32 ulp away from the kernel's.  Returns
    (base, floor, flips): the SIGNED margins lhs - rhs with correctly rounded trig, a per-pixel bound on how far such a
    libm can move them -- the sum over the six values of the larger of the +1 / -1 ulp responses, x 1.5 for the
    interaction of simultaneous shifts, plus 4 ulps of the margin's operands for re-rounding -- and, per mask, the
    number of pixels that really flipped in one of the twelve single-ulp runs (all of them must lie inside the floor:
    checked here, so the floor is validated on every use)."""
    with O.trig("cr"):
        base = geom_margins(inp, ac, S, signed=True)
    floor = {k: [np.zeros_like(a) for a in base[k]] for k in POSE_FED}
    flips = {k: 0 for k in POSE_FED}
    runs = []
    for slot in range(6):
        worst = {k: [np.zeros_like(a) for a in base[k]] for k in POSE_FED}
        for step in (1, -1):
            shift = [0] * 6
            shift[slot] = step
            with O.trig("cr", shift):
                m = geom_margins(inp, ac, S, signed=True)
            runs.append(m)
            for k in POSE_FED:
                for s in range(S):
                    worst[k][s] = np.maximum(worst[k][s], np.abs(m[k][s] - base[k][s]))
        for k in POSE_FED:
            for s in range(S):
                floor[k][s] += worst[k][s]
    for k in POSE_FED:
        for s in range(S):
            floor[k][s] = 1.5 * floor[k][s] + 5e-7
    for m in runs:
        for k in POSE_FED:
            for s in range(S):
                f = (m[k][s] < 0) != (base[k][s] < 0)
                flips[k] += int(f.sum())
                assert (np.abs(base[k][s][f]) <= floor[k][s][f]).all(), (k, s)
    return base, floor, flips


def within_counts(margins):
    """{mask: number of pixels (all scales) whose margin is below the family's TAU}."""
    return {k: int(sum((a < TAU[family(k)]).sum() for a in v)) for k, v in margins.items()}


def min_margins(margins):
    return {k: np.array([float(a.min()) for a in v]) for k, v in margins.items()}
