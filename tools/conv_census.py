"""Census of the MIOpen convolution calls of one training step and what each costs in isolation.

  python tools/conv_census.py [--mode geom] [--batch 4]          # on an MI355X box

A TorchDispatchMode records every aten::convolution / aten::convolution_backward of one forward + backward of the model
(including the ones issued from inside this build's autograd operators), then each distinct call is replayed alone and
timed with device events -- the time includes whatever MIOpen launches around its kernel (NCHW<->NHWC transposes, split-K
zero fills).  Output: one row per distinct call, sorted by time per step.  wrw = weight gradient, dgrad = data gradient."""
import argparse
import collections
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.calls = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if name == "aten::convolution":
            x, w, b, stride, pad, dil = args[:6]
            self.calls[("fwd", tuple(x.shape), tuple(w.shape), tuple(stride), tuple(pad), tuple(dil), b is not None)] += 1
        elif name == "aten::convolution_backward":
            gy, x, w, bs, stride, pad, dil, tr, opad, groups, mask = args[:11]
            for i, kind in ((0, "dgrad"), (1, "wrw")):
                if mask[i]:
                    self.calls[(kind, tuple(x.shape), tuple(w.shape), tuple(stride), tuple(pad), tuple(dil), bool(mask[2]))] += 1
        return func(*args, **(kwargs or {}))


def time_call(key, reps=30):
    kind, xs, ws, stride, pad, dil, flag = key
    dev = torch.device("cuda:0")
    x = torch.randn(*xs, device=dev)
    w = torch.randn(*ws, device=dev)
    bias = torch.randn(ws[0], device=dev) if (kind == "fwd" and flag) else None
    y = torch.nn.functional.conv2d(x, w, bias, stride, pad, dil)
    gy = torch.randn_like(y)
    if kind == "fwd":
        fn = lambda: torch.nn.functional.conv2d(x, w, bias, stride, pad, dil)
    else:
        mask = [kind == "dgrad", kind == "wrw", False]
        fn = lambda: torch.ops.aten.convolution_backward(gy, x, w, None, list(stride), list(pad), list(dil), False, [0, 0], 1, mask)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, 2.0 * y.numel() * ws[1] * ws[2] * ws[3]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="geom")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--only", default="", help="comma list of kinds to time (fwd,dgrad,wrw); default all")
    args = ap.parse_args()
    from unsupervised_depth_opticalflow_egomotion_amd import synthetic
    from unsupervised_depth_opticalflow_egomotion_amd.models import get_model
    from unsupervised_depth_opticalflow_egomotion_amd.train_step import make_cfg, make_optimizer, train_step
    dev = torch.device("cuda:0")
    cfg = make_cfg(num_scales=3, img_hw=(args.height, args.width), mode=args.mode)
    torch.manual_seed(1234)
    model = get_model(args.mode)(cfg).to(dev).train()
    opt = make_optimizer(model, cfg.lr)
    inputs = [torch.from_numpy(a).to(dev) for a in synthetic.make_triplet_batch(args.batch, args.height, args.width, 3, seed=1234)]
    train_step(model, opt, inputs, cfg)
    torch.cuda.synchronize()
    with Census() as c:
        train_step(model, opt, inputs, cfg)
    torch.cuda.synchronize()
    kinds = set(args.only.split(",")) if args.only else {"fwd", "dgrad", "wrw"}
    rows = []
    for key, n in c.calls.items():
        if key[0] not in kinds:
            continue
        us, flop = time_call(key)
        rows.append((n * us, n, us, flop, key))
    rows.sort(reverse=True)
    tot = collections.Counter()
    print("%-5s %3s %9s %9s %7s  %-22s %-18s %s" % ("kind", "n", "us/call", "us/step", "TF/s", "x", "w", "stride/pad/dil"))
    for t, n, us, flop, key in rows:
        kind, xs, ws, stride, pad, dil, flag = key
        tot[kind] += t
        print("%-5s %3d %9.1f %9.1f %7.1f  %-22s %-18s %s/%s/%s" % (kind, n, us, t, flop / us / 1e6, "x".join(map(str, xs)),
                                                                   "x".join(map(str, ws)), stride[0], pad[0], dil[0]))
    for k, v in tot.items():
        print("total %-5s %.2f ms/step" % (k, v / 1e3))


if __name__ == "__main__":
    main()
