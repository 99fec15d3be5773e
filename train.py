#!/usr/bin/env python3
"""Training driver with the reference's CLI surface (train.py:227-250) on the MI355X engine.

  python train.py -c config/kitti_geom.yaml --mode geom --model_dir ./models --batch_size 4
  torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py -c config/kitti_geom.yaml --mode geom

Differences from the reference, all forced by what it cannot do: data parallelism is one process per GPU
with one process per GPU over RCCL (ddp.wrap: one flat gradient all-reduce per step) instead of nn.DataParallel (``--multi_gpu`` is accepted and ignored; launch with
torchrun); the data source is the synthetic KITTI-shaped triplet generator (no KITTI on this machine);
periodic KITTI evaluation (train.py:136-164) needs the datasets and is skipped (``--no_test`` semantics).
Checkpoints keep the reference format: {iteration, model_state_dict, optimizer_state_dict} in
iter_{N}.pth and last.pth (train.py:21-29)."""
import argparse
import os
import pickle
import shutil
import time

import torch
import yaml

from unsupervised_depth_opticalflow_egomotion_amd import ddp, ops, synthetic
from unsupervised_depth_opticalflow_egomotion_amd.models import get_model
from unsupervised_depth_opticalflow_egomotion_amd.train_step import train_step, make_optimizer, GraphedTrainStep, LOSS_WEIGHT_ATTR


class pObject(object):
    pass


def save_model(iter_, model_dir, filename, model, optimizer):
    torch.save({"iteration": iter_, "model_state_dict": ddp.unwrap(model).state_dict(),
                "optimizer_state_dict": optimizer.state_dict()}, os.path.join(model_dir, filename))


def load_model(model_dir, filename, model, optimizer):
    data = torch.load(os.path.join(model_dir, filename), map_location="cpu")
    ddp.unwrap(model).load_state_dict(data["model_state_dict"])
    optimizer.load_state_dict(data["optimizer_state_dict"])
    return data["iteration"], model, optimizer


def print_loss(iter_, loss_pack, weights, total):
    line = "iter {:6d} total {:.4f} | ".format(iter_, float(total))
    line += " ".join("{}:{:.4f}".format(k.replace("loss_", ""), float(v.mean()) * weights[k]) for k, v in loss_pack.items())
    print(line, flush=True)


def train(cfg):
    world, rank, local = ddp.init_process_group()
    dev = torch.device("cuda", local) if torch.cuda.is_available() else None
    if dev is None:
        raise RuntimeError("train.py needs a HIP device: the loss stack has no CPU fallback")
    ops.set_align_corners(bool(getattr(cfg, "align_corners", False)))
    model = get_model(cfg.mode)(cfg)
    if cfg.mode == "geom":
        for attr, path in (("flow_pretrained_model", ("fpyramid.", "pwc_model.")),
                           ("depth_pretrained_model", ("depth_net.", "pose_net."))):
            f = getattr(cfg, attr, None)
            if f:
                data = torch.load(f, map_location="cpu")["model_state_dict"]
                data = {k.replace("module.", "", 1): v for k, v in data.items()}
                print("load %s from %s" % (attr, f))
                model.load_state_dict({k: v for k, v in data.items() if k.startswith(path)}, strict=False)
    model = model.to(dev)
    for flag, names in (("fix_depth", ("depth_net",)), ("fix_pose", ("pose_net",)), ("fix_flow", ("fpyramid", "pwc_model"))):
        if getattr(cfg, flag, False):
            for name, p in model.named_parameters():
                if any(n in name for n in names):
                    p.requires_grad = False
    model.train()
    model = ddp.wrap(model, dev)
    use_graph = bool(getattr(cfg, "graph", False)) and world == 1
    optimizer = make_optimizer(model, cfg.lr, capturable=use_graph)
    start = 0
    if cfg.resume:
        fn = "iter_{}.pth".format(cfg.iter_start) if cfg.iter_start > 0 else "last.pth"
        start, model, optimizer = load_model(cfg.model_dir, fn, model, optimizer)
    weights = {k: getattr(cfg, a) for k, a in LOSS_WEIGHT_ATTR.items()}
    h, w = cfg.img_hw
    n_iter = cfg.num_iterations - start
    raw_pipeline = bool(getattr(cfg, "device_pipeline", False))
    if raw_pipeline:   # raw uint8 triplets at KITTI's native size; resize / flip / normalise run on the device
        dataset = synthetic.SyntheticRawTriplets(n_iter * cfg.batch_size * world, (375, 1242), (h, w), cfg.num_scales, seed=1234)
    else:
        dataset = synthetic.SyntheticTriplets(n_iter * cfg.batch_size * world, (h, w), cfg.num_scales, seed=1234)
    prof, graphed = None, None
    if getattr(cfg, "profile", False):       # the reference's Profiler marks (core/visualize/profiler.py) + roctx ranges per HIP launcher
        from unsupervised_depth_opticalflow_egomotion_amd import profiling
        profiling.enable()
        prof = profiling.Profiler(silent=rank != 0)
    t0 = time.time()
    for it in range(start, cfg.num_iterations):
        base = (it - start) * cfg.batch_size * world + rank * cfg.batch_size
        samples = [dataset[base + j] for j in range(cfg.batch_size)]
        if raw_pipeline:
            raw = torch.stack([s[0] for s in samples]).pin_memory().to(dev, non_blocking=True)
            flip = torch.tensor([s[3] for s in samples], dtype=torch.uint8)
            inputs = [ops.prepare_triplets(raw, (h, w), flip)] + \
                [torch.stack([s[i] for s in samples]).to(dev, non_blocking=True) for i in (1, 2)]
        else:
            inputs = [torch.stack([s[i] for s in samples]).to(dev, non_blocking=True) for i in range(3)]
        if prof is not None:
            prof.reset()
        if use_graph:      # one hipGraph launch per iteration (train_step.GraphedTrainStep): captured on the first batch
            if graphed is None:
                graphed = GraphedTrainStep(model, optimizer, inputs, cfg)
            loss, loss_pack, mask_pack = graphed(inputs)
        else:
            loss, loss_pack, mask_pack = train_step(model, optimizer, inputs, cfg, prof if (prof is not None and it % cfg.log_interval == 0) else None)
        if rank == 0 and it % cfg.log_interval == 0:
            print_loss(it, loss_pack, weights, loss)
        if rank == 0 and (it + 1) % cfg.save_interval == 0:
            save_model(it + 1, cfg.model_dir, "iter_{}.pth".format(it + 1), model, optimizer)
            save_model(it + 1, cfg.model_dir, "last.pth", model, optimizer)
    if rank == 0:
        print("done: %d iterations in %.1f s" % (cfg.num_iterations - start, time.time() - t0))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="joint depth / flow / pose training on MI355X")
    ap.add_argument("-c", "--config_file", default="config/kitti_geom.yaml")
    ap.add_argument("-g", "--gpu", type=str, default="0", help="kept for CLI compatibility; use torchrun for >1 GPU")
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--iter_start", type=int, default=0)
    ap.add_argument("--lr", type=float, default=0.0001)
    ap.add_argument("--num_workers", type=int, default=6)
    ap.add_argument("--log_interval", type=int, default=100)
    ap.add_argument("--test_interval", type=int, default=2000)
    ap.add_argument("--save_interval", type=int, default=2000)
    ap.add_argument("--vis_interval", type=int, default=100)
    ap.add_argument("--mode", type=str, default="geom", help="flow | depth | geom")
    ap.add_argument("--model_dir", type=str, default=None)
    ap.add_argument("--prepared_save_dir", type=str, default="data_s1")
    ap.add_argument("--flow_pretrained_model", type=str, default=None)
    ap.add_argument("--depth_pretrained_model", type=str, default=None)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--multi_gpu", action="store_true")
    ap.add_argument("--no_test", action="store_true")
    ap.add_argument("--fix_depth", action="store_true")
    ap.add_argument("--fix_pose", action="store_true")
    ap.add_argument("--fix_flow", action="store_true")
    ap.add_argument("--num_iterations", type=int, default=None)
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole training step (three network streams, loss stack, backward, Adam) in a hipGraph after three "
                         "eager steps and replay it: one launch per iteration (single process, static shapes; pays when the host is the "
                         "bound, e.g. small batches)")
    ap.add_argument("--profile", action="store_true",
                    help="roctx ranges around every HIP launcher (rocprofv3 --marker-trace) and the reference Profiler's forward / backward / "
                         "optimizer wall times at every log interval (synchronises the device at each mark)")
    ap.add_argument("--device_pipeline", action="store_true",
                    help="feed raw uint8 triplets and run resize / flip / normalise on the device (ops.prepare_triplets)")
    args = ap.parse_args()
    with open(args.config_file) as fh:
        cfg = yaml.safe_load(fh)
    cfg["img_hw"] = (cfg["img_hw"][0], cfg["img_hw"][1])
    cfg["model_dir"] = os.path.join(args.model_dir or "./models", args.mode)
    for k, v in vars(args).items():        # every CLI attribute overrides / extends the YAML (train.py:272-274)
        if k in ("model_dir",) or (k == "num_iterations" and v is None):
            continue
        cfg[k] = v
    os.makedirs(cfg["model_dir"], exist_ok=True)
    if int(os.environ.get("RANK", "0")) == 0:
        shutil.copy(args.config_file, cfg["model_dir"])
        with open(os.path.join(cfg["model_dir"], "config.pkl"), "wb") as fh:
            pickle.dump(cfg, fh)
    cfg_new = pObject()
    for k, v in cfg.items():
        setattr(cfg_new, k, v)
    train(cfg_new)
