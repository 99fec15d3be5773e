#!/usr/bin/env python3
"""Evaluation driver with the reference's CLI surface (test.py:314-377) on the MI355X engine.

  python test.py -c config/kitti_geom.yaml --mode geom --task demo [--pretrained_model last.pth] [--result_dir out]
  python test.py -c config/kitti_geom.yaml --mode geom --task kitti_flow_2015 --pretrained_model last.pth

Same flags and the same model construction / ``load_state_dict(weights['model_state_dict'], strict=False)`` /
``eval()`` sequence as the reference (test.py:347-360).  Tasks:

* ``kitti_flow_2012`` / ``kitti_flow_2015`` (test.py:21-87): image pairs + 16-bit flow ground truth of ``cfg.gt_2012_dir`` /
  ``cfg.gt_2015_dir`` -> ``inference_flow`` -> ``eval_flow_avg`` (with the object-map moving masks for 2015);
* ``kitti_depth`` (test.py:104-135): the Eigen split listed in ``<eigen_dir>/test_files.txt`` over ``cfg.raw_base_dir``,
  ground truth ``<eigen_dir>/gt_depths.npz`` (``eigen_dir`` defaults to ./data/eigen) -> ``infer_depth`` -> ``eval_depth``;
* ``kitti_pose`` (test.py:137-194): 3-frame snippets of ``cfg.kitti_odom_dir`` / ``cfg.sequences`` -> ``infer_pose`` ->
  ATE / RE;
* ``demo``: with ``--image_path`` the single-image depth demo (test.py:269-283); without it, the three inference entry
  points on a synthetic KITTI-shaped triplet scored against synthetic ground truth (a plumbing check needing no data).

Files are read by ``kitti_io`` (PIL for 8-bit images in cv2's BGR order, an own PNG codec for the 16-bit flow maps, a
half-pixel bilinear resize for ``cv2.resize``); predictions go to ``--result_dir`` when given (flow: KITTI 16-bit PNGs at
the ground-truth size, depth / demo: ``.npy`` + 16-bit PNG).  None of the datasets exists on the build or GPU machines:
a task whose directory is missing fails with a clear error, and the file paths are exercised by the test suite on
miniature KITTI-shaped trees.  Fixes to the shipped reference: ``test_kitti_2012`` is called with its four parameters
(test.py:372 passes five)."""
import argparse
import os

import numpy as np
import torch
import yaml

from core.evaluation import eval_depth, eval_flow_avg
from core.networks import Model_depth, Model_flow, Model_geometry
from unsupervised_depth_opticalflow_egomotion_amd import kitti_io, ops, synthetic
from unsupervised_depth_opticalflow_egomotion_amd.structures import pose_vec2mat


class pObject(object):
    pass


def require_dir(cfg, key):
    path = getattr(cfg, key, None)
    if not path or not os.path.isdir(path):
        raise FileNotFoundError("task needs the KITTI data directory cfg.%s (got %r): not present on this machine" % (key, path))
    return path


def test_synthetic(cfg, model, dev, num=2):
    """infer_depth / infer_pose / inference_flow on synthetic triplets + the metric tables (no dataset needed)."""
    h, w = cfg.img_hw
    gt_flows, nocs, preds, gt_depths, pred_depths = [], [], [], [], []
    for i in range(num):
        images, _, _ = synthetic.make_triplet_batch(1, h, w, cfg.num_scales, seed=4321 + i)
        images = torch.from_numpy(images).to(dev)
        img_l, img, img_r = images[:, :, :h], images[:, :, h:2 * h], images[:, :, 2 * h:]
        with torch.no_grad():
            if hasattr(model, "pwc_model"):
                flow = model.inference_flow(img, img_r)                       # [1,2,h,w]
                preds.append(flow[0].permute(1, 2, 0).contiguous())
                gt = np.zeros((375, 1242, 3), np.float32); gt[:, :, 0] = 8.0 * 1242 / w; gt[:, :, 2] = 1.0   # the panning camera
                gt_flows.append(gt); nocs.append(gt[:, :, 2].copy())
            if hasattr(model, "depth_net"):
                depth = model.infer_depth(img)                                # [1,1,h,w]
                pred_depths.append(torch.nn.functional.interpolate(depth, (375, 1242), mode="bilinear", align_corners=False)[0, 0])
                gt_depths.append(np.full((375, 1242), 10.0, np.float32))
                pose = model.infer_pose(torch.cat([img_l, img, img_r], 1))
                assert tuple(pose.shape) == (1, 2, 6)
    if preds:
        print("[EVAL] [synthetic flow]")
        print(eval_flow_avg(gt_flows, nocs, preds, cfg))
    if pred_depths:
        res = eval_depth(gt_depths, pred_depths)
        print("[EVAL] [synthetic depth]")
        print("{:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}".format("abs_rel", "sq_rel", "rms", "log_rms", "a1", "a2", "a3"))
        print("{:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}".format(*res))


def _count(dirname, suffix):
    return len([f for f in os.listdir(dirname) if f.endswith(suffix)]) if os.path.isdir(dirname) else 0


def _result_dir(cfg, sub):
    if not getattr(cfg, "result_dir", None):
        return None
    d = os.path.join(cfg.result_dir, sub)
    os.makedirs(d, exist_ok=True)
    return d


def test_kitti_flow(cfg, model, dev, year):
    """test.py:21-87 (test_kitti_2012 / test_kitti_2015)."""
    gt_dir = require_dir(cfg, "gt_%d_dir" % year)
    num = _count(os.path.join(gt_dir, "flow_occ"), "_10.png")
    if num == 0:
        raise FileNotFoundError("no flow_occ/*_10.png under %s" % gt_dir)
    gt_flows, noc_masks = kitti_io.load_gt_flow_kitti(gt_dir, "kitti_%d" % year, num)
    gt_masks = kitti_io.load_gt_mask(gt_dir, num) if year == 2015 and os.path.isdir(os.path.join(gt_dir, "obj_map")) else None
    dataset = kitti_io.KITTIFlowPairs(gt_dir, cfg.img_hw, num, year)
    out_dir = _result_dir(cfg, "flow_%d" % year)
    preds = []
    for i in range(len(dataset)):
        img, _, _ = dataset[i]
        img = img[None].to(dev)
        h = img.shape[2] // 2
        with torch.no_grad():
            flow = model.inference_flow(img[:, :, :h].contiguous(), img[:, :, h:].contiguous())
        pred = flow[0].permute(1, 2, 0).contiguous()          # [h, w, 2] on the device: eval_flow_avg keeps it there
        preds.append(pred)
        if out_dir:
            H, W = gt_flows[i].shape[:2]
            full = pred.clone()
            full[:, :, 0] *= W / cfg.img_hw[1]
            full[:, :, 1] *= H / cfg.img_hw[0]
            from core.evaluation import resize_flow_like_cv2
            kitti_io.write_flow_png(os.path.join(out_dir, "%06d_10.png" % i), resize_flow_like_cv2(full, (H, W)).cpu().numpy())
    res = eval_flow_avg(gt_flows, noc_masks, preds, cfg, moving_masks=gt_masks, write_img=False)
    print("CONFIG: {0}, mode: {1}".format(cfg.config_file, cfg.mode))
    print("[EVAL] [KITTI %d]" % year)
    print(res)
    return res


def test_eigen_depth(cfg, model, dev):
    """test.py:104-135: Eigen split, cv2-style resize to the training size, 1 / (disp + 1e-4) at the ground-truth size."""
    raw = require_dir(cfg, "raw_base_dir")
    eigen = getattr(cfg, "eigen_dir", None) or "./data/eigen"
    listing, gt_file = os.path.join(eigen, "test_files.txt"), os.path.join(eigen, "gt_depths.npz")
    for f in (listing, gt_file):
        if not os.path.exists(f):
            raise FileNotFoundError("kitti_depth needs %s (Eigen split files of the reference's ./data/eigen)" % f)
    names = [ln.strip().split(" ") for ln in open(listing) if ln.strip()]
    gt_depths = np.load(gt_file, allow_pickle=True)["data"]
    out_dir = _result_dir(cfg, "depth")
    pred_depths = []
    for i, (path1, idx, _) in enumerate(names):
        img = kitti_io.read_image_bgr(os.path.join(raw, path1, "image_02/data/" + str(idx) + ".png"))
        x = torch.from_numpy(kitti_io.resize_bilinear_u8(img, cfg.img_hw) / 255.0).float().to(dev).unsqueeze(0).permute(0, 3, 1, 2).contiguous()
        with torch.no_grad():
            disp = model.infer_depth(x)                        # the reference names infer_depth's output "disp" (test.py:117)
        h, w = gt_depths[i].shape
        disp = torch.nn.functional.interpolate(disp, (h, w), mode="bilinear", align_corners=False)[0, 0]
        pred_depths.append(1.0 / (disp + 1e-4))
        if out_dir:
            np.save(os.path.join(out_dir, "%04d.npy" % i), pred_depths[-1].cpu().numpy())
    res = eval_depth(list(gt_depths), pred_depths)
    print("{:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10} ".format("abs_rel", "sq_rel", "rms", "log_rms", "a1", "a2", "a3"))
    print("{:10.4f}, {:10.4f}, {:10.3f}, {:10.3f}, {:10.3f}, {:10.3f}, {:10.3f} ".format(*res))
    return res


def test_pose_odom(cfg, model, dev):
    """test.py:137-194: ATE / RE over 3-frame odometry snippets."""
    root = require_dir(cfg, "kitti_odom_dir")
    dataset = kitti_io.KITTIPoseSnippets(root, getattr(cfg, "sequences", ["09", "10"]), 3)
    print("{} snippets to test".format(len(dataset)))
    errors = np.zeros((len(dataset), 2), np.float32)
    for j in range(len(dataset)):
        sample = dataset[j]
        frames = [kitti_io.resize_bilinear_u8(im, cfg.img_hw) for im in sample["imgs"]]
        x = torch.from_numpy(np.concatenate(frames, 2) / 255.0).float().to(dev).unsqueeze(0).permute(0, 3, 1, 2).contiguous()
        with torch.no_grad():
            poses = model.infer_pose(x)[0]
        poses = torch.cat([poses[0].view(-1, 6), torch.zeros(1, 6, device=poses.device), poses[1].view(-1, 6)])
        inv_t = pose_vec2mat(poses).cpu().numpy().astype(np.float64)
        rot = np.linalg.inv(inv_t[:, :, :3])
        tr = -rot @ inv_t[:, :, -1:]
        tm = np.concatenate([rot, tr], axis=-1)
        first = inv_t[0]
        final = first[:, :3] @ tm
        final[:, :, -1:] += first[:, -1:]
        errors[j] = kitti_io.compute_pose_error(sample["poses"], final)
    mean, std = errors.mean(0), errors.std(0)
    print("Results")
    print("\t {:>10}, {:>10}".format("ATE", "RE"))
    print("mean \t {:10.4f}, {:10.4f}".format(*mean))
    print("std \t {:10.4f}, {:10.4f}".format(*std))
    return mean, std


def test_single_image(img_path, model, dev, training_hw, save_dir):
    """test.py:269-283: depth of one image file; saves the disparity at the input size (.npy + 16-bit PNG)."""
    img = kitti_io.read_image_bgr(img_path)
    h, w = img.shape[:2]
    x = torch.from_numpy(kitti_io.resize_bilinear_u8(img, training_hw).transpose(2, 0, 1)).float().to(dev).unsqueeze(0) / 255.0
    with torch.no_grad():
        disp = model.infer_depth(x.contiguous())
    disp = torch.nn.functional.interpolate(disp, (h, w), mode="bilinear", align_corners=False)[0, 0].cpu().numpy()
    save_dir = save_dir or "./"
    os.makedirs(save_dir, exist_ok=True)
    np.save(os.path.join(save_dir, "demo_disp.npy"), disp)
    lo, hi = float(disp.min()), float(disp.max())
    kitti_io.write_png(os.path.join(save_dir, "demo_disp.png"), ((disp - lo) / max(hi - lo, 1e-12) * 65535.0).astype(np.uint16))
    print("Depth prediction saved in " + save_dir)


def main():
    ap = argparse.ArgumentParser(description="TrianFlow testing.")
    ap.add_argument("-c", "--config_file", default=None, help="config file.")
    ap.add_argument("-g", "--gpu", type=str, default=0, help="gpu id.")
    ap.add_argument("--mode", type=str, default="depth", help="mode for testing.")
    ap.add_argument("--task", type=str, default="kitti_depth",
                    help="To test on which task, kitti_depth or kitti_flow_2012/2015 or kitti_pose or demo")
    ap.add_argument("--image_path", type=str, default=None, help="Set this only when task==demo. Depth demo for single image.")
    ap.add_argument("--pretrained_model", type=str, default=None, help="directory for loading pretrained models")
    ap.add_argument("--result_dir", type=str, default=None, help="directory for saving predictions")
    args = ap.parse_args()
    if not args.config_file or not os.path.exists(args.config_file):
        raise ValueError("config file not found.")
    with open(args.config_file, "r") as f:
        cfg = yaml.safe_load(f)
    cfg["img_hw"] = (cfg["img_hw"][0], cfg["img_hw"][1])
    cfg["model_dir"] = args.result_dir
    for attr in dir(args):
        if attr[:2] != "__":
            cfg[attr] = getattr(args, attr)
    cfg_new = pObject()
    for attr in list(cfg.keys()):
        setattr(cfg_new, attr, cfg[attr])

    if args.mode == "flow":
        model = Model_flow(cfg_new)
    elif args.mode == "depth":
        model = Model_depth(cfg_new)
    elif args.mode == "geom" or args.task == "demo":
        model = Model_geometry(cfg_new)
    else:
        raise ValueError("mode must be flow, depth or geom")
    if not torch.cuda.is_available():
        raise RuntimeError("test.py needs a HIP device: the loss-stack operators have no CPU fallback")
    dev = torch.device("cuda", int(args.gpu) if str(args.gpu).isdigit() else 0)
    ops.set_align_corners(bool(getattr(cfg_new, "align_corners", False)))
    model.to(dev)
    if args.pretrained_model:
        weights = torch.load(args.pretrained_model, map_location="cpu")
        state = {k.replace("module.", "", 1): v for k, v in weights["model_state_dict"].items()}
        model.load_state_dict(state, strict=False)
    model.eval()
    print("Model Loaded.")

    if args.task == "demo":
        if args.image_path is not None:
            test_single_image(args.image_path, model, dev, cfg["img_hw"], args.result_dir)
        else:
            test_synthetic(cfg_new, model, dev)
    elif args.task == "kitti_depth":
        test_eigen_depth(cfg_new, model, dev)
    elif args.task == "kitti_flow_2015":
        test_kitti_flow(cfg_new, model, dev, 2015)
    elif args.task == "kitti_flow_2012":
        test_kitti_flow(cfg_new, model, dev, 2012)
    elif args.task == "kitti_pose":
        test_pose_odom(cfg_new, model, dev)
    else:
        raise ValueError("unknown task %r" % args.task)


if __name__ == "__main__":
    main()
