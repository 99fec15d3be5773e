"""Evaluation metrics of the reference as device-side reductions.

``eval_flow_avg`` (core/evaluation/evaluate_flow.py:85-174: KITTI EPE all / noc / occ / moving / static and the
3 px & 5 % outlier rate) and ``eval_depth`` / ``compute_errors`` (core/evaluation/evaluate_depth.py:13-52,
evaluation_utils.py:11-32: Eigen-crop, median scaling, abs_rel / sq_rel / rmse / rmse_log / a1-a3) with the reference's
signatures, argument meaning and return values.  The reference loops over numpy arrays on the host (cv2.resize per
prediction); here predictions stay tensors on whatever device they arrive on -- the resize is ``F.interpolate`` (bilinear,
half-pixel centres = cv2.INTER_LINEAR) and every sum / mean / median is a device reduction, with one host transfer
of the final scalars.  numpy inputs are accepted and converted.  No HIP kernels: these run a few times per training
run on a few hundred images; the point is not to round-trip predictions through the host."""
import numpy as np
import torch
import torch.nn.functional as F


def _t(a, device=None):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    t = t.float() if t.dtype != torch.float32 and t.dtype != torch.float64 else t
    return t.to(device) if device is not None else t


def resize_flow_like_cv2(flow_hw2, out_hw):
    """cv2.resize(flow, (W, H), interpolation=cv2.INTER_LINEAR) for an [h,w,2] map: bilinear, half-pixel centres,
    edge replication, no antialiasing."""
    t = flow_hw2.permute(2, 0, 1).unsqueeze(0)
    return F.interpolate(t, size=tuple(out_hw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)


def calculate_error_rate(epe_map, gt_flow, mask):
    """evaluate_flow.py:85-90."""
    mag = torch.clamp(torch.sqrt(torch.sum(gt_flow * gt_flow, dim=2)), min=1e-10)
    bad = (epe_map * mask > 3) & (epe_map * mask / mag > 0.05)
    return bad.sum().double() / mask.sum().double()


def eval_flow_avg(gt_flows, noc_masks, pred_flows, cfg, moving_masks=None, write_img=False):
    """evaluate_flow.py:93-174.  Returns the same formatted table string."""
    if write_img:
        raise NotImplementedError("write_img needs cv2 / flow_to_image (visualisation is out of scope)")
    num = len(gt_flows)
    acc = None          # [8] float64 on the predictions' device; ONE host transfer after the loop
    for i, (gt_flow, noc_mask, pred_flow) in enumerate(zip(gt_flows, noc_masks, pred_flows)):
        dev = pred_flow.device if isinstance(pred_flow, torch.Tensor) else None
        gt, noc, pred = _t(gt_flow, dev), _t(noc_mask, dev), _t(pred_flow, dev).clone()
        H, W = gt.shape[0:2]
        pred[:, :, 0] = pred[:, :, 0] / cfg.img_hw[1] * W
        pred[:, :, 1] = pred[:, :, 1] / cfg.img_hw[0] * H
        flo = resize_flow_like_cv2(pred, (H, W))
        valid = gt[:, :, 2]
        epe = torch.sqrt(torch.sum((flo[:, :, 0:2] - gt[:, :, 0:2]) ** 2, dim=2))
        row = [torch.sum(epe * valid).double() / valid.sum().double(),
               torch.sum(epe * noc).double() / noc.sum().double(),
               torch.sum(epe * (valid - noc)).double() / torch.clamp((valid - noc).sum().double(), min=1.0),
               calculate_error_rate(epe, gt[:, :, 0:2], valid)]
        if moving_masks:
            mv = _t(moving_masks[i], dev)
            row += [torch.sum(epe * valid * mv).double() / (valid * mv).sum().double(),
                    torch.sum(epe * valid * (1.0 - mv)).double() / (valid * (1.0 - mv)).sum().double(),
                    calculate_error_rate(epe, gt[:, :, 0:2], valid * mv),
                    calculate_error_rate(epe, gt[:, :, 0:2], valid * (1.0 - mv))]
        else:
            row += [torch.zeros((), dtype=torch.float64, device=epe.device)] * 4
        row = torch.stack([r.double() for r in row])
        acc = row if acc is None else acc + row.to(acc.device)
    error, error_noc, error_occ, error_rate, error_move, error_static, move_rate, static_rate = (acc / num).cpu().tolist()
    if moving_masks:
        result = "{:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10}, {:>10} \n".format(
            "epe", "epe_noc", "epe_occ", "epe_move", "epe_static", "move_err_rate", "static_err_rate", "err_rate")
        result += "{:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f}, {:10.4f} \n".format(
            error, error_noc, error_occ, error_move, error_static, move_rate, static_rate, error_rate)
        return result
    result = "{:>10}, {:>10}, {:>10}, {:>10} \n".format("epe", "epe_noc", "epe_occ", "err_rate")
    result += "{:10.4f}, {:10.4f}, {:10.4f}, {:10.4f} \n".format(error, error_noc, error_occ, error_rate)
    return result


def compute_errors(gt, pred, nyu=False):
    """evaluation_utils.py:11-32 on 1-D tensors."""
    thresh = torch.maximum(gt / pred, pred / gt)
    a1, a2, a3 = (thresh < 1.25).float().mean(), (thresh < 1.25 ** 2).float().mean(), (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    log10 = torch.mean(torch.abs(torch.log10(gt) - torch.log10(pred)))
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean(((gt - pred) ** 2) / gt)
    return (abs_rel, sq_rel, rmse, log10 if nyu else rmse_log, a1, a2, a3)


def _median_np(x):
    """numpy.median semantics (mean of the two middle values for even counts); torch.median returns the lower one."""
    n = x.numel()
    s = torch.sort(x.reshape(-1))[0]
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def eval_depth(gt_depths, pred_depths, min_depth=1e-3, max_depth=80, nyu=False):
    """evaluate_depth.py:13-52: Eigen crop, median scaling, clamping, seven metrics averaged over the samples."""
    rows = []
    for gt_depth, pred_depth in zip(gt_depths, pred_depths):
        dev = pred_depth.device if isinstance(pred_depth, torch.Tensor) else None
        gt, pred = _t(gt_depth, dev), _t(pred_depth, dev)
        mask = (gt > min_depth) & (gt < max_depth)
        if not nyu:
            h, w = gt.shape
            c = np.array([0.40810811 * h, 0.99189189 * h, 0.03594771 * w, 0.96405229 * w]).astype(np.int32)
            crop = torch.zeros_like(mask)
            crop[c[0]:c[1], c[2]:c[3]] = True
            mask = mask & crop
        g, p = gt[mask], pred[mask]
        p = p * (_median_np(g) / _median_np(p))
        p = torch.clamp(p, min_depth, max_depth)
        g = torch.clamp(g, min_depth, max_depth)
        rows.append(torch.stack([v.float() for v in compute_errors(g, p, nyu=nyu)]))
    dev0 = rows[0].device
    return [float(v) for v in torch.stack([r.to(dev0) for r in rows]).float().mean(0).cpu()]
