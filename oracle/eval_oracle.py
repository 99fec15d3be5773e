"""ORACLE -- test infrastructure only.  numpy restatement of the reference's evaluation metrics
(core/evaluation/evaluate_flow.py:85-174, evaluate_depth.py:13-52, evaluation_utils.py:11-32).  The reference calls
cv2.resize(..., INTER_LINEAR), which is absent here: ``resize_linear`` states its published definition (half-pixel
centres, edge replication, no antialiasing).  Parity unpinned by reference outputs for the resize (cv2 cannot be
imported); the metric arithmetic is a line-by-line restatement."""
import numpy as np


def resize_linear(img, out_hw):
    h, w = img.shape[:2]
    H, W = out_hw
    ys = np.clip((np.arange(H) + 0.5) * (h / H) - 0.5, 0, None)
    xs = np.clip((np.arange(W) + 0.5) * (w / W) - 0.5, 0, None)
    y0 = np.minimum(np.floor(ys).astype(int), h - 1); x0 = np.minimum(np.floor(xs).astype(int), w - 1)
    y1 = np.minimum(y0 + 1, h - 1); x1 = np.minimum(x0 + 1, w - 1)
    wy = (ys - y0)[:, None, None]; wx = (xs - x0)[None, :, None]
    im = img.astype(np.float64)
    top = im[y0][:, x0] * (1 - wx) + im[y0][:, x1] * wx
    bot = im[y1][:, x0] * (1 - wx) + im[y1][:, x1] * wx
    return (top * (1 - wy) + bot * wy).astype(np.float32)


def calculate_error_rate(epe_map, gt_flow, mask):
    bad = np.logical_and(epe_map * mask > 3,
                         epe_map * mask / np.maximum(np.sqrt(np.sum(np.square(gt_flow), axis=2)), 1e-10) > 0.05)
    return bad.sum() / mask.sum()


def eval_flow_avg(gt_flows, noc_masks, pred_flows, img_hw, moving_masks=None):
    acc = np.zeros(8)
    for i, (gt_flow, noc_mask, pred_flow) in enumerate(zip(gt_flows, noc_masks, pred_flows)):
        H, W = gt_flow.shape[0:2]
        pred_flow = np.copy(pred_flow)
        pred_flow[:, :, 0] = pred_flow[:, :, 0] / img_hw[1] * W
        pred_flow[:, :, 1] = pred_flow[:, :, 1] / img_hw[0] * H
        flo = resize_linear(pred_flow, (H, W))
        epe = np.sqrt(np.sum(np.square(flo[:, :, 0:2] - gt_flow[:, :, 0:2]), axis=2))
        v = gt_flow[:, :, 2]
        acc[0] += np.sum(epe * v) / np.sum(v)
        acc[1] += np.sum(epe * noc_mask) / np.sum(noc_mask)
        acc[2] += np.sum(epe * (v - noc_mask)) / max(np.sum(v - noc_mask), 1.0)
        acc[3] += calculate_error_rate(epe, gt_flow[:, :, 0:2], v)
        if moving_masks:
            mv = moving_masks[i]
            acc[4] += np.sum(epe * v * mv) / np.sum(v * mv)
            acc[5] += np.sum(epe * v * (1.0 - mv)) / np.sum(v * (1.0 - mv))
            acc[6] += calculate_error_rate(epe, gt_flow[:, :, 0:2], v * mv)
            acc[7] += calculate_error_rate(epe, gt_flow[:, :, 0:2], v * (1.0 - mv))
    return acc / len(gt_flows)


def compute_errors(gt, pred):
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    return np.mean(np.abs(gt - pred) / gt), np.mean(((gt - pred) ** 2) / gt), rmse, rmse_log, a1, a2, a3


def eval_depth(gt_depths, pred_depths, min_depth=1e-3, max_depth=80):
    rows = []
    for gt_depth, pred_depth in zip(gt_depths, pred_depths):
        mask = np.logical_and(gt_depth > min_depth, gt_depth < max_depth)
        h, w = gt_depth.shape
        c = np.array([0.40810811 * h, 0.99189189 * h, 0.03594771 * w, 0.96405229 * w]).astype(np.int32)
        crop = np.zeros(mask.shape); crop[c[0]:c[1], c[2]:c[3]] = 1
        mask = np.logical_and(mask, crop)
        g, p = gt_depth[mask].copy(), pred_depth[mask].copy()
        p *= np.median(g) / np.median(p)
        p = np.clip(p, min_depth, max_depth); g = np.clip(g, min_depth, max_depth)
        rows.append(compute_errors(g, p))
    return np.array(rows, np.float32).mean(0)
