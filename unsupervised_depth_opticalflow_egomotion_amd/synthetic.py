"""Deterministic synthetic KITTI-shaped triplets and loss-stack inputs.

There is no KITTI data on the build or GPU boxes, so every config of
BASELINE.json is driven by these generators.  Everything is produced with
``numpy.random.Generator(PCG64(seed))`` (bit-stable across numpy versions and
machines) so that the golden-vector script (run next to the reference), the
tests and ``bench.py`` all see identical inputs without shipping them.

Tuple contract of a sample follows the reference data loader
(``core/dataset/kitti_prepared.py:132-152``): ``(images[3,3H,W] in [0,1] with the
frames stacked left/target/right along H, K_ms[S,3,3], K_inv_ms[S,3,3])``.

The loss-stack generator follows SURVEY.md Appendix A.6: flows are the rigid flow
of (disp, pose) plus noise plus a few "moving object" rectangles and the
neighbouring frames are the target frame resampled along those flows, so that the
occlusion / texture / dynamic masks are a non-trivial mix of zeros and ones.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List

import numpy as np


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(int(seed)))


def _box3(a: np.ndarray) -> np.ndarray:
    """3x3 box filter with edge replication on the last two axes."""
    p = np.pad(a, [(0, 0)] * (a.ndim - 2) + [(1, 1), (1, 1)], mode="edge")
    h, w = a.shape[-2:]
    out = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            out += p[..., dy:dy + h, dx:dx + w]
    return out / 9.0


def smooth_texture(rng: np.random.Generator, shape, passes: int = 2) -> np.ndarray:
    """Low-pass filtered uniform noise stretched back to [0,1] (float32)."""
    a = rng.random(shape, dtype=np.float64)
    for _ in range(passes):
        a = _box3(a)
    lo = a.min(axis=(-2, -1), keepdims=True)
    hi = a.max(axis=(-2, -1), keepdims=True)
    a = (a - lo) / np.maximum(hi - lo, 1e-12)
    return a.astype(np.float32)


def kitti_like_intrinsics(h: int, w: int) -> np.ndarray:
    """K = [[0.58W,0,0.5W],[0,1.92H,0.5H],[0,0,1]]."""
    return np.array([[0.58 * w, 0.0, 0.5 * w],
                     [0.0, 1.92 * h, 0.5 * h],
                     [0.0, 0.0, 1.0]], dtype=np.float64)


def multiscale_intrinsics(h: int, w: int, num_scales: int):
    """K_ms[s] = K with rows 0-1 divided by 2**s, K_inv_ms = inverse (float32)."""
    k = kitti_like_intrinsics(h, w)
    ks, kis = [], []
    for s in range(num_scales):
        ksc = k.copy()
        ksc[0:2] /= float(2 ** s)
        ks.append(ksc)
        kis.append(np.linalg.inv(ksc))
    return np.stack(ks).astype(np.float32), np.stack(kis).astype(np.float32)


def scale_hw(h: int, w: int, s: int):
    return int(h / (2 ** s)), int(w / (2 ** s))


def _euler_rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    xm = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ym = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    zm = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return xm @ ym @ zm


def rigid_flow_np(depth: np.ndarray, pose6: np.ndarray, k: np.ndarray) -> np.ndarray:
    """float64 rigid flow of one sample: depth [H,W], pose (tx,ty,tz,rx,ry,rz), K 3x3 -> [2,H,W]."""
    h, w = depth.shape
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    pix = np.stack([xs, ys, np.ones_like(xs)], 0).reshape(3, -1)
    cam = (np.linalg.inv(k) @ pix) * depth.reshape(1, -1)
    r = _euler_rot(*pose6[3:])
    p = (k @ r) @ cam + (k @ pose6[:3].reshape(3, 1))
    z = np.maximum(p[2], 1e-3)
    return np.stack([(p[0] / z).reshape(h, w) - xs, (p[1] / z).reshape(h, w) - ys], 0)


def _bilinear_np(img: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Clamped bilinear lookup img[C,H,W] at float coords -> [C,H',W'] (float64)."""
    c, h, w = img.shape
    x = np.clip(x, 0.0, w - 1.0)
    y = np.clip(y, 0.0, h - 1.0)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1 = np.minimum(x0 + 1, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    fx = x - x0
    fy = y - y0
    return (img[:, y0, x0] * (1 - fx) * (1 - fy) + img[:, y0, x1] * fx * (1 - fy)
            + img[:, y1, x0] * (1 - fx) * fy + img[:, y1, x1] * fx * fy)


def robust_pose(pose: np.ndarray, max_steps: int = 400) -> np.ndarray:
    """Nudge every rotation angle (last three entries of each 6-vector) by a few float32 ulps until the exact
    values of cos and sin are at most 0.3 ulp from their nearest float32.

    The reference evaluates ``torch.cos/sin`` through MKL VML (HA mode, <= 0.6 ulp); the kernels round a double
    evaluation (<= 0.5 ulp).  Two implementations with those error bounds must both return the nearest float
    when the exact value is less than 0.4 ulp from it, so for poses treated this way the rotation matrices -- and
    with them every projected coordinate and every mask fed by it -- are decided on identical bits.  This is
    test-input conditioning in the sense of SURVEY.md A.5 (decision margins), not a change of the model."""
    out = np.array(pose, dtype=np.float32, copy=True)
    flat = out.reshape(-1, 6)

    def safe(x32):
        for fn in (np.cos, np.sin):
            v = fn(np.float64(x32))
            r = np.float32(v)
            ulp = np.float64(np.spacing(np.abs(r))) if r != 0 else np.float64(np.finfo(np.float32).tiny)
            if abs(v - np.float64(r)) / ulp >= 0.3:
                return False
        return True

    for row in flat:
        for k in (3, 4, 5):
            x0 = np.float64(row[k])
            # cos moves by |sin x| dx: step so that it advances ~0.07 ulp of cos per unit (never less than 1 ulp of
            # x; below 1e-4 rad cos is 1 - O(1e-9) and safe as it stands).  Triangular multiples of the step break
            # the aliasing of "k ulps of x = almost an integer number of ulps of sin x".
            step = np.float64(np.spacing(np.abs(np.float32(x0)))) if x0 != 0 else 1e-12
            if abs(x0) >= 1e-4:
                step = max(step, 0.07 * np.float64(np.spacing(np.float32(np.abs(np.cos(x0))))) / abs(np.sin(x0)))
            x = np.float32(x0)
            for i in range(max_steps):
                x = np.float32(x0 + (i * (i + 1) // 2) * step)
                if safe(x):
                    break
            else:
                raise RuntimeError("robust_pose: no safe angle near %r" % float(x0))
            row[k] = x
    return out


@dataclass
class LossStackInputs:
    """Everything the loss stack consumes for one batch (numpy, float32).

    ``imgs``: (left, target, right) each [B,3,H,W]; ``disps``: 3 lists (left,
    target, right) of S arrays [B,1,Hs,Ws]; ``pose`` [B,2,6] (index 0 = target->left
    "bwd", 1 = target->right "fwd", reference model_geometry.py:789-790);
    ``flows_bwd/fwd``: lists of ``num_flow_scales`` arrays [B,2,Hs,Ws];
    ``K``/``K_inv`` [B,3,3] (scale-0 intrinsics)."""
    imgs: List[np.ndarray]
    disps: List[List[np.ndarray]]
    pose: np.ndarray
    flows_bwd: List[np.ndarray]
    flows_fwd: List[np.ndarray]
    K: np.ndarray
    K_inv: np.ndarray
    num_scales: int = 3
    meta: dict = field(default_factory=dict)


# Pose conditioning (robust_pose) is a TEST-INPUT device: the parity suite and the golden generator switch it on
# (tests/conftest.py, tests/golden/make_golden.py) so that masks can be compared for equality with a reference whose
# cos / sin come from a vendor libm; the library's own generator (bench.py, smoke()) draws raw poses.
CONDITION_POSE = False


def make_loss_stack_inputs(batch: int, h: int, w: int, num_scales: int = 3, seed: int = 1234,
                           num_flow_scales: int | None = None, pose_sigma: float = 0.02,
                           flow_noise: float = 0.3, condition_pose: bool | None = None) -> LossStackInputs:
    """Synthetic net outputs + frames for the loss stack.  ``condition_pose``: nudge the rotation angles
    with ``robust_pose`` (None = the module default ``CONDITION_POSE``, off unless a test harness enabled it)."""
    if condition_pose is None:
        condition_pose = CONDITION_POSE
    rng = _rng(seed)
    if num_flow_scales is None:
        num_flow_scales = num_scales + 1
    k64 = kitti_like_intrinsics(h, w)
    img_t = smooth_texture(rng, (batch, 3, h, w))
    # disparities: sigmoid of smooth noise, one per frame and scale
    disps = []
    for _frame in range(3):
        lst = []
        for s in range(num_scales):
            hs, ws = scale_hw(h, w, s)
            z = (smooth_texture(rng, (batch, 1, hs, ws), passes=3).astype(np.float64) - 0.5) * 4.0
            lst.append((1.0 / (1.0 + np.exp(-z))).astype(np.float32))
        disps.append(lst)
    pose = (pose_sigma * rng.standard_normal((batch, 2, 6))).astype(np.float32)
    pose[:, :, 3:] *= 0.25  # rotations smaller than translations
    if condition_pose:
        pose = robust_pose(pose)  # cos / sin unambiguous for every <= 0.6-ulp implementation (see robust_pose)
    flows = [[], []]
    for d in range(2):
        for s in range(num_flow_scales):
            hs, ws = scale_hw(h, w, s)
            ks = k64.copy()
            ks[0:2] /= (h / hs)
            fl = np.zeros((batch, 2, hs, ws), np.float64)
            for b in range(batch):
                dsrc = disps[1][min(s, num_scales - 1)][b, 0].astype(np.float64)
                if dsrc.shape != (hs, ws):  # extra (dropped) 1/8 flow scale: subsample
                    dsrc = dsrc[:: dsrc.shape[0] // hs, :: dsrc.shape[1] // ws][:hs, :ws]
                fl[b] = rigid_flow_np(dsrc, pose[b, d].astype(np.float64), ks)
            fl += flow_noise / (2 ** s) * rng.standard_normal(fl.shape)
            # moving-object rectangles
            for b in range(batch):
                for _ in range(2):
                    y0 = int(rng.integers(0, max(hs - hs // 4, 1)))
                    x0 = int(rng.integers(0, max(ws - ws // 4, 1)))
                    fl[b, :, y0:y0 + hs // 4, x0:x0 + ws // 6] += (5.0 / (2 ** s)) * (1 if d else -1)
            flows[d].append(fl.astype(np.float32))
    # neighbour frames: target resampled against the full-res flows (+ photometric noise)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    nbrs = []
    for d in range(2):
        out = np.zeros((batch, 3, h, w), np.float64)
        for b in range(batch):
            f = flows[d][0][b].astype(np.float64)
            out[b] = _bilinear_np(img_t[b].astype(np.float64), xs - f[0], ys - f[1])
        out += 0.02 * rng.standard_normal(out.shape)
        nbrs.append(np.clip(out, 0.0, 1.0).astype(np.float32))
    kb = np.broadcast_to(k64.astype(np.float32), (batch, 3, 3)).copy()
    kib = np.broadcast_to(np.linalg.inv(k64).astype(np.float32), (batch, 3, 3)).copy()
    return LossStackInputs(imgs=[nbrs[0], img_t, nbrs[1]], disps=disps, pose=pose,
                           flows_bwd=flows[0], flows_fwd=flows[1], K=kb, K_inv=kib,
                           num_scales=num_scales, meta=dict(batch=batch, h=h, w=w, seed=seed))


def make_triplet_batch(batch: int, h: int, w: int, num_scales: int = 3, seed: int = 1234):
    """(images[B,3,3H,W], K_ms[B,S,3,3], K_inv_ms[B,S,3,3]) float32 numpy."""
    rng = _rng(seed)
    base = smooth_texture(rng, (batch, 3, h, w + 16))
    frames = [base[..., 0:w], base[..., 8:8 + w], base[..., 16:16 + w]]  # a panning camera
    frames = [np.clip(f + 0.01 * rng.standard_normal(f.shape).astype(np.float32), 0, 1) for f in frames]
    images = np.concatenate(frames, axis=2).astype(np.float32)
    k_ms, k_inv_ms = multiscale_intrinsics(h, w, num_scales)
    k_ms = np.broadcast_to(k_ms, (batch,) + k_ms.shape).copy()
    k_inv_ms = np.broadcast_to(k_inv_ms, (batch,) + k_inv_ms.shape).copy()
    return images, k_ms, k_inv_ms


class SyntheticTriplets:
    """Map-style dataset with the reference's sample tuple contract."""

    def __init__(self, num_samples: int, img_hw=(256, 832), num_scales: int = 3, seed: int = 1234):
        self.n, self.hw, self.s, self.seed = int(num_samples), tuple(img_hw), int(num_scales), int(seed)

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        import torch
        im, k, ki = make_triplet_batch(1, self.hw[0], self.hw[1], self.s, seed=self.seed + int(idx))
        return torch.from_numpy(im[0]), torch.from_numpy(k[0]), torch.from_numpy(ki[0])


class SyntheticRawTriplets:
    """Raw counterpart of ``SyntheticTriplets`` for the device-side input pipeline (ops.prepare_triplets): what
    ``cv2.imread`` hands ``KITTI_Prepared.__getitem__`` (kitti_prepared.py:143) -- a uint8 [3*H0, W0, 3] stacked
    triplet at the dataset's native size -- plus the multiscale intrinsics for the training size and the flip draw.
    ``ds[i] -> (raw_u8, K_ms, K_inv_ms, flip)``."""

    def __init__(self, num_samples: int, raw_hw=(375, 1242), img_hw=(256, 832), num_scales: int = 3, seed: int = 1234):
        self.n, self.raw_hw, self.hw, self.s, self.seed = int(num_samples), tuple(raw_hw), tuple(img_hw), int(num_scales), int(seed)

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        import torch
        h0, w0 = self.raw_hw
        im, _, _ = make_triplet_batch(1, h0, w0, 1, seed=self.seed + int(idx))
        raw = np.clip(np.rint(im[0] * 255.0), 0, 255).astype(np.uint8).transpose(1, 2, 0)      # [3*H0, W0, 3]
        k_ms, k_inv_ms = multiscale_intrinsics(self.hw[0], self.hw[1], self.s)
        flip = int(_rng(self.seed * 7919 + int(idx)).random() > 0.5)
        return torch.from_numpy(np.ascontiguousarray(raw)), torch.from_numpy(k_ms), torch.from_numpy(k_inv_ms), flip
