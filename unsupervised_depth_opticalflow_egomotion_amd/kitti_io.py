"""KITTI file formats for ``test.py`` (reference: core/evaluation/flowlib.py:107-127, evaluate_flow.py:16-83,
evaluate_mask.py:199-213, core/dataset/kitti_2012.py / kitti_2015.py, test.py:104-194) without cv2 / pypng / imageio,
none of which exist on this machine.

* 8-bit images go through PIL; KITTI's flow ground truth is a 16-bit RGB PNG, which PIL truncates to 8 bits, so this
  module carries its own PNG codec (``read_png`` / ``write_png``: 8/16-bit grey, grey+alpha, RGB, RGBA, non-interlaced).
  The un-filtering of a PNG is sequential along a row and down the rows; it is vectorised here over anti-diagonals
  (every pixel with row + column = k depends only on pixels of the two previous diagonals), ~0.2 s per KITTI frame in
  numpy.
* ``cv2.imread`` returns BGR and the reference feeds that order to its networks (checkpoints are trained on it):
  ``read_image_bgr`` keeps the order.  ``cv2.resize(INTER_LINEAR)`` becomes a bilinear resize with half-pixel centres
  (``resize_bilinear_u8``): cv2's 8-bit path rounds its weights to 11 bits, so a resized pixel can differ by 1/255.

Host-side data plumbing around the hot path -- out of SURVEY.md section 8's scope, kept small, here so that the
reference's evaluation commands run when the datasets are present."""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _unfilter(raw: np.ndarray, h: int, stride: int, bpp: int) -> np.ndarray:
    """PNG row filters 0-4 undone for all rows at once.  raw: uint8 [h, 1 + stride].  Vectorised over anti-diagonals
    of (row, pixel): recon(r, p) needs recon(r, p-1), recon(r-1, p), recon(r-1, p-1)."""
    ftype = raw[:, 0].astype(np.int64)
    npx = stride // bpp
    data = raw[:, 1:].reshape(h, npx, bpp).astype(np.int32)
    rec = np.zeros((h + 1, npx + 1, bpp), np.int32)       # row 0 / column 0 = the zero border
    if np.all(ftype == 0):
        return data.astype(np.uint8).reshape(h, stride)
    if np.all((ftype == 0) | (ftype == 2)):               # only None / Up: a running sum down the rows
        out = data.copy()
        for r in range(1, h):
            if ftype[r] == 2:
                out[r] = (out[r] + out[r - 1]) & 255
        return out.astype(np.uint8).reshape(h, stride)
    rows_all = np.arange(h)
    for k in range(h + npx - 1):
        r0, r1 = max(0, k - npx + 1), min(h - 1, k)
        rows = rows_all[r0:r1 + 1]
        cols = k - rows
        x = data[rows, cols]
        a = rec[rows + 1, cols]          # left
        b = rec[rows, cols + 1]          # up
        c = rec[rows, cols]              # up-left
        ft = ftype[rows][:, None]
        p = a + b - c
        pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
        paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        pred = np.where(ft == 1, a, np.where(ft == 2, b, np.where(ft == 3, (a + b) >> 1, np.where(ft == 4, paeth, 0))))
        rec[rows + 1, cols + 1] = (x + pred) & 255
    return rec[1:, 1:].astype(np.uint8).reshape(h, stride)


def read_png(path: str) -> np.ndarray:
    """[H, W] or [H, W, C] array, uint8 or uint16 (big-endian samples converted), of a non-interlaced PNG."""
    with open(path, "rb") as fh:
        blob = fh.read()
    if blob[:8] != _PNG_SIG:
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, hdr, palette = 8, [], None, None
    while pos < len(blob):
        n, kind = struct.unpack(">I4s", blob[pos:pos + 8])
        body = blob[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace or depth not in (8, 16) or (ctype == 3 and depth != 8):
        raise ValueError("%s: unsupported PNG (bit depth %d, colour type %d, interlace %d)" % (path, depth, ctype, interlace))
    ch = 1 if ctype == 3 else _CHANNELS[ctype]
    bpp = ch * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + stride)
    px = _unfilter(raw, h, stride, bpp)
    if depth == 16:
        out = px.reshape(h, w, ch, 2).astype(np.uint16)
        out = (out[..., 0] << 8) | out[..., 1]
    else:
        out = px.reshape(h, w, ch)
    if ctype == 3:
        out = palette[out[..., 0]]
    return out[..., 0] if out.shape[-1] == 1 else out


def write_png(path: str, img: np.ndarray) -> None:
    """uint8 / uint16 [H,W] or [H,W,C] (C = 1..4) -> PNG (filter 0 rows)."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, ch = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    if a.dtype == np.uint16:
        depth, rows = 16, a.astype(">u2").tobytes()
    elif a.dtype == np.uint8:
        depth, rows = 8, a.tobytes()
    else:
        raise ValueError("write_png takes uint8 or uint16")
    stride = w * ch * depth // 8
    raw = b"".join(b"\x00" + rows[r * stride:(r + 1) * stride] for r in range(h))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    with open(path, "wb") as fh:
        fh.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ------------------------------------------------------------------------------------------------ flow ground truth
def read_flow_png(path: str) -> np.ndarray:
    """flowlib.py:107-127: [H,W,3] float64 = (u, v, valid) with u, v = (raw - 2^15) / 64 and 0 where invalid."""
    raw = read_png(path).astype(np.float64)
    flow = np.zeros(raw.shape[:2] + (3,), np.float64)
    flow[:, :, 2] = raw[:, :, 2]
    flow[:, :, 0:2] = (raw[:, :, 0:2] - 2 ** 15) / 64.0
    flow[raw[:, :, 2] == 0, 0:2] = 0
    return flow


def write_flow_png(path: str, flow_hw2: np.ndarray, valid: np.ndarray | None = None) -> None:
    """flowlib.py:130-139 (KITTI submission format)."""
    h, w = flow_hw2.shape[:2]
    out = np.ones((h, w, 3), np.float64)
    out[:, :, 0:2] = np.clip(np.asarray(flow_hw2, np.float64)[:, :, 0:2] * 64.0 + 2 ** 15, 0, 2 ** 16 - 1)
    if valid is not None:
        out[:, :, 2] = valid
    write_png(path, out.astype(np.uint16))


def load_gt_flow_kitti(gt_dataset_dir: str, mode: str, num: int | None = None):
    """evaluate_flow.py:54-83: (flow_occ maps, noc masks) of KITTI 2012 (194 pairs) / 2015 (200 pairs)."""
    if mode not in ("kitti_2012", "kitti_2015"):
        raise ValueError("Mode {} not found.".format(mode))
    n = num if num is not None else (194 if mode == "kitti_2012" else 200)
    flows, nocs = [], []
    for i in range(n):
        name = str(i).zfill(6) + "_10.png"
        flows.append(read_flow_png(os.path.join(gt_dataset_dir, "flow_occ", name)))
        nocs.append(read_flow_png(os.path.join(gt_dataset_dir, "flow_noc", name))[:, :, 2])
    return flows, nocs


def load_gt_mask(gt_dataset_dir: str, num: int = 200):
    """evaluate_mask.py:193-213: KITTI 2015 object maps as {0,1} moving masks."""
    masks = []
    for i in range(num):
        m = read_png(os.path.join(gt_dataset_dir, "obj_map", str(i).zfill(6) + "_10.png")).astype(np.float64)
        if m.ndim == 3:
            m = m[:, :, 0]
        m[m > 0.0] = 1.0
        masks.append(m)
    return masks


# ------------------------------------------------------------------------------------------------ images / calibration
def read_image_bgr(path: str) -> np.ndarray:
    """cv2.imread(path): uint8 [H,W,3] in B, G, R order."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def read_image_rgb(path: str) -> np.ndarray:
    """imageio.imread(path): uint8 [H,W,3] in R, G, B order -- what the reference's odometry loader feeds infer_pose
    (core/dataset/kitti_pose.py:6,18), unlike its cv2-based flow / depth loaders."""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB")))


def resize_bilinear_u8(img: np.ndarray, out_hw) -> np.ndarray:
    """cv2.resize(img, (W, H), INTER_LINEAR) for uint8 / float [h,w,C]: half-pixel centres, edge replication, no
    antialiasing; float32 result (not rounded back to uint8)."""
    h, w = img.shape[:2]
    H, W = int(out_hw[0]), int(out_hw[1])
    ys = np.clip((np.arange(H) + 0.5) * (h / H) - 0.5, 0, h - 1)
    xs = np.clip((np.arange(W) + 0.5) * (w / W) - 0.5, 0, w - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    wy, wx = (ys - y0).astype(np.float32), (xs - x0).astype(np.float32)
    im = img.astype(np.float32)
    if im.ndim == 2:
        im = im[:, :, None]
    top = im[y0][:, x0] * (1 - wx)[None, :, None] + im[y0][:, x1] * wx[None, :, None]
    bot = im[y1][:, x0] * (1 - wx)[None, :, None] + im[y1][:, x1] * wx[None, :, None]
    out = top * (1 - wy)[:, None, None] + bot * wy[:, None, None]
    return out if img.ndim == 3 else out[:, :, 0]


def read_raw_calib_file(path: str) -> dict:
    """evaluate_flow.py:31-45 (pykitti): 'key: v0 v1 ...' lines -> {key: float array}; non-numeric values skipped."""
    data = {}
    with open(path) as fh:
        for line in fh:
            if ":" not in line:
                continue
            key, value = line.split(":", 1)
            try:
                data[key] = np.array([float(x) for x in value.split()])
            except ValueError:
                pass
    return data


def load_intrinsics_raw(calib_file: str) -> np.ndarray:
    """evaluate_flow.py:20-29: the 3x3 of P_rect_02 (raw / 2015 calib files) or P2 (odometry / 2012)."""
    d = read_raw_calib_file(calib_file)
    p = d["P_rect_02"] if "P_rect_02" in d else d["P2"]
    return np.reshape(p, (3, 4))[:3, :3].copy()


def rescale_intrinsics(K: np.ndarray, hw_orig, hw_new) -> np.ndarray:
    """kitti_prepared.py rescale_intrinsics: focal lengths / principal point scaled with the image."""
    out = np.array(K, np.float64, copy=True)
    out[0, :] *= hw_new[1] / hw_orig[1]
    out[1, :] *= hw_new[0] / hw_orig[0]
    return out


class KITTIFlowPairs:
    """core/dataset/kitti_2012.py / kitti_2015.py: pair i = image_2/{i:06d}_10.png, _11.png + calib_cam_to_cam/{i:06d}.txt.
    ``ds[i] -> (img [3, 2H, W] float in [0,1] (the two frames stacked along H, BGR), K [3,3], K_inv [3,3])``."""

    def __init__(self, data_dir: str, img_hw=(256, 832), num: int | None = None, year: int = 2015):
        self.data_dir, self.img_hw = data_dir, tuple(img_hw)
        self.n = num if num is not None else (194 if year == 2012 else 200)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        import torch
        if i >= self.n:
            raise IndexError(i)
        stem = str(i).zfill(6)
        im1 = read_image_bgr(os.path.join(self.data_dir, "image_2", stem + "_10.png"))
        im2 = read_image_bgr(os.path.join(self.data_dir, "image_2", stem + "_11.png"))
        hw0 = im1.shape[:2]
        both = np.concatenate([resize_bilinear_u8(im1, self.img_hw), resize_bilinear_u8(im2, self.img_hw)], 0) / 255.0
        K = load_intrinsics_raw(os.path.join(self.data_dir, "calib_cam_to_cam", stem + ".txt"))
        K[0, 1] = K[1, 0] = K[2, 0] = K[2, 1] = 0.0
        K = rescale_intrinsics(K, hw0, self.img_hw)
        return (torch.from_numpy(np.ascontiguousarray(both.transpose(2, 0, 1))).float(), torch.from_numpy(K).float(),
                torch.from_numpy(np.linalg.inv(K)).float())


# ------------------------------------------------------------------------------------------------ odometry snippets
def read_odometry_poses(path: str) -> np.ndarray:
    """KITTI odometry ground truth: one 3x4 row-major matrix per line -> [N,3,4] float64."""
    return np.loadtxt(path, dtype=np.float64).reshape(-1, 3, 4)


class KITTIPoseSnippets:
    """core/dataset (KITTI_pose, test.py:137-139): 3-frame snippets of the odometry sequences with their ground-truth
    poses expressed relative to the first frame of the snippet.  ``ds[j] -> {'imgs': [3 x uint8 HxWx3 RGB], 'poses': [3,3,4]}``
    -- RGB: the reference reads these frames with imageio (kitti_pose.py:18), not with cv2."""

    def __init__(self, root: str, sequences, seq_length: int = 3):
        self.samples = []
        k = (seq_length - 1) // 2
        for seq in sequences:
            seq = "%02d" % int(seq)
            img_dir = os.path.join(root, "sequences", seq, "image_2")
            poses = read_odometry_poses(os.path.join(root, "poses", seq + ".txt"))
            files = sorted(f for f in os.listdir(img_dir) if f.endswith(".png"))
            for i in range(k, min(len(files), len(poses)) - k):
                idx = list(range(i - k, i + k + 1))
                self.samples.append(([os.path.join(img_dir, files[q]) for q in idx], poses[idx]))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, j):
        paths, poses = self.samples[j]
        first = np.vstack([poses[0], [0, 0, 0, 1]])
        inv = np.linalg.inv(first)
        rel = np.stack([(inv @ np.vstack([p, [0, 0, 0, 1]]))[:3] for p in poses])
        return {"imgs": [read_image_rgb(p) for p in paths], "poses": rel}


def compute_pose_error(gt: np.ndarray, pred: np.ndarray):
    """test.py:196-212: scale-aligned ATE and mean rotation error of one snippet ([N,3,4] each)."""
    n = gt.shape[0]
    scale = np.sum(gt[:, :, -1] * pred[:, :, -1]) / np.sum(pred[:, :, -1] ** 2)
    ate = np.linalg.norm((gt[:, :, -1] - scale * pred[:, :, -1]).reshape(-1))
    re = 0.0
    for g, p in zip(gt, pred):
        R = g[:, :3] @ np.linalg.inv(p[:, :3])
        s = np.linalg.norm([R[0, 1] - R[1, 0], R[1, 2] - R[2, 1], R[0, 2] - R[2, 0]])
        re += np.arctan2(s, np.trace(R) - 1)
    return ate / n, re / n
