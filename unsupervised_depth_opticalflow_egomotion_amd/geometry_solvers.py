"""Batched geometric solvers behind the reference's zero-weighted placeholders ``loss_triangle`` / ``loss_eight_point``
(core/networks/model_geometry.py:427-470, 532-683), on the device, as a mixin of the models.

* match sampling (``top_ratio_sample``, ``robust_rand_sample``, ``sample_match``, model_geometry.py:427-470) -- torch
  top-k / gathers on the device; the random pick uses the global torch RNG exactly as the reference does (one
  ``torch.randint`` over the kept matches), so it follows the same seed.
* mid-point triangulation and its depth-registration loss (``midpoint_triangulate`` ... ``compute_triangulate_loss``,
  model_geometry.py:569-683): closed-form, batched over B x N matches; pinned by golden G11 (the reference's own methods
  called unbound, like G5).
* fundamental matrix from matches (``compute_fundmental_mat`` / ``compute_eight_point_loss``, model_geometry.py:532-566):
  the reference calls ``cv2.findFundamentalMat(FM_RANSAC, 0.1, 0.99)`` (``FM_LMEDS, 0.99`` for nyuv2) on the host, one
  sample at a time.  Here (round 4) the same robust estimate as ONE batched computation on the device: ``ransac_iters``
  minimal sets of 8 matches per sample drawn from a seeded generator, one normalised eight-point solve per set (batched
  SVDs), OpenCV's residual -- the larger of the two squared point-to-epipolar-line distances -- for all matches x all
  hypotheses at once, consensus by inlier count at the reference's threshold (RANSAC) or by the median residual with
  OpenCV's robust sigma (LMedS), then Hartley's normalised eight-point refit on the consensus set, rank 2 enforced,
  scaled like OpenCV's result (F[2,2] = 1).  cv2 does not exist here and its RANSAC draw cannot be pinned: property
  tests (exact recovery from noise-free matches, recovery within 1e-2 with 30 % gross outliers, rank 2).
* perspective-n-point (``pnp`` / ``compute_pnp_loss``, model_geometry.py:473-530; the reference:
  ``cv2.solvePnPRansac(confidence=0.9999, reprojectionError=1)`` + ``cv2.solvePnP(ITERATIVE, useExtrinsicGuess)`` per
  sample on the host): batched RANSAC -- six-point DLT hypotheses, reprojection residuals of all correspondences, the
  1-pixel consensus set -- then a Levenberg-Marquardt refinement on SE(3) over that set (float64 inside), same
  (T, axis-angle) output convention; property tests (30 % gross outliers, pose within 1e-2).

None of this is reached by ``Model_geometry.forward`` (the reference keeps the calls commented, :939-951); the methods
exist so that those lines can be switched on as written."""
import torch
import torch.nn.functional as F

from .structures import compute_essential_matrix, compute_projection_matrix


class GeometrySolvers:
    ratio = 0.3
    num = 6000
    dataset = "kitti_depth"
    ransac_iters = 256          # hypotheses per sample (cv2's adaptive count is at most ~300 at its defaults for >= 50 % inliers)
    ransac_seed = 0             # the draw is a function of (seed, batch shape): reproducible, unlike cv2's

    # ------------------------------------------------------------------ match sampling (model_geometry.py:427-470)
    def top_ratio_sample(self, match, depth, mask, ratio):
        """match [b,4,n], depth [b,1,n], mask [b,1,n] (scores) -> the int(ratio * n) best-scored matches."""
        n = match.shape[-1]
        scores, idx = torch.topk(mask, int(ratio * n), dim=-1)
        sel = idx.squeeze(1)
        return (torch.gather(match, 2, sel.unsqueeze(1).expand(-1, 4, -1)),
                torch.gather(depth, 2, sel.unsqueeze(1)), scores)

    def robust_rand_sample(self, match, depth, mask, num):
        """``num`` uniformly drawn matches; if some scores are zero, draw among the non-zero ones of every sample."""
        b, n = match.shape[0], match.shape[2]
        nonzero = int(torch.sum(mask > 0, dim=-1).min())
        if nonzero == n:
            pick = torch.randint(0, n, [num]).to(match.device)
            return match[:, :, pick], depth[:, :, pick], num
        num = min(nonzero, num)
        rows = []
        for i in range(b):
            nz = torch.nonzero(mask[i, 0, :]).squeeze(1)
            rows.append(nz[torch.randint(0, nz.shape[0], [int(num)]).to(nz.device)])
        sel = torch.stack(rows, 0)
        return (torch.gather(match, 2, sel.unsqueeze(1).expand(-1, 4, -1)), torch.gather(depth, 2, sel.unsqueeze(1)), num)

    def sample_match(self, flow, depth, mask):
        """(x, y, x + u, y + v) of the top-``ratio`` scored pixels, then ``num`` random ones -> ([b,4,m], [b,1,m])."""
        b, _, h, w = flow.shape
        grid = self.meshgrid(b, h, w).to(flow.device)
        match = torch.cat([grid, grid + flow], 1).view(b, 4, -1)
        m, d, s = self.top_ratio_sample(match, depth.reshape(b, 1, -1), mask.reshape(b, 1, -1), self.ratio)
        m, d, _ = self.robust_rand_sample(m, d, s, self.num)
        return m, d

    # ------------------------------------------------------------------ triangulation (model_geometry.py:569-683)
    def midpoint_triangulate(self, match, K, K_inv, P1, P2):
        """Mid-point of the common perpendicular of the two viewing rays of every match -> homogeneous points [b,n,4]."""
        b, n = match.shape[0], match.shape[2]
        ones = torch.ones(b, 1, n, device=match.device, dtype=match.dtype)
        rays, origins = [], []
        for P, xy in ((P1, match[:, :2]), (P2, match[:, 2:])):
            RT = K_inv.bmm(P)
            Rt = RT[:, :, :3].transpose(1, 2)
            d = Rt.bmm(K_inv).bmm(torch.cat([xy, ones], 1))
            rays.append(d / (torch.norm(d, dim=1, keepdim=True, p=2) + 1e-12))
            origins.append(-1 * Rt.bmm(RT[:, :, 3].unsqueeze(-1)))
        cr = torch.cross(rays[0], rays[1], dim=1)
        denom = 1.0 / (torch.sum(cr * cr, dim=1, keepdim=True) + 1e-12)
        base = (origins[1] - origins[0]).repeat(1, 1, n)
        a1 = torch.sum(torch.cross(base, rays[1], dim=1) * cr, dim=1, keepdim=True) * denom
        a2 = torch.sum(torch.cross(base, rays[0], dim=1) * cr, dim=1, keepdim=True) * denom
        point = ((origins[0] + a1 * rays[0]) + (origins[1] + a2 * rays[1])) / 2.0
        return torch.cat([point, ones], dim=1).transpose(1, 2)

    def reproject(self, P, point3d):
        """P [b,3,4], points [b,n,4] -> pixel coordinates [b,n,2] and depths [b,n,1]."""
        p = P.bmm(point3d.transpose(1, 2))
        z = p[:, 2, :].unsqueeze(1)
        return (p[:, :2, :] / (z + 1e-12)).transpose(1, 2), z.transpose(1, 2)

    def scale_adapt(self, depth1, depth2, eps=1e-12):
        with torch.no_grad():
            A = torch.sum((depth1 ** 2) / (depth2 ** 2 + eps), dim=1)
            C = torch.sum(depth1 / (depth2 + eps), dim=1)
            return C / (A + eps)

    def affine_adapt(self, depth1, depth2, use_translation=True, eps=1e-12):
        a_scale = self.scale_adapt(depth1, depth2, eps=eps)
        if not use_translation:
            return a_scale, torch.zeros_like(a_scale)
        with torch.no_grad():
            A = torch.sum((depth1 ** 2) / (depth2 ** 2 + eps), dim=1)
            B = torch.sum(depth1 / (depth2 ** 2 + eps), dim=1)
            C = torch.sum(depth1 / (depth2 + eps), dim=1)
            D = torch.sum(1.0 / (depth2 ** 2 + eps), dim=1)
            E = torch.sum(1.0 / (depth2 + eps), dim=1)
            det = B * B - A * D
            a = (B * E - D * C) / (det + 1e-12)
            b = (B * C - A * E) / (det + 1e-12)
            ok = (torch.abs(det) > 1e-4).float()
            return a * ok + a_scale * (1 - ok), b * ok

    def register_depth(self, depth_pred, coord_tri, depth_tri):
        """Sample the predicted depth at the reprojected points (reflection padding), normalise by the median ratio to
        the triangulated depths and fit the remaining scale (model_geometry.py:633-651)."""
        bsz, _, h, w = depth_pred.shape
        n = depth_tri.shape[1]
        grid = torch.stack([2.0 * coord_tri[:, :, 0] / (w - 1.0) - 1.0, 2.0 * coord_tri[:, :, 1] / (h - 1.0) - 1.0], -1)
        from . import ops                   # the reference leaves align_corners to the installed torch (ops.set_align_corners)
        inter = F.grid_sample(depth_pred, grid.view(bsz, n, 1, 2), padding_mode="reflection",
                              align_corners=ops.get_align_corners()).squeeze(-1).transpose(1, 2)
        scale = (torch.median(inter, 1)[0] / (torch.median(depth_tri, 1)[0] + 1e-12)).detach()
        inter_s = inter / (scale.unsqueeze(-1) + 1e-12)
        pred_s = depth_pred / (scale.unsqueeze(-1).unsqueeze(-1) + 1e-12)
        a, b = self.affine_adapt(inter_s, depth_tri, use_translation=False)
        return (a.unsqueeze(-1).unsqueeze(-1) * pred_s + b.unsqueeze(-1).unsqueeze(-1), a.unsqueeze(1) * inter_s + b.unsqueeze(1))

    def get_trian_loss(self, tri_depth, pred_tri_depth):
        return torch.pow(1.0 - pred_tri_depth / (tri_depth + 1e-12), 2).mean((1, 2))

    def compute_triangulate_loss(self, match, pose, K, K_inv, depth_pred1, depth_pred2):
        d1, d2 = depth_pred1[0], depth_pred2[0]
        P1, P2 = compute_projection_matrix(pose, K)
        pts = self.midpoint_triangulate(match, K, K_inv, P1, P2)
        c1, z1 = self.reproject(P1, pts)
        c2, z2 = self.reproject(P2, pts)
        _, i1 = self.register_depth(d1, c1, z1)
        _, i2 = self.register_depth(d2, c2, z2)
        return self.get_trian_loss(z1, i1) + self.get_trian_loss(z2, i2)

    # ------------------------------------------------------------------ eight-point (model_geometry.py:532-566)
    @staticmethod
    def _hartley(xy):
        """Similarity that moves the centroid of [b,2,n] points to the origin and their mean distance to sqrt(2)."""
        c = xy.mean(2, keepdim=True)
        d = torch.sqrt(((xy - c) ** 2).sum(1)).mean(1)
        s = (2.0 ** 0.5) / (d + 1e-12)
        T = torch.zeros(xy.shape[0], 3, 3, device=xy.device, dtype=xy.dtype)
        T[:, 0, 0] = s
        T[:, 1, 1] = s
        T[:, 0, 2] = -s * c[:, 0, 0]
        T[:, 1, 2] = -s * c[:, 1, 0]
        T[:, 2, 2] = 1.0
        return T

    @staticmethod
    def _weighted_hartley(xy, w):
        """_hartley over the matches with weight 1 (w [b,n] in {0,1}; at least one per sample)."""
        cnt = w.sum(1).clamp_min(1.0)
        c = (xy * w.unsqueeze(1)).sum(2, keepdim=True) / cnt.view(-1, 1, 1)
        d = (torch.sqrt(((xy - c) ** 2).sum(1)) * w).sum(1) / cnt
        s = (2.0 ** 0.5) / (d + 1e-12)
        T = torch.zeros(xy.shape[0], 3, 3, device=xy.device, dtype=xy.dtype)
        T[:, 0, 0] = s
        T[:, 1, 1] = s
        T[:, 0, 2] = -s * c[:, 0, 0]
        T[:, 1, 2] = -s * c[:, 1, 0]
        T[:, 2, 2] = 1.0
        return T

    @classmethod
    def _eight_point(cls, m, w=None):
        """Normalised eight-point over the matches m [b,4,n] (float64) with 0/1 weights w [b,n]: least squares, rank 2
        enforced, un-normalised; not yet scaled."""
        b, _, n = m.shape
        if w is None:
            w = torch.ones(b, n, device=m.device, dtype=m.dtype)
        ones = torch.ones(b, 1, n, device=m.device, dtype=m.dtype)
        T1, T2 = cls._weighted_hartley(m[:, :2], w), cls._weighted_hartley(m[:, 2:], w)
        p1 = T1.bmm(torch.cat([m[:, :2], ones], 1))
        p2 = T2.bmm(torch.cat([m[:, 2:], ones], 1))
        A = (p2.unsqueeze(2) * p1.unsqueeze(1)).reshape(b, 9, n)        # row (i, j) = p2_i * p1_j
        _, vecs = torch.linalg.eigh((A * w.unsqueeze(1)).bmm(A.transpose(1, 2)))   # ascending: column 0 = the null direction
        Fn = vecs[:, :, 0].reshape(b, 3, 3)
        U, S, Vh = torch.linalg.svd(Fn)
        S = S.clone()
        S[:, 2] = 0.0
        Fn = U.bmm(torch.diag_embed(S)).bmm(Vh)
        return T2.transpose(1, 2).bmm(Fn).bmm(T1)

    @staticmethod
    def _epipolar_residual(Fm, m):
        """OpenCV's residual of findFundamentalMat (modules/calib3d/src/fundam.cpp computeError): the larger of the squared
        distances of x2 to the line F x1 and of x1 to the line F^T x2.  Fm [b,h,3,3], m [b,4,n] -> [b,h,n]."""
        b, _, n = m.shape
        ones = torch.ones(b, 1, n, device=m.device, dtype=m.dtype)
        x1 = torch.cat([m[:, :2], ones], 1).unsqueeze(1)                # [b,1,3,n]
        x2 = torch.cat([m[:, 2:], ones], 1).unsqueeze(1)
        l2 = Fm.matmul(x1)                                              # lines in image 2  [b,h,3,n]
        l1 = Fm.transpose(2, 3).matmul(x2)                              # lines in image 1
        num = (x2 * l2).sum(2)                                          # x2^T F x1        [b,h,n]
        d2 = num * num / (l2[:, :, 0] ** 2 + l2[:, :, 1] ** 2 + 1e-300)
        d1 = num * num / (l1[:, :, 0] ** 2 + l1[:, :, 1] ** 2 + 1e-300)
        return torch.maximum(d1, d2)

    def _minimal_sets(self, b, n, k, device):
        """ransac_iters index sets of k matches per sample [b,h,k] from a generator seeded with ransac_seed (a set may
        repeat an index: such a hypothesis is degenerate and simply loses the vote)."""
        g = torch.Generator(device="cpu")
        g.manual_seed(int(self.ransac_seed) + 1000003 * n + 7919 * k)
        return torch.randint(0, n, (b, int(self.ransac_iters), k), generator=g).to(device)

    def compute_fundmental_mat(self, matches, pose_vec=None, intrinsics=None, intrinsics_inverse=None, robust=True):
        """F [b,3,3] with x2^T F x1 = 0 for matches [b,4,n] = (x1, y1, x2, y2), as cv2.findFundamentalMat returns it
        (F[2,2] = 1): RANSAC at 0.1 px (LMedS for dataset 'nyuv2'), then the normalised eight-point refit on the
        consensus set (model_geometry.py:532-543).  ``robust=False``: plain least squares over all matches.  float64
        inside (the 9x9 normal matrix squares the condition number)."""
        m = matches.detach().double()
        b, _, n = m.shape
        if n < 8:
            raise ValueError("the eight-point algorithm needs at least 8 matches")
        w = None
        if robust and n > 8:
            idx = self._minimal_sets(b, n, 8, m.device)                              # [b,h,8]
            h = idx.shape[1]
            sets = torch.gather(m.unsqueeze(1).expand(b, h, 4, n), 3, idx.unsqueeze(2).expand(b, h, 4, 8))
            Fh = self._eight_point(sets.reshape(b * h, 4, 8)).reshape(b, h, 3, 3)
            err = self._epipolar_residual(Fh, m)                                     # [b,h,n]
            err = torch.where(torch.isfinite(err), err, torch.full_like(err, float("inf")))
            if self.dataset == "nyuv2":            # cv2.FM_LMEDS: least median of the residuals, then OpenCV's robust sigma
                med = err.median(dim=2)[0]                                           # [b,h]
                best = med.argmin(1)
                e = torch.gather(err, 1, best.view(b, 1, 1).expand(b, 1, n)).squeeze(1)
                sigma = 2.5 * 1.4826 * (1.0 + 5.0 / max(n - 8, 1)) * torch.sqrt(torch.gather(med, 1, best.view(b, 1)).clamp_min(0.0))
                w = (e <= (sigma * sigma).clamp_min(1e-12)).double()
            else:                                  # cv2.FM_RANSAC, ransacReprojThreshold = 0.1
                inl = err <= 0.1 * 0.1
                best = inl.sum(2).argmax(1)
                w = torch.gather(inl, 1, best.view(b, 1, 1).expand(b, 1, n)).squeeze(1).double()
            few = w.sum(1) < 8                                                       # no consensus: fall back to all matches
            w = torch.where(few.unsqueeze(1), torch.ones_like(w), w)
        Fm = self._eight_point(m, w)
        if w is not None:      # one more pass: the consensus set of the refit (cv2 refits on its inliers once; this is that set)
            e = self._epipolar_residual(Fm.unsqueeze(1), m).squeeze(1)
            thr = 0.1 * 0.1 if self.dataset != "nyuv2" else None
            if thr is not None:
                w2 = (e <= thr).double()
                w2 = torch.where((w2.sum(1) < 8).unsqueeze(1), w, w2)
                Fm = self._eight_point(m, w2)
        Fm = Fm / Fm[:, 2:3, 2:3]
        return Fm.to(matches.dtype)

    def compute_eight_point_loss(self, matches, pose_vec, intrinsics, intrinsics_inverse):
        """smooth-L1 between the fundamental matrix of the matches and the one the pose predicts,
        K^-T ([t]x R) K^-1 (model_geometry.py:545-566)."""
        target = self.compute_fundmental_mat(matches, pose_vec, intrinsics, intrinsics_inverse)
        E = compute_essential_matrix(pose_vec)
        F_pred = torch.inverse(intrinsics.permute([0, 2, 1])).bmm(E.bmm(intrinsics_inverse))
        return F.smooth_l1_loss(F_pred, target)

    # ------------------------------------------------------------------ perspective-n-point (model_geometry.py:473-530)
    @staticmethod
    def _so3_exp(w):
        """Rodrigues: axis-angle [b,3] -> rotation matrices [b,3,3] (series near zero)."""
        th = torch.sqrt((w * w).sum(1, keepdim=True) + 1e-30).unsqueeze(-1)
        z = torch.zeros_like(w[:, 0])
        Kx = torch.stack([z, -w[:, 2], w[:, 1], w[:, 2], z, -w[:, 0], -w[:, 1], w[:, 0], z], 1).view(-1, 3, 3)
        a = torch.where(th > 1e-6, torch.sin(th) / th, 1.0 - th * th / 6.0)
        b = torch.where(th > 1e-6, (1.0 - torch.cos(th)) / (th * th), 0.5 - th * th / 24.0)
        return torch.eye(3, dtype=w.dtype, device=w.device).unsqueeze(0) + a * Kx + b * Kx.bmm(Kx)

    @staticmethod
    def _so3_log(R):
        """Rotation matrices [b,3,3] -> axis-angle [b,3] (angles below pi)."""
        c = ((R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2] - 1.0) / 2.0).clamp(-1.0, 1.0)
        th = torch.acos(c)
        v = torch.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1) / 2.0
        k = torch.where(th > 1e-6, th / torch.sin(th).clamp_min(1e-30), 1.0 + th * th / 6.0)
        return v * k.unsqueeze(1)

    def _pnp_hypotheses(self, x, X, Kd):
        """Six-point DLT poses for ransac_iters minimal sets per sample: x [b,n,2] pixels, X [b,n,3] points, float64 ->
        R [b,h,3,3], T [b,h,3] (R projected onto SO(3), the scale taken from its singular values, the sign from det)."""
        b, n = X.shape[0], X.shape[1]
        idx = self._minimal_sets(b, n, 6, X.device)                                   # [b,h,6]
        h = idx.shape[1]
        Xs = torch.gather(X.unsqueeze(1).expand(b, h, n, 3), 2, idx.unsqueeze(3).expand(b, h, 6, 3))
        xs = torch.gather(x.unsqueeze(1).expand(b, h, n, 2), 2, idx.unsqueeze(3).expand(b, h, 6, 2))
        Kinv = torch.inverse(Kd)
        ones = torch.ones(b, h, 6, 1, dtype=X.dtype, device=X.device)
        xn = torch.cat([xs, ones], 3).matmul(Kinv.t())                                # normalised image coordinates
        Xh = torch.cat([Xs, ones], 3)                                                 # [b,h,6,4]
        zero = torch.zeros_like(Xh)
        A = torch.cat([torch.cat([Xh, zero, -xn[..., 0:1] * Xh], 3), torch.cat([zero, Xh, -xn[..., 1:2] * Xh], 3)], 2)   # [b,h,12,12]
        _, _, Vh = torch.linalg.svd(A)
        P = Vh[..., -1, :].reshape(b, h, 3, 4)
        U, S, Wh = torch.linalg.svd(P[..., :3])
        det = torch.det(U.matmul(Wh))
        sign = torch.where(det < 0, -torch.ones_like(det), torch.ones_like(det))
        R = U.matmul(Wh) * sign.view(b, h, 1, 1)
        T = P[..., 3] * (sign / S.mean(-1).clamp_min(1e-300)).unsqueeze(-1)
        return R, T

    def pnp(self, pts2d, pts3d, K, ini_pose=None, iterations=20, robust=True):
        """Pose [b,6] = (T, axis-angle) minimising the reprojection error of pts3d [b,n,3] onto pts2d [b,n,2] under the
        single camera matrix K [3,3] -- the output convention of the reference's ``pnp`` (model_geometry.py:473-495), which
        runs cv2.solvePnPRansac(confidence=0.9999, reprojectionError=1) + an iterative refinement per sample on the host.
        Here, batched on the device in float64: RANSAC (six-point DLT hypotheses, 1-pixel consensus) unless ``ini_pose``
        is given (axis-angle first, then T, as the reference reads it) or ``robust=False``, then Levenberg-Marquardt on
        SE(3) over the consensus set (the reference refines over all points from the RANSAC pose; with gross outliers
        that undoes the RANSAC, so the refinement here is weighted by the consensus and re-scored once)."""
        X = pts3d.detach().double()
        x = pts2d.detach().double()
        Kd = K.detach().double()
        b, n = X.shape[0], X.shape[1]
        fx, fy, cx, cy = Kd[0, 0], Kd[1, 1], Kd[0, 2], Kd[1, 2]
        wgt = torch.ones(b, n, dtype=X.dtype, device=X.device)
        use_ransac = robust and ini_pose is None and n > 6
        if use_ransac:
            Rh, Th = self._pnp_hypotheses(x, X, Kd)                                   # [b,h,3,3], [b,h,3]
            Yh = X.unsqueeze(1).matmul(Rh.transpose(2, 3)) + Th.unsqueeze(2)          # [b,h,n,3]
            zh = Yh[..., 2]
            eu = fx * Yh[..., 0] / zh.clamp_min(1e-9) + cx - x[:, :, 0].unsqueeze(1)
            ev = fy * Yh[..., 1] / zh.clamp_min(1e-9) + cy - x[:, :, 1].unsqueeze(1)
            inl = ((eu * eu + ev * ev) <= 1.0) & (zh > 0)                            # reprojectionError = 1 pixel
            best = inl.sum(2).argmax(1)
            R = torch.gather(Rh, 1, best.view(b, 1, 1, 1).expand(b, 1, 3, 3)).squeeze(1)
            T = torch.gather(Th, 1, best.view(b, 1, 1).expand(b, 1, 3)).squeeze(1)
            wgt = torch.gather(inl, 1, best.view(b, 1, 1).expand(b, 1, n)).squeeze(1).double()
            wgt = torch.where((wgt.sum(1) < 6).unsqueeze(1), torch.ones_like(wgt), wgt)
        elif ini_pose is None:
            R = torch.eye(3, dtype=X.dtype, device=X.device).unsqueeze(0).repeat(b, 1, 1)
            T = torch.zeros(b, 3, dtype=X.dtype, device=X.device)
        else:
            R, T = self._so3_exp(ini_pose[:, 0:3].detach().double()), ini_pose[:, 3:6].detach().double().clone()
        R, T = self._pnp_refine(X, x, (fx, fy, cx, cy), R, T, wgt, iterations)
        if use_ransac:       # the consensus set of the refined pose, and one more refinement on it
            Y = X.bmm(R.transpose(1, 2)) + T.unsqueeze(1)
            z = Y[:, :, 2]
            eu = fx * Y[:, :, 0] / z.clamp_min(1e-9) + cx - x[:, :, 0]
            ev = fy * Y[:, :, 1] / z.clamp_min(1e-9) + cy - x[:, :, 1]
            w2 = (((eu * eu + ev * ev) <= 1.0) & (z > 0)).double()
            w2 = torch.where((w2.sum(1) < 6).unsqueeze(1), wgt, w2)
            R, T = self._pnp_refine(X, x, (fx, fy, cx, cy), R, T, w2, max(iterations // 2, 1))
        return torch.cat([T, self._so3_log(R)], 1).to(pts2d.dtype)

    def _pnp_refine(self, X, x, cam, R, T, wgt, iterations):
        """Levenberg-Marquardt on SE(3) over the correspondences with weight 1 (wgt [b,n])."""
        fx, fy, cx, cy = cam
        b, n = X.shape[0], X.shape[1]
        sw = wgt.unsqueeze(2)

        def residual(R, T):
            Y = X.bmm(R.transpose(1, 2)) + T.unsqueeze(1)
            z = Y[:, :, 2].clamp_min(1e-9)
            r = torch.stack([fx * Y[:, :, 0] / z + cx - x[:, :, 0], fy * Y[:, :, 1] / z + cy - x[:, :, 1]], 2) * sw
            return Y, z, r
        lam = torch.full((b, 1, 1), 1e-3, dtype=X.dtype, device=X.device)
        Y, z, r = residual(R, T)
        cost = (r * r).sum((1, 2))
        eye6 = torch.eye(6, dtype=X.dtype, device=X.device).unsqueeze(0)
        for _ in range(iterations):
            # d(u, v) / dY, dY / d(omega, T) with R <- exp([omega]x) R: dY/domega = -[Y - T]x, dY/dT = I
            zi = 1.0 / z
            du = torch.stack([fx * zi, torch.zeros_like(zi), -fx * Y[:, :, 0] * zi * zi], 2)
            dv = torch.stack([torch.zeros_like(zi), fy * zi, -fy * Y[:, :, 1] * zi * zi], 2)
            P = Y - T.unsqueeze(1)
            zero = torch.zeros_like(zi)
            Pc = torch.stack([zero, -P[:, :, 2], P[:, :, 1], P[:, :, 2], zero, -P[:, :, 0], -P[:, :, 1], P[:, :, 0], zero], 2).view(b, n, 3, 3)
            Ju = torch.cat([-(du.unsqueeze(2) @ Pc).squeeze(2), du], 2) * sw     # [b,n,6]
            Jv = torch.cat([-(dv.unsqueeze(2) @ Pc).squeeze(2), dv], 2) * sw
            J = torch.cat([Ju, Jv], 1)                                           # [b,2n,6]
            rr = torch.cat([r[:, :, 0], r[:, :, 1]], 1).unsqueeze(2)             # [b,2n,1]
            H = J.transpose(1, 2).bmm(J)
            g = J.transpose(1, 2).bmm(rr)
            step = torch.linalg.solve(H + lam * (eye6 * torch.diagonal(H, dim1=1, dim2=2).unsqueeze(1).clamp_min(1e-12)), -g).squeeze(2)
            Rn, Tn = self._so3_exp(step[:, :3]).bmm(R), T + step[:, 3:]
            Yn, zn, rn = residual(Rn, Tn)
            cn = (rn * rn).sum((1, 2))
            ok = (cn < cost)
            m3, m1 = ok.view(b, 1, 1), ok.view(b, 1)
            R, T = torch.where(m3, Rn, R), torch.where(m1, Tn, T)
            Y, z, r = torch.where(m3, Yn, Y), torch.where(m1, zn, z), torch.where(m3, rn, r)
            cost = torch.where(ok, cn, cost)
            lam = torch.where(m3, lam * 0.3, lam * 5.0).clamp(1e-9, 1e6)
        return R, T

    def compute_pnp_loss(self, depth, matches, pose_vec, K, K_inv):
        """model_geometry.py:498-530: back-project the first view's matched pixels with their depth, solve the pose that
        reprojects them onto the second view's matched pixels, L1 between (T, axis-angle) and the pose vector's
        (translation, Euler angles) -- the reference compares those two parameterisations as they stand."""
        b, _, n = matches.shape
        ones = torch.ones(b, 1, n, device=matches.device, dtype=matches.dtype)
        pts3d = (K_inv.bmm(torch.cat([matches[:, :2], ones], 1)) * depth).transpose(1, 2)
        pose_pred = self.pnp(matches[:, 2:].transpose(1, 2), pts3d, K[0])
        beta = getattr(self, "beta", 1)
        return F.l1_loss(pose_pred[:, :3], pose_vec[:, :3], reduction="none") + \
            beta * F.l1_loss(pose_pred[:, 3:], pose_vec[:, 3:], reduction="none")
