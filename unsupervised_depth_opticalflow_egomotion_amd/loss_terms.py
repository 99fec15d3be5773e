"""Per-method loss/mask API of the reference models, on the HIP device.

``Model_geometry`` / ``Model_depth`` / ``Model_flow`` inherit these so that every ``compute_*`` /
``fusion_*`` method of the reference (model_geometry.py:46-765, model_flow.py:94-152) can still be called
one at a time.  The gathers / stencils run the per-operator HIP kernels (warp_flow, inverse_warp2,
calculate_rigid_flow, SSIM, resize); every {0,1} mask decision runs the fused stack's own decision code
(ops.occ_masks / texture_mask / dynamic_mask), the differentiable element-wise glue stays in torch device ops.
``Model_geometry.forward`` does NOT go through these: it calls the fused stack (loss_stack.py)."""
import torch
import torch.nn.functional as F

from . import ops
from .pytorch_ssim import SSIM
from .structures import warp_flow, inverse_warp2, calculate_rigid_flow, compute_essential_matrix


class LossTerms:
    num_scales = 3
    flow_consist_alpha = 0.01
    flow_consist_beta = 0.5
    rigid_thres = 0.5
    inlier_thres = 0.1

    # ---- norms (model_geometry.py:46-63)
    def get_flow_norm(self, flow, p=2):
        return torch.norm(flow, p=p, dim=1).unsqueeze(1) + 1e-12

    def get_flow_normalization(self, flow, p=2):
        return flow / self.get_flow_norm(flow, p).repeat(1, 2, 1, 1)

    # ---- pyramids (model_geometry.py:65-78, model_flow.py:58-64)
    def generate_img_pyramid(self, img, num_pyramid):
        h, w = img.shape[2], img.shape[3]
        return [img if s == 0 else ops.resize(img, (int(h / 2 ** s), int(w / 2 ** s)), "bilinear")
                for s in range(num_pyramid)]

    def generate_img_pyramid_avgpool(self, img, num_pyramid):
        h, w = img.shape[2], img.shape[3]
        return [img.detach() if s == 0 else ops.resize(img, (int(h / 2 ** s), int(w / 2 ** s)), "area")
                for s in range(num_pyramid)]

    def warp_flow_pyramid(self, img_pyramid, flow_pyramid):
        return [warp_flow(i, f, use_mask=True) for i, f in zip(img_pyramid, flow_pyramid)]

    # ---- rigid reconstruction (model_geometry.py:80-103)
    def reconstruction(self, ref_img, intrinsics, depth, depth_ref, pose, padding_mode="zeros"):
        rec, valid, pdepth, cdepth = [], [], [], []
        for s in range(self.num_scales):
            _, _, h, w = depth[s].size()
            src = ref_img if (h, w) == tuple(ref_img.shape[2:]) else ops.resize(ref_img, (h, w), "area")
            down = ref_img.size(2) / h
            k_s = torch.cat((intrinsics[:, 0:2] / down, intrinsics[:, 2:]), dim=1)
            a, b, c, d = inverse_warp2(src, depth[s], depth_ref[s], pose, k_s, padding_mode)
            rec.append(a); valid.append(b); pdepth.append(c); cdepth.append(d)
        return rec, valid, pdepth, cdepth

    # ---- masks
    def compute_occ_weight(self, from_l, tgt, from_r):
        """Hard occlusion masks (model_geometry.py:105-132): (1 - softmax([dl, dr])) > 0.48."""
        w_bwd, w_fwd, v_bwd, v_fwd = [], [], [], []
        for s in range(self.num_scales):
            ob, of, vb, vf = ops.occ_masks(from_l[s], tgt[s], from_r[s])   # the fused stack's own decision code
            w_bwd.append(ob); w_fwd.append(of); v_bwd.append(vb); v_fwd.append(vf)
        return w_bwd, w_fwd, v_bwd, v_fwd

    def compute_diff_weight(self, from_l, tgt, from_r):
        """Soft gaussian occlusion weights of Model_flow (model_flow.py:105-138)."""
        d_bwd, d_fwd, w_bwd, w_fwd = [], [], [], []
        for s in range(self.num_scales):
            il, it, ir = from_l[s], tgt[s], from_r[s]
            vf = 1 - (ir == 0).prod(1, keepdim=True).type_as(ir)
            vb = 1 - (il == 0).prod(1, keepdim=True).type_as(il)
            dl = torch.abs(it - il).mean(1, True)
            dr = torch.abs(it - ir).mean(1, True)
            wgt = (1 - F.softmax(torch.cat((dl, dr), 1), 1)).detach()
            wgt = 2 * torch.exp(-(wgt - 0.5) ** 2 / 0.03)
            w_bwd.append(wgt[:, 0:1] * vb); w_fwd.append(wgt[:, 1:2] * vf)
            d_fwd.append(dr); d_bwd.append(dl)
        return d_bwd, d_fwd, w_bwd, w_fwd

    def compute_texture_mask(self, img_list, img_warped_list, img_list_source):
        return [ops.texture_mask(img_list[s], img_warped_list[s], img_list_source[s]) for s in range(self.num_scales)]

    def compute_dynamic_mask(self, intrinsics, depth, pose, flow):
        """(model_geometry.py:685-713) -> flow_diffs, dynamic masks, scores."""
        diffs, masks, scores = [], [], []
        h0 = depth[0].size(2)
        for s in range(self.num_scales):
            down = h0 / depth[s].size(2)
            k_s = torch.cat((intrinsics[:, 0:2] / down, intrinsics[:, 2:]), dim=1)
            rigid = calculate_rigid_flow(depth[s], pose, k_s)
            diffs.append(torch.abs(rigid - flow[s]))
            m, sc = ops.dynamic_mask(flow[s], rigid, self.flow_consist_alpha, self.flow_consist_beta)
            masks.append(m); scores.append(sc)
        return diffs, masks, scores

    def get_rigid_mask(self, dist_map):
        with torch.no_grad():
            rigid = (dist_map < self.rigid_thres).float()
            inlier = (dist_map < self.inlier_thres).float()
            score = rigid * 1.0 / (1.0 + dist_map)
        return rigid, inlier, score

    def fusion_mask(self, valid_mask, occ_mask, dynamic_mask):
        return [valid_mask[s] * occ_mask[s] * dynamic_mask[s] for s in range(self.num_scales)]

    def fusion_mask_4item(self, valid_mask, occ_mask, dynamic_mask, texture_mask):
        return [valid_mask[s] * occ_mask[s] * dynamic_mask[s] * texture_mask[s] for s in range(self.num_scales)]

    def fusion_mask_2item(self, valid_mask, occ_mask):
        return [valid_mask[s] * occ_mask[s] for s in range(self.num_scales)]

    # ---- losses
    @staticmethod
    def _masked_mean(value, mask, channels):
        div = mask.mean((1, 2, 3))
        return (value * mask.repeat(1, channels, 1, 1)).mean((1, 2, 3)) / (div + 1e-12)

    def compute_photometric_loss(self, img_list, img_warped_list, mask_list):
        terms = [self._masked_mean(torch.abs(img_list[s] - img_warped_list[s]), mask_list[s], 3)[:, None]
                 for s in range(self.num_scales)]
        return torch.cat(terms, 1).sum(1)

    def compute_loss_with_mask(self, diff_list, occ_mask_list):
        terms = [self._masked_mean(diff_list[s], occ_mask_list[s], 3)[:, None] for s in range(self.num_scales)]
        return torch.cat(terms, 1).sum(1)

    def compute_ssim_loss(self, img_list, img_warped_list, mask_list):
        terms = []
        for s in range(self.num_scales):
            m3 = mask_list[s].repeat(1, 3, 1, 1)
            val = torch.clamp((1.0 - SSIM(img_list[s] * m3, img_warped_list[s] * m3)) / 2.0, 0, 1).mean((1, 2, 3))
            terms.append((val / (mask_list[s].mean((1, 2, 3)) + 1e-12))[:, None])
        return torch.cat(terms, 1).sum(1)

    compute_loss_ssim = compute_ssim_loss   # Model_flow's name (model_flow.py:141-152)

    def compute_consis_loss(self, predicted_depth_list, computed_depth_list, mask_list):
        """Disabled in the reference's forward but part of its API (model_geometry.py:182-193)."""
        terms = []
        for s in range(self.num_scales):
            p, c, m = predicted_depth_list[s], computed_depth_list[s], mask_list[s]
            diff = ((c - p).abs() / (c + p).abs()).clamp(0, 1) * m
            terms.append((diff.mean((1, 2, 3)) / (m.mean((1, 2, 3)) + 1e-12))[:, None])
        return torch.cat(terms, 1).sum(1)

    def compute_smooth_loss(self, img, disps):
        h, w = img.shape[2], img.shape[3]
        gix = torch.exp(-torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True))
        giy = torch.exp(-torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True))
        terms = []
        for s in range(self.num_scales):
            d = F.interpolate(disps[s], size=(h, w), mode="bilinear", align_corners=False)
            gx = torch.abs(d[:, :, :, :-1] - d[:, :, :, 1:]) * gix
            gy = torch.abs(d[:, :, :-1, :] - d[:, :, 1:, :]) * giy
            terms.append((gx.mean((1, 2, 3)) + gy.mean((1, 2, 3)))[:, None])
        return torch.cat(terms, 1).sum(1)

    def gradients(self, img):
        return img[:, :, :, 1:] - img[:, :, :, :-1], img[:, :, 1:, :] - img[:, :, :-1, :]

    def cal_grad2_error(self, flow, img):
        ix, iy = self.gradients(img)
        wx = torch.exp(-10.0 * torch.abs(ix).mean(1).unsqueeze(1))
        wy = torch.exp(-10.0 * torch.abs(iy).mean(1).unsqueeze(1))
        dx, dy = self.gradients(flow)
        dx2, _ = self.gradients(dx)
        _, dy2 = self.gradients(dy)
        err = (wx[:, :, :, 1:] * torch.abs(dx2)).mean((1, 2, 3)) + (wy[:, :, 1:, :] * torch.abs(dy2)).mean((1, 2, 3))
        return err / 2.0

    def compute_loss_flow_smooth(self, optical_flows, img_pyramid):
        terms = [self.cal_grad2_error(optical_flows[s] / 20.0, img_pyramid[s])[:, None] for s in range(self.num_scales)]
        return torch.cat(terms, 1).sum(1)

    def compute_loss_flow_consis(self, fwd_flow_pyramid, bwd_flow_pyramid, occ_mask_list):
        terms = []
        for s in range(self.num_scales):
            uf = self.get_flow_normalization(fwd_flow_pyramid[s])
            ub = self.get_flow_normalization(bwd_flow_pyramid[s]).float().detach()
            inv = 1 - occ_mask_list[s]
            val = (torch.abs(uf + ub) * inv).mean((1, 2, 3)) / (inv.mean((1, 2, 3)) + 1e-12)
            terms.append(val[:, None])
        return torch.cat(terms, 1).sum(1)

    def compute_depth_flow_consis_loss(self, flow_diffs, masks=None, scales=3):
        terms = []
        for s in range(scales):
            diff = flow_diffs[s]
            b, _, hh, ww = diff.size()
            mask = torch.ones(b, 1, hh, ww, device=diff.device) if masks is None else masks[s]
            terms.append(self._masked_mean(diff, mask, 2)[:, None])
        return torch.cat(terms, 1).sum(1)

    # ---- epipolar (model_geometry.py:304-425)
    def meshgrid(self, B, H, W):
        xs = torch.arange(0, W).view(1, 1, 1, W).expand(B, 1, H, W)
        ys = torch.arange(0, H).view(1, 1, H, 1).expand(B, 1, H, W)
        return torch.cat((xs, ys), 1).float()

    def compute_epipolar_map(self, pose, flow, intrinsics, intrinsics_inverse):
        b, _, h, w = flow.size()
        grid = self.meshgrid(b, h, w).to(flow.device)
        ones = torch.ones(b, 1, h * w, device=flow.device)
        p1 = torch.cat([grid.view(b, 2, -1), ones], 1)
        p2 = torch.cat([(grid + flow).view(b, 2, -1), ones], 1)
        E = compute_essential_matrix(pose)
        Fm = intrinsics_inverse.transpose(1, 2).bmm(E.bmm(intrinsics_inverse))
        line = Fm.bmm(p1)
        div = torch.sqrt(line[:, 0:1] * line[:, 0:1] + line[:, 1:2] * line[:, 1:2]) + 1e-6
        dist = torch.abs(torch.sum(p2 * line, dim=1, keepdim=True)) / div
        return dist.view(b, 1, h, w)

    def compute_epipolar_loss(self, dist_map, rigid_mask):
        """The reference overwrites the masked mean with the plain mean (model_geometry.py:413-418)."""
        return dist_map.mean((1, 2, 3))

    # ---- depth (model_geometry.py:274-292)
    def disp2depth(self, disp, min_depth=0.1, max_depth=100.0):
        min_disp, max_disp = 1 / max_depth, 1 / min_depth
        return 1 / (min_disp + (max_disp - min_disp) * disp)
