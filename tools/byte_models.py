"""Algorithmic byte models of the fused loss stack's kernels: the reads and writes a launch cannot
avoid, fp32, for a batch of B triplets at H x W with S scales.  Used by bench.py (the `roofline` object) and by
tools/roofline_table.py (the per-kernel table), so both always price a kernel with the same figure.

Level 0 of the bilinear and of the area pyramid of a source frame is the frame itself -- one tensor, read for both
purposes -- so it is counted once there (round 2 counted it twice: 105 B/px at every level)."""


def scale_pixels(H, W, S):
    return [int(H / 2 ** s) * int(W / 2 ** s) for s in range(S)]


def models(B, H, W, S, mode="geom"):
    N = scale_pixels(H, W, S)
    SN, N0, NL = sum(N), N[0], sum(N[1:])
    m = {
        "k_geom_pyramids": (3 * B * 3 * N0 * 4 + 3 * B * 3 * NL * 4 + 2 * B * 3 * NL * 4,
                            "read 3 frames 12 B/px + write bilinear levels>=1 of 3 frames + area levels>=1 of 2 frames"),
        "k_geom_pyramids12": (3 * B * 3 * N0 * 4 + 3 * B * 3 * NL * 4 + 2 * B * 3 * NL * 4,
                              "levels 1-2 from one read: read 3 frames 12 B/px + write bilinear levels 1-2 of 3 frames + area levels 1-2 of 2 frames"),
        "k_geom_point_fwd": (B * (81 * N0 + 105 * NL),
                             "per px: target 12 + 2 flows 16 + 2 bilinear sources 24 + 2 area sources 24 (level 0: the same tensor, counted once) + disp 4; mask 1 + masked warps 24 = 105 (81 at level 0)"),
        "k_depth_point_fwd": (B * (41 * N0 + 65 * NL), "per px: target 12 + 2 area sources 24 + 2 bilinear sources 24 (level 0: once) + disp 4; mask 1 = 65 (41 at level 0)"),
        "k_flow_point_fwd": (B * 84 * SN, "84 B/px: target 12 + 2 flows 16 + 2 sources 24; weights 8 + weighted warps 24"),
        "k_geom_ssim_fwd_roll": (2 * 25 * B * SN, "25 B/px/dir: target 12 + warp 12 + mask 1"),
        "k_geom_flow_smooth_fwd": (28 * B * SN, "28 B/px: target 12 + 2 flows 16"),
        "k_geom_disp_smooth_fwd": (3 * B * (16 * N0 + 4 * NL), "3 frames: image 12 + disp 4 per full-res px + the low-res disparities"),
        "k_geom_ssim_bwd_roll": (2 * 37 * B * SN, "37 B/px/dir: 25 + write dL/dwarp 12"),
        "k_geom_point_bwd": (B * (100 * N0 + 124 * NL), "per px: forward reads 80 (56 at level 0) + dL/dwarp 24; write grad_flow 16 + grad_disp 4"),
        "k_geom_flow_smooth_bwd": (60 * B * SN, "60 B/px: 28 + read-modify-write of grad_flow 32"),
        "k_geom_disp_smooth_bwd1": (3 * B * N0 * (12 + 4 + 2 + 4 + 4 * (S - 1)),
                                    "3 frames per full-res px: image 12 + disp 4 + up-sampled rows ~2; write grad_disp0 4 + up-sampled grads 4(S-1)"),
        "k_geom_disp_smooth_bwd2": (3 * B * (4 * (S - 1) * N0 + 4 * NL), "read the up-sampled grads once, write grad_disp of levels >= 1"),
    }
    # adjoint of the bilinear up-sampling in two passes (round 4, loss_stack_bwd.hip k_geom_adj_rows / k_geom_adj_cols): every
    # level >= 1 for generic sizes, the levels coarser than 1/4 where levels 1-2 are exact (the register gather keeps those)
    Hs, Ws = [int(H / 2 ** s) for s in range(S)], [int(W / 2 ** s) for s in range(S)]
    exact = all((Hs[s] << s) == H and (Ws[s] << s) == W for s in range(1, min(S, 3)))
    s0 = 1
    if exact:
        while s0 < S and H <= 4 * Hs[s0] and W <= 4 * Ws[s0]:
            s0 += 1
    if s0 < S:
        seg = 0
        for s in range(s0, S):
            L = max(1, min(32, int(6.0 * H / Hs[s])))
            seg += ((H + L - 1) // L) * 8 * W          # partial-sum floats per (frame, sample) of this level
        m["k_geom_adj_rows"] = (3 * B * 4 * ((S - s0) * N0 + seg), "read the up-sampled gradients of the levels >= %d once (4 B per full-res px and level), write the per-segment column sums" % s0)
        m["k_geom_adj_cols"] = (3 * B * 4 * (seg + sum(N[s0:])), "read the segment sums once, write grad_disp of the levels >= %d" % s0)
        if s0 == 1:
            m.pop("k_geom_disp_smooth_bwd2", None)
        else:
            m["k_geom_disp_smooth_bwd2"] = (3 * B * (4 * (s0 - 1) * N0 + 4 * sum(N[1:s0])), "read the up-sampled grads of levels 1..%d once, write their grad_disp" % (s0 - 1))
    return m


POINT_KERNEL = {"geom": "k_geom_point_fwd", "depth": "k_depth_point_fwd", "flow": "k_flow_point_fwd"}


# ---- VALU-issue roof of the pointwise kernels (bench.py roofline.valu_frac)
# Executed vector-ALU instructions per wave (rocprofv3 SQ_INSTS_VALU / SQ_WAVES, profiles/r03_pmc_loss_stack.json; the
# static mix is profiles/r03_point_fwd_isa_mix.md) and what one of them costs a SIMD when >= 2 waves share it, measured
# with tools/ubench/valu.hip (profiles/r02_issue_cost_model.md section 1, at an assumed 2.4 GHz): 2.64 cycles for
# mul / add / fma, 3.9 for compare + select pairs (22 % of this kernel's mix), ~8 for the few transcendentals.
VALU_PER_WAVE = {"k_geom_point_fwd": 887.1}
VALU_CYCLES_PER_INST = 0.76 * 2.64 + 0.22 * 3.9 + 0.02 * 8.2      # = 3.03
GPU_SIMDS, GPU_CLOCK_HZ = 256 * 4, 2.4e9


def valu_time_s(kernel, pixels_threads, valu_per_wave=None):
    """Time the launch's vector-ALU instructions alone need on the whole chip (every SIMD issuing back to back)."""
    v = valu_per_wave if valu_per_wave is not None else VALU_PER_WAVE.get(kernel)
    if v is None:
        return None
    waves = (pixels_threads + 63) // 64
    return waves * v * VALU_CYCLES_PER_INST / GPU_SIMDS / GPU_CLOCK_HZ
