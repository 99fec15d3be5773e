for k in 0 16; do
python tools/ls_segments.py --flow-noise 0 --tag "S3B4 smooth tile$k" --lib scratch/abl/libdfe_hip_tile$k.so 2>/dev/null | grep " us" | sed 's/prep.*point_fwd/point_fwd/; s/ssim_fwd.*//; s/flow_smooth_bwd.*//'
python tools/ls_segments.py --flow-noise 0 --tag "cfg4B16 smooth tile$k" --batch 16 --height 375 --width 1242 --scales 6 --iters 40 --lib scratch/abl/libdfe_hip_tile$k.so 2>/dev/null | grep " us" | sed 's/prep.*point_fwd/point_fwd/; s/ssim_fwd.*//; s/flow_smooth_bwd.*//'
python tools/ls_segments.py --flow-noise 0 --tag "cfg4B2 smooth tile$k" --batch 2 --height 375 --width 1242 --scales 6 --iters 100 --lib scratch/abl/libdfe_hip_tile$k.so 2>/dev/null | grep " us" | sed 's/prep.*point_fwd/point_fwd/; s/ssim_fwd.*//; s/flow_smooth_bwd.*//'
python tools/ls_segments.py --tag "cfg4B2 noise tile$k" --batch 2 --height 375 --width 1242 --scales 6 --iters 100 --lib scratch/abl/libdfe_hip_tile$k.so 2>/dev/null | grep " us" | sed 's/prep.*point_fwd/point_fwd/; s/ssim_fwd.*//; s/flow_smooth_bwd.*//'
done
