#!/bin/bash
# Round-6 quality evidence on one MI355X (about 40 GPU-minutes):  bash tools/ab_round6.sh
#  1. event-only training, paired by seed: A (split-bf16 default) / X (exact fp32 MFMA) / B (reference-shaped route on this
#     library's entry points) -- tools/psnr_ab_events.py
#  2. the RGB-MSE A/B of earlier rounds again, on the kernels as they are now -- tools/psnr_ab.py, then with route B on the
#     reference's own kernels (tests/refcheck/psnr_vs_reference_kernels.py)
#  3. event-only arm B on the REFERENCE's own kernels + torch GEMMs + torch Adam (138 ms/step: a handful of seeds)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06/ab
mkdir -p $OUT
cd $R
STEPS=${STEPS:-1000}
python tools/psnr_ab_events.py $STEPS ${N_AXB:-24} $OUT/events_AXB_s0.json 0 AXB > $OUT/events_AXB_s0.log 2>&1
tail -1 $OUT/events_AXB_s0.log | cut -c1-1500
python tools/psnr_ab_events.py $STEPS ${N_AX:-40} $OUT/events_AX_s24.json ${N_AXB:-24} AX > $OUT/events_AX_s24.log 2>&1
tail -1 $OUT/events_AX_s24.log | cut -c1-1500
ENERF_PSNR_NO_DROPS=1 python tools/psnr_ab.py 1500 ${N_RGB:-30} $OUT/rgb_AB_nodrops.json > $OUT/rgb_AB_nodrops.log 2>&1
tail -1 $OUT/rgb_AB_nodrops.log | cut -c1-800
ENERF_PSNR_NO_DROPS=1 python -B tests/refcheck/psnr_vs_reference_kernels.py 1500 ${N_RGB_REF:-12} $OUT/rgb_AB_refkernels_nodrops.json > $OUT/rgb_AB_refkernels_nodrops.log 2>&1
tail -1 $OUT/rgb_AB_refkernels_nodrops.log | cut -c1-800
python -B tests/refcheck/psnr_events_vs_reference_kernels.py $STEPS ${N_REF:-6} $OUT/events_Bref_s0.json 0 B > $OUT/events_Bref_s0.log 2>&1
tail -1 $OUT/events_Bref_s0.log | cut -c1-800
