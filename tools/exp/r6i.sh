#!/bin/bash
# round 6: the full-width fused-output-conv launches (rpn_*_conv_3_lvl_0) on the 8 x 32 / two-workgroup tiles in the 16 x 16 x 32 form
# (RD_CONV_HEAD30=2 in the dev library: rd_conv3x3_mfma16_ok then says yes at W 2656) against the 8 x 62 / one-workgroup 32 x 32 x 16 form.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6i; mkdir -p $O
bash tools/exp/ab.sh r6i "" "RD_CONV_HEAD30=2" 3 both 2>&1 | tee $O/ab_all.txt
