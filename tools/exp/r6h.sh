#!/bin/bash
# round 6: the v_mfma_f32_16x16x32 form of the cout-128 convs (RD_MFMA16, lower._mark_mfma16) against the 32x32x16 form, same box:
# parity tests first, per-step tables, three bench alternations; then the MFMA power probe with the two MFMA orders.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6h; mkdir -p $O
bash tools/exp/ab.sh r6h "RD_NO_MFMA16=1" "" 3 both "mfma16 or fused_with_head_out" 2>&1 | tee $O/ab_all.txt
timeout -s KILL 600 tools/micro/mfma_power.bin long 2>&1 | tee $O/mfma_power_long.txt | grep -E "post-ReLU" | cut -c1-140
