#!/bin/bash
# round 6: the fused 64-channel BasicBlocks in the 16 x 16 x 32 form too (rd_block64_m16_bn_act) against the 32 x 32 x 16 blocks, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6n; mkdir -p $O
bash tools/exp/ab.sh r6n "RD_NO_MFMA16_BLOCK=1" "" 3 both "block64 or block_fusion or every_production_launch" 2>&1 | tee $O/ab_all.txt
