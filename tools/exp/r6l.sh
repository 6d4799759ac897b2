#!/bin/bash
# round 6: the final 16 x 16 x 32 forms (plain + fused output conv straight from the registers) against the 32 x 32 x 16 plan, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=gpurun_out/r6l; mkdir -p $O
bash tools/exp/ab.sh r6l "RD_NO_MFMA16=1" "" 3 both "mfma16 or fused_with_head_out or every_production_launch" 2>&1 | tee $O/ab_all.txt
