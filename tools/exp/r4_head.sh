#!/bin/bash
# round 4: fused output conv straight from the accumulators (no LDS transpose) -- base build vs new build, then the tile choice
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/exp/ab.sh r4h1 "RANGEDET_HIP_LIB=rangedet_amd/librangedet_hip_base.so" "" 2 both "head or pair_equals or production_launch"
bash tools/exp/ab.sh r4h2 "RD_CONV_HEAD30=2" "" 1 both
