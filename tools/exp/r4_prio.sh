#!/bin/bash
# round 4: (i) the 16-byte output store of the fused box-regression head, base build vs this build; (ii) queue priorities of the
# launch streams / the post-processing streams (RD_LAUNCH_STREAM_PRIO, RD_POST_STREAM_PRIO: 0 normal, -1 high)
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/exp/ab.sh r4i1 "RANGEDET_HIP_LIB=rangedet_amd/librangedet_hip_base.so" "" 2 both "fused_with_head or pair_equals"
bash tools/exp/ab.sh r4i2 "RD_LAUNCH_STREAM_PRIO=-1" "" 2 bench
bash tools/exp/ab.sh r4i3 "RD_POST_STREAM_PRIO=-1" "" 2 bench
