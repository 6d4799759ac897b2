cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3x; mkdir -p $O
python tools/profile_steps.py bf16 10 8 2>&1 | grep -v amdgpu > $O/steps_pair.txt
RD_NO_PAIR=1 python tools/profile_steps.py bf16 10 8 2>&1 | grep -v amdgpu > $O/steps_nopair.txt
grep -E "rpn|cls_|sum of" $O/steps_pair.txt; grep -E "rpn|sum of" $O/steps_nopair.txt | cut -c1-110
