cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zd; mkdir -p $O
python tools/exp/r3zd_clip_rate.py 2>&1 | grep -v amdgpu | tee $O/clips.txt
