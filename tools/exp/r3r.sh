# round 3, batch r: fused output conv of the full-width level on the 8 x 32 two-workgroup tiles (RD_CONV_HEAD30=2) vs the 8 x 62 tile
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3r; mkdir -p $O
b() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["meta_dla_forward"]["frac_hbm_peak"],4), round(d["roofline"]["frac"],4))'; }
for i in 1 2 3; do echo "HEAD30=1 $(RD_CONV_HEAD30=1 b)"; echo "HEAD30=2 $(RD_CONV_HEAD30=2 b)"; done | tee $O/ab.txt
