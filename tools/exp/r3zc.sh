# round 3, batch zc: the NMS chain ALONE (tools/wnms_bench.py) under rocprofv3: unloaded per-kernel durations
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3zc; mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python tools/wnms_bench.py > $O/log.txt 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); cp $f $O/nms_alone_kernel_stats.csv; rm -rf $O/st
grep -E "wnms|filter|dets12" $O/nms_alone_kernel_stats.csv | cut -c1-40,100-
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r3zc/nms_alone_kernel_stats.csv')):
    n=r['Name']
    if any(k in n for k in ('wnms','filter','dets12')):
        print("%-60s calls %4s avg %9.1f us" % (n[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
