# Per-launch-class counters of the persistent 3x3 conv at W 2656 x 8 frames through the PRODUCTION entry point (tools/conv64_bench.py):
#   64->64 (conv1 of a BasicBlock), 64->64 + residual (conv2), 128->128 (head tower) -- VERDICT r4 "What's weak" 3 / next-round 1(b):
#   is the cout-64 form (a wave owns 2 pixel x 2 channel fragments: 1.0 LDS fragment read per MFMA) LDS-, HBM- or clock-limited
#   against the cout-128 form (2 x 4 fragments: 0.75 reads per MFMA)?  Separate --pmc passes, kernel-trace only.
# Run through gpurun: bash tools/pmc_conv_classes.sh [out tag]   -> gpurun_out/pcc/summary.json (copy to profiles/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/pcc; mkdir -p $O
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
         "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_LEVEL_LDS"
         "FETCH_SIZE" "WRITE_SIZE")
for cls in c64 c64add c128; do
  case $cls in c64) E="RES=0";; c64add) E="RES=1";; c128) E="RES=0 C128=only";; esac
  # wall time of the class without counters (and its TFLOP/s line)
  env $E WS=2656 ITER=20 timeout -s KILL 120 python tools/conv64_bench.py > $O/$cls.time.txt 2>&1
  i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1))
    env $E WS=2656 ITER=6 timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$cls.g$i -- python tools/conv64_bench.py > $O/$cls.g$i.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections, json, re
out = {}
for cls in ("c64", "c64add", "c128"):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pcc/%s.g*/**/*counter_collection.csv" % cls, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "conv3x3_stream" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, c), v in per.items():
            acc[c].append(v)
    e = {c: sum(v) / len(v) for c, v in acc.items()}
    e["dispatches_sampled"] = max((len(v) for v in acc.values()), default=0)
    try:
        t = open("gpurun_out/pcc/%s.time.txt" % cls).read()
        m = re.search(r"([\d.]+) us\s+([\d.]+) TFLOP/s", t)
        e["us"], e["tflops"] = float(m.group(1)), float(m.group(2))
    except Exception as ex:  # noqa: BLE001
        e["time_error"] = repr(ex)
    g = e.get("GRBM_GUI_ACTIVE")
    if g and e.get("us"):
        e["shader_clock_mhz_from_gui_active"] = g / e["us"]          # (summed over instances: divide by the instance count printed in the log)
    if e.get("SQ_BUSY_CYCLES"):
        for k in ("SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if k in e:
                e[k + "_per_SQ_BUSY_CYCLES"] = e[k] / e["SQ_BUSY_CYCLES"]
    if e.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in e:
                e[k + "_per_SQ_WAVE_CYCLES"] = e[k] / e["SQ_WAVE_CYCLES"]
    if e.get("SQ_LDS_IDX_ACTIVE") and e.get("SQ_LDS_BANK_CONFLICT") is not None:
        e["lds_bank_conflict_frac_of_lds_active"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    out[cls] = e
json.dump(out, open("gpurun_out/pcc/summary.json", "w"), indent=1)
for cls, e in out.items():
    print(cls, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if "per_" in k or k in ("us", "tflops", "lds_bank_conflict_frac_of_lds_active")})
PY
rm -rf $O/*.g[0-9]
