"""Summarise rocprofv3 --pmc passes (csv) per kernel: python tools/pmc_summary.py <batch> <dir> [<dir> ...] > profiles/<name>.json
Every <dir> is the -d output of one `rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python bench.py ...`
pass.  Per kernel-name prefix: mean counter value per dispatch; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB
(gfx950: FETCH_SIZE under-reports wide coalesced reads 2x, MI355X_MICROARCH.md HBM section)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = {"conv3x3_stream_kernel": "conv3x3_stream_kernel", "block64_stream_kernel": "block64_stream_kernel", "conv1x1_stream_kernel": "conv1x1_stream_kernel",
           "conv_taps_kernel": "conv_taps_kernel", "input_transform_kernel": "input_transform_kernel",
           "meta16_kernel": "meta_kernel", "meta_bf16_kernel": "meta_kernel", "head_out_mfma_kernel": "head_out_mfma_kernel"}
batch = int(sys.argv[1])
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)      # (dispatch, counter) -> value summed over dimensions (XCC / SE instances)
        name_of = {}
        for row in csv.DictReader(open(f)):
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per[key] += float(row["Counter_Value"])
            name_of[row["Dispatch_Id"]] = row["Kernel_Name"]
        for (disp, cname), v in per.items():
            kn = name_of[disp]
            for pref, out in KERNELS.items():
                if pref in kn:
                    acc[out][cname].append(v)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rangedet_amd.build import source_hash  # noqa: E402
res = {"csrc_sha16": source_hash(), "command": "rocprofv3 --kernel-trace --pmc <COUNTERS> --output-format csv -- python bench.py --steps 3 --warmup 1 "
                  "--repeats 1 --backbone-reps 0 --no-cpu-baseline --inflight 1  (one pass per counter group)", "batch": batch,
       "note": "per-dispatch means; FETCH_SIZE/WRITE_SIZE in KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 "
               "FETCH correction, MI355X_MICROARCH.md HBM section)"}
for k, cs in acc.items():
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["dispatches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        # same normalisation as round-1a: MFMA-busy cycles summed over instances / (GPU-active cycles * 128)
        e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] * 128)
    res[k] = e
print(json.dumps(res, indent=1))
