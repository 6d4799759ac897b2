"""Device input transform chain at full size (dev tool): python tools/input_bench.py [B]  -> us per launch, GB/s"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rangedet_amd import synth  # noqa: E402
from rangedet_amd.input_transform import DeviceInputTransform  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
recs = [synth.raw_record(i) for i in range(B)]
T = DeviceInputTransform((64, 2656))
out = T(recs)
torch.cuda.synchronize()
ref = synth.transform(recs[0])
for k, v in out.items():
    got = v[0].cpu().numpy()
    err = np.abs(got - ref[k][0]).max()
    assert err < 1e-6, (k, err)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
A, L = T.A, T.L
import ctypes  # noqa: E402
ri = A.upload(np.stack([r['range_image'] for r in recs]))
pc = A.upload(np.stack([r['pc_vehicle_frame'] for r in recs]))
inc = A.upload(np.stack([r['inclination'] for r in recs]))
ptrs = [v.data_ptr() for v in out.values()]
names = list(out.keys())
order = ['input_data', 'coord_s1', 'pc_vehicle_frame_s1', 'pc_vehicle_frame_s2', 'pc_vehicle_frame_s4',
         'range_image_mask_s1', 'range_image_mask_s2', 'range_image_mask_s4']
args = [out[k].data_ptr() for k in order]
st = torch.cuda.current_stream().cuda_stream
e0.record()
for _ in range(20):
    L.call("rd_input_transform", A.ptr(ri), A.ptr(pc), A.ptr(inc), ctypes.addressof(T.norm), B, 64, 2650, 64, 2656, *args, st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
byts = B * (64 * 2650 * 7 * 4 + 64 * 2656 * (11 + 4 * (1 + 0.5 + 0.25)) * 4)
print("input transform: %d frames, %.1f us per launch, %.0f GB/s (%.1f MB moved), identical to the host chain (azimuth 1e-6)" % (B, us, byts / us / 1e3, byts / 1e6))
