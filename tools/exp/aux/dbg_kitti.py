import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from rangedet_amd import lib as R, synth
from rangedet_amd.config import rangedet_veh_wo_aug_4_18e as cfgmod
from rangedet_amd.pipeline import RangeDetPipeline
from oracle import graph_ref as G, input_ref as IR
H, W = 64, 2048
P = synth.make_weights(seed=5, width=W, in_ch=cfgmod.KITTI_INPUT_CHANNELS, num_classes=2)
fr = IR.make_batch([2, 3], W=W, pad_W=W, H=H)
fr['input_data'] = np.ascontiguousarray(fr['input_data'][:, [0, 3, 4, 5, 1]])
pipe = RangeDetPipeline(P, dtype=R.RD_F16, variant="kitti", feat_size=(H, W), pad_field=(H, W), batch=2,
                        pre_nms_top_n={'veh': 50000, 'ped': 5000}, wnms_cap=8192)
outs = pipe.enqueue(fr)
frames = pipe.collect()
for ci, c in enumerate(pipe.class_names):
    sc = outs[1 + 3 * ci].cpu().numpy(); bx = outs[2 + 3 * ci].cpu().numpy()
    for b in range(2):
        got = frames[b]["per_class"][c]
        dets, rows, keep, d8 = G.postprocess(sc[b], bx[b], cls=c)
        d = np.abs(got["wnms_rows"] - rows)
        bad = np.argwhere(d > 1e-5)
        print(c, b, "K", dets.shape[0], "kept", len(keep), "bad entries", len(bad), "cols", sorted(set(bad[:, 1].tolist())))
        for r, cidx in bad[:6]:
            print("   row", r, "col", cidx, got["wnms_rows"][r, cidx], rows[r, cidx], "keep idx", keep[r])
