"""The reference's evaluation loop (tools/test.py:84-238) on the HIP path, without MXNet: roidb records -> device input
transform -> forward + post-processing -> `output_dict` / `annotation_dict` -> the pickle tools/create_prediction_bin_3d.py
(rangedet_amd.export) reads.

    python -m rangedet_amd.evaluate --roidb 'data/validation/*.roidb' --prefix experiments/<cfg>/checkpoint --epoch 18 \
        [--out experiments/<cfg>/checkpoint_output_dict_18e.pkl] [--bin-dir <dir> --config-name <cfg>] [--batch 8]
    python -m rangedet_amd.evaluate --synthetic 16 --random-weights --out /tmp/out.pkl         (self-contained dry run)

A roidb record is the dict datasets/create_range_image_roidb.py writes: `pc_url` (npz with the arrays `range_image`,
`pc_vehicle_frame`, `inclination`, `azimuth`, ...: create_range_image_roidb.py:119-124,164; read by LoadRecord,
rangedet/core/input.py:23-38), `gt_bbox_imu`, `gt_class`.  Frames are batched; each frame's
result is what tools/test.py:200-232 computes for it: `det_xyzlwhyaws[TYPE_VEHICLE]` (M,8) + `meta_info`.
"""
import argparse
import glob
import pickle as pkl

import numpy as np

mapping = {'veh': 'TYPE_VEHICLE', 'ped': 'TYPE_PEDESTRIAN', 'cyc': 'TYPE_CYCLIST'}   # tools/test.py:170


def load_record(rec):
    """LoadRecord (rangedet/core/input.py:14-42) for one roidb entry: the raw arrays the device transform needs."""
    if 'range_image' in rec:                                   # already loaded (synthetic records)
        return rec
    with np.load(rec['pc_url']) as z:                          # same keys and float32 casts as LoadRecord.apply (:29-36)
        return dict(range_image=z['range_image'].astype(np.float32), pc_vehicle_frame=z['pc_vehicle_frame'].astype(np.float32),
                    inclination=z['inclination'].astype(np.float32))


def meta_info(rec, rid):
    url = rec.get('pc_url')
    if not url:
        return {'name': 'synthetic', 'timestamp_micros': int(rid)}
    name = url.split('/')[-2].replace('segment-', '').replace('_with_camera_labels', '')   # tools/test.py:226-228
    return {'name': name, 'timestamp_micros': int(url.split('/')[-1][:-4])}


def run(roidb, params, batch=8, variant='veh', wnms=True, progress=None, pre_nms_top_n=50000):
    """-> (annotation_dict, output_dict) exactly as tools/test.py:166-233 builds them (frames without detections are absent)."""
    from .input_transform import DeviceInputTransform
    from .pipeline import RangeDetPipeline
    H, W = np.asarray(load_record(roidb[0])['range_image']).shape[:2]
    Wp = -(-W // 32) * 32
    pipe = RangeDetPipeline(params, batch=batch, feat_size=(H, W), pad_field=(H, Wp), variant=variant, wnms=wnms,
                            pre_nms_top_n=pre_nms_top_n)
    to_inputs = DeviceInputTransform(pad_hw=(H, Wp), lib=pipe.lib, alloc=pipe.alloc)
    cls = mapping[variant if variant in mapping else 'veh']
    output_dict, annotation_dict = {}, {}
    for i0 in range(0, len(roidb), batch):
        chunk = roidb[i0:i0 + batch]
        recs = [load_record(r) for r in chunk]
        recs += [recs[-1]] * (batch - len(recs))               # the last batch is padded with its last frame
        res = pipe.run(to_inputs(recs))
        frames = res['frames'] if batch > 1 else [res]
        for j, rec in enumerate(chunk):
            rid = rec.get('rec_id', i0 + j)
            det = frames[j]['det_xyzlwhyaws']
            if det.shape[0] == 0:
                continue                                       # tools/test.py:204-205, 222-223
            output_dict[rid] = {'det_xyzlwhyaws': {cls: det}, 'meta_info': meta_info(rec, rid)}
            annotation_dict[rid] = rec.get('gt_bbox_imu')
        if progress:
            progress(min(i0 + batch, len(roidb)), len(roidb))
    return annotation_dict, output_dict


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--roidb', help="glob of .roidb pickles (lists of records)")
    ap.add_argument('--synthetic', type=int, default=0, help="use N synthetic records instead of --roidb")
    ap.add_argument('--prefix'), ap.add_argument('--epoch', type=int)
    ap.add_argument('--random-weights', action='store_true')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--out', required=True, help="pickle path (annotation_dict, then output_dict: tools/test.py:235-237)")
    ap.add_argument('--bin-dir'), ap.add_argument('--config-name', default='rangedet_veh_wo_aug_4_18e')
    ap.add_argument('--nms3d', action='store_true', help="RpnParam.wnms = False: contrib.NMS3D instead of the weighted NMS")
    a = ap.parse_args(argv)
    from . import synth
    if a.synthetic:
        roidb = [dict(synth.raw_record(i), rec_id=i) for i in range(a.synthetic)]
    else:
        roidb = []
        for s in sorted(glob.glob(a.roidb)):
            roidb += pkl.load(open(s, 'rb'), encoding='latin1')
        for i, r in enumerate(roidb):
            r['rec_id'] = i                                     # tools/test.py:113-114
    if a.random_weights:
        params = synth.make_weights(seed=18)
    else:
        from .load_model import load_params
        params = load_params(a.prefix, a.epoch)
    ann, out = run(roidb, params, batch=a.batch, wnms=not a.nms3d,
                   progress=lambda d, n: print('%d of %d records' % (d, n), flush=True))
    with open(a.out, 'wb') as fw:
        pkl.dump(ann, fw)
        pkl.dump(out, fw)
    print('%d frames with detections of %d -> %s' % (len(out), len(roidb), a.out))
    if a.bin_dir:
        from . import export
        export.main(a.out, a.config_name, a.bin_dir)


if __name__ == '__main__':
    main()
