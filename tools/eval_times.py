#!/usr/bin/env python3
"""Time of one evaluation batch (parse_predictions + parse_groundtruths + AP over the batch):
the GPU path (votenet/ap_helper.py) against the numpy restatement of the reference's per-box
host code (oracle/eval_oracle.py; the reference itself adds a Delaunay triangulation per box
and one `.cpu()` per scalar).  Usage: python tools/eval_times.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backtoreality_amd.votenet import ap_helper, config, synthetic  # noqa: E402
from oracle import eval_oracle as eo  # noqa: E402

dev = torch.device("cuda:0")
cfg = config.scannet_md40()
for name, cd in (("train settings", {'remove_empty_box': False, 'use_3d_nms': True,
                                     'nms_iou': 0.25, 'use_old_type_nms': False,
                                     'cls_nms': True, 'per_class_proposal': True,
                                     'conf_thresh': 0.05}),
                 ("+ remove_empty_box", {'remove_empty_box': True, 'use_3d_nms': True,
                                         'nms_iou': 0.25, 'use_old_type_nms': False,
                                         'cls_nms': True, 'per_class_proposal': True,
                                         'conf_thresh': 0.05})):
    cd = dict(cd, dataset_config=cfg)
    case = synthetic.make_eval_case(1, 8, 40000, cfg, num_proposal=256, device=dev)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred = ap_helper.parse_predictions(case, cd)
        gt = ap_helper.parse_groundtruths(case, cd)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        calc = ap_helper.APCalculator(0.25)
        calc.step(pred, gt)
        m = calc.compute_metrics()
        t2 = time.perf_counter()
    host = {k: v.cpu().numpy() for k, v in case.items() if torch.is_tensor(v)}
    t3 = time.perf_counter()
    pred_o, _, _ = eo.parse_predictions(host, cd)
    gt_o = eo.parse_groundtruths(host, cd)
    t4 = time.perf_counter()
    mo = eo.metrics(pred_o, gt_o, 0.25)
    t5 = time.perf_counter()
    print("%-20s 8 scenes x 256 proposals x 40000 points: GPU path parse %.1f ms + AP %.1f ms"
          " (mAP %.4f) | numpy port parse %.0f ms + AP %.0f ms (mAP %.4f)" % (
              name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, m['mAP'], (t4 - t3) * 1e3,
              (t5 - t4) * 1e3, mo['mAP']), flush=True)
