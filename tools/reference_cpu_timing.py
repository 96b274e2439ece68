"""Times the REFERENCE classes themselves (build container only: imports /root/reference) on bench.py's workload -- BoT-SORT +
OSNet-x0.25, 64 dets x 256 tracks, 1080p, botsort.yaml defaults, use_cmc=False, ReID inside update -- under the cv2 / lap stand-ins
of oracle/ref_harness.py (cv2.resize = the restated fixed-point bilinear, lap.lapjv = oracle/lapjv.c).  This is the "reference CPU path
on host cores" number DESIGN.md section 6 quotes next to bench.py's cpu_baseline (which times the oracle port on the GPU box, where
/root/reference does not exist).  Usage: python tools/reference_cpu_timing.py [frames]"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import torch

    from boxmot_amd.reid_weights import reference_init_state_dict
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    from oracle import ref_harness
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    BotSort = ref_harness.load_botsort()
    osnet = ref_harness.load_osnet_module()
    model = osnet.osnet_x0_25(num_classes=1, pretrained=False).eval()
    model.load_state_dict(reference_init_state_dict("osnet_x0_25", seed=0), strict=False)
    reid = ref_harness.RefReID(model)
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    trk = BotSort(reid_model=None, use_cmc=False, **kw)
    trk.model = reid                      # the constructor builds its ReID through the registry (not importable offline)
    sc = Scenario(64, 256, stream=0)
    for t in range(3):                    # confirmation frames (256 detections each)
        trk.update(sc.frame(t, with_embs=False)[0], sc.image)
    t0 = time.perf_counter()
    for t in range(3, 3 + n):
        out = trk.update(sc.frame(t, with_embs=False)[0], sc.image)
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "reference BotSort + reference OSNet-x0.25 (torch CPU) under cv2/lap stand-ins, 64 dets x 256 tracks, 1080p",
                      "frames": n, "frames_per_s": n / dt, "ms_per_frame": 1e3 * dt / n, "torch_threads": torch.get_num_threads(),
                      "host_logical_cores": os.cpu_count(), "rows_last": int(len(out))}))


if __name__ == "__main__":
    main()
