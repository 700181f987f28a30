"""Latency of the extractors Tracking constructs for the monocular sensors, one image per call (Tracking's rhythm): mpIniORBextractor = ORBextractor(5 * nFeatures)
(first quadtree levels in the node pool) beside mpORBextractorLeft = ORBextractor(nFeatures) (LDS form), and the pool form forced onto the ordinary extractor
(what the node pool costs against LDS on the same trees).  Prints one JSON object.  GPU box only."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from orb_slam3_detailed_comments_amd import ORBextractor, synth


def ms_per_call(ex, img, n=60, warm=10):
    for _ in range(warm):
        ex(img)
    t = time.perf_counter()
    for _ in range(n):
        ex(img)
    return (time.perf_counter() - t) / n * 1e3


def main():
    out = {}
    for (w, h, nf) in ((1241, 376, 2000), (512, 512, 1500), (752, 480, 1000)):
        for kind, img in (("corner_field", synth.corner_field(w, h, seed=40, nrect=int(3000 * w * h / (752 * 480)))), ("natural", synth.natural(w, h, seed=41))):
            row = {}
            ex = ORBextractor(nf, 1.2, 8, 20, 7, device_id=0)
            row["nFeatures_ms"] = round(ms_per_call(ex, img), 4)
            ex.debug_quadtree_lds_nodes(0)
            row["nFeatures_forced_pool_ms"] = round(ms_per_call(ex, img), 4)
            ini = ORBextractor(5 * nf, 1.2, 8, 20, 7, device_id=0)
            row["5x_nFeatures_ms"] = round(ms_per_call(ini, img), 4)
            row["5x_pool_levels"] = ini.debug_quadtree_pool_levels()
            row["5x_keypoints"] = int(len(ini(img)[1]))
            ini.profile(True, serial=True); ini(img); ini(img)
            row["5x_stage_ms_serial"] = {k: round(float(v), 4) for k, v in ini.stage_ms().items()} if isinstance(ini.stage_ms(), dict) else [round(float(v), 4) for v in ini.stage_ms()]
            out["%dx%d n=%d %s" % (w, h, nf, kind)] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
