"""Writes tests/golden/mono_settings.json: the extractor parameters of every monocular / monocular-inertial settings file the reference ships
(Examples*/Monocular*/**/*.yaml).  Tracking builds TWO extractors from each (src/Tracking.cc:631-635, :1328-1332): mpORBextractorLeft with nFeatures and,
for the monocular sensors, mpIniORBextractor with 5 * nFeatures - the instance whose level-0 quadtree outgrows the LDS.  tests/test_mono_init.py runs every
distinct (image size, 5 * nFeatures) through the drop-in.  Data only: names and numbers, no reference text.  Run in the build container (needs /root/reference)."""
import glob
import json
import os
import re

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def value(text, keys, cast=float):
    for k in keys:
        m = re.search(r"^\s*" + re.escape(k) + r"\s*:\s*([-0-9.eE]+)", text, re.M)
        if m:
            return cast(float(m.group(1)))
    return None


def main():
    rows = []
    for f in sorted(glob.glob(os.path.join(REF, "Examples*", "Monocular*", "**", "*.yaml"), recursive=True)):
        t = open(f, errors="replace").read()
        w, h = value(t, ["Camera.width", "Camera1.width"], int), value(t, ["Camera.height", "Camera1.height"], int)
        nw, nh = value(t, ["Camera.newWidth"], int), value(t, ["Camera.newHeight"], int)
        n = value(t, ["ORBextractor.nFeatures"], int)
        if w is None or n is None:
            continue
        rows.append({"file": os.path.relpath(f, REF), "width": nw or w, "height": nh or h, "nFeatures": n,
                     "scaleFactor": value(t, ["ORBextractor.scaleFactor"]), "nLevels": value(t, ["ORBextractor.nLevels"], int),
                     "iniThFAST": value(t, ["ORBextractor.iniThFAST"], int), "minThFAST": value(t, ["ORBextractor.minThFAST"], int)})
    out = os.path.join(ROOT, "tests", "golden", "mono_settings.json")
    json.dump(rows, open(out, "w"), indent=0)
    distinct = sorted({(r["width"], r["height"], 5 * r["nFeatures"]) for r in rows})
    print("%d settings files, %d distinct (size, 5 x nFeatures): %s" % (len(rows), len(distinct), distinct))


if __name__ == "__main__":
    main()
