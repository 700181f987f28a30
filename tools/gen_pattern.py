"""Regenerates orb_slam3_detailed_comments_amd/csrc/brief_pattern.inc from the reference's data table
(src/ORBextractor.cc:206-464).  Only runs where /root/reference exists (this container)."""
import re, sys
src = open('/root/reference/src/ORBextractor.cc', encoding='utf-8', errors='ignore').read().split('\n')
rows = []
for ln in src[205:464]:
    m = re.match(r'\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*,?', ln)
    if m:
        rows.append([int(g) for g in m.groups()])
assert len(rows) == 256
flat = [v for r in rows for v in r]
for i in range(0, 1024, 32):
    sys.stdout.write('    ' + ','.join(str(v) for v in flat[i:i + 32]) + ',\n')
