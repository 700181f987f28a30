"""reads the FASTCELL lines of tools/experiments/fastspan_probe.py (last run) and prints when the cells of image 0 start and end inside the launch"""
import sys
runs = open(sys.argv[1]).read().split("--- run")[1:]
rows = [l.split()[1:] for l in runs[-1].splitlines() if l.startswith("FASTCELL")]      # (device printf is flushed late: the last block may hold several runs)
rows = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4])) + tuple(float(x) for x in r[5:11]) + (int(r[11]),) for r in rows]
t0 = min(r[3] for r in rows)
print("%d cells; first start 0, last start %.1f us, first end %.1f, last end %.1f us" % (len(rows), (max(r[3] for r in rows) - t0) / 100, (min(r[4] for r in rows) - t0) / 100, (max(r[4] for r in rows) - t0) / 100))
for lv in range(8):
    rr = [r for r in rows if r[0] == lv]
    if rr:
        d = [(r[4] - r[3]) / 100 for r in rr]
        print("level %d: %3d cells, start %.1f..%.1f, duration min %.1f mean %.1f max %.1f us; phases (mean) load %.2f zero %.2f A %.2f B %.2f C %.2f D %.2f" % ((lv, len(rr), (min(r[3] for r in rr) - t0) / 100, (max(r[3] for r in rr) - t0) / 100, min(d), sum(d) / len(d), max(d)) + tuple(sum(r[5 + k] for r in rr) / len(rr) for k in range(6))))
worst = sorted(rows, key=lambda r: r[3] - r[4])[:5]
for r in worst:
    print("  longest: level %d x0 %d y0 %d start %.1f dur %.1f: load %.2f zero %.2f A %.2f B %.2f C %.2f D %.2f corners %d" % ((r[0], r[1], r[2], (r[3] - t0) / 100, (r[4] - r[3]) / 100) + r[5:11] + (r[11],)))
