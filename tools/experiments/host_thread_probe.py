"""Which threads of a rank burn host CPU during the headline loop, and what makes them stop: runs a short loop of the bench's shape (4 handles, 64 pairs)
and prints per-thread CPU seconds with each thread's kernel wait channel and current syscall, under the environment it was started with."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from orb_slam3_detailed_comments_amd import ORBextractor, load_hip, synth

lib = load_hip()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if mode:
    lib.check(lib.L.orbx_set_host_wait(0, mode))
P = 64
L, R = zip(*[synth.stereo_pair(752, 480, seed=i) for i in range(4)])
batch = np.stack([L[i % 4] for i in range(P)] + [R[i % 4] for i in range(P)])
hs = [ORBextractor(1200, 1.2, 8, 20, 7, device_id=0, lib=lib) for _ in range(4)]
ins = [h.input_upload(batch) for h in hs]
def tcpu():
    out = {}
    tck = float(os.sysconf("SC_CLK_TCK"))
    for t in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % t).read(); f = st[st.rindex(")") + 2:].split()
            out[int(t)] = (int(f[11]) + int(f[12])) / tck
        except Exception:
            pass
    return out
BF, B = 458.654 * 0.110074, 0.110074
def step(i):
    h = hs[i]; lp, shape, stride, istride = ins[i]
    h.enqueue(None, (0, 0), device_ptr=lp, shape=shape, stride=stride, image_stride=istride)
    lib.check(lib.L.orbm_stereo_match(h._h, 0, h._h, P, P, BF, B))
for i in range(4): step(i)
for h in hs: h.sync()
a = tcpu(); t0 = time.time(); n = 0
while time.time() - t0 < 3.0:
    i = n % 4; hs[i].sync(); step(i); n += 1
for h in hs: h.sync()
dt = time.time() - t0; b = tcpu()
print("mode", mode, "env", {k: v for k, v in os.environ.items() if k.startswith(("HSA", "ROC", "HIP", "GPU_", "AMD"))}, "steps/s", round(n / dt, 1), "main tid", os.getpid())
for t in sorted(b, key=lambda t: -(b[t] - a.get(t, 0)))[:5]:
    rd = lambda f: (open("/proc/self/task/%d/%s" % (t, f)).read().strip()[:80] if os.path.exists("/proc/self/task/%d/%s" % (t, f)) else "?")
    try:
        info = (rd("comm"), rd("wchan"), rd("syscall"))
    except Exception as e:
        info = repr(e)
    print("  tid", t, "cpu_s", round(b[t] - a.get(t, 0), 2), "of", round(dt, 2), info)
