"""Ad-hoc GPU check (not a pytest file): parity of the HIP library vs the oracle + first timings."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as ol
from orb_slam3_detailed_comments_amd import synth, _lib, ORBextractor
from orb_slam3_detailed_comments_amd import matcher as M
lib = _lib.OrbxLib(os.environ["ORBX_QUICK_LIB"]) if os.environ.get("ORBX_QUICK_LIB") else _lib.load_hip()      # ORBX_QUICK_LIB: A/B of library builds (tools/gpu_r4_ab3.sh)
print('devices', lib.L.orbx_device_count())
ok_all = True
for name, img, nf, lap in [
  ('S1 752x480', synth.corner_field(seed=0), 1200, (0,0)),
  ('S1 mono lap', synth.corner_field(seed=1), 1000, (0,1000)),
  ('S2 lowtex', synth.corner_field(seed=2, contrast_div=6.0), 1200, (0,0)),
  ('S3 sparse', synth.sparse_corners(seed=0), 1200, (0,0)),
  ('S4 noise', synth.uniform_noise(seed=0), 1200, (100,400)),
  ('512 fisheye', synth.corner_field(512,512,seed=3), 1500, (0,511)),
  ('640x480', synth.corner_field(640,480,seed=4), 1000, (0,0)),
  ('600x350 5000', synth.corner_field(600,350,seed=5), 5000, (0,1000)),
]:
    ex = ORBextractor(nf,1.2,8,20,7)
    mono,k,d = ex(img, None, lap)
    o = ol.OracleExtractor(nf); mo,ko,do = o.extract(img,lap)
    same = (mono==mo and ol.kps_equal(k,ko) and np.array_equal(d,do))
    ok_all &= same
    print(name, len(k), len(ko), 'SAME' if same else 'DIFF', flush=True)
    if not same:
        for l in range(8):
            pi = np.array_equal(ex.pyramid_level(l), o.level_image(l)); pb = np.array_equal(ex.pyramid_level(l, blurred=True), o.level_image(l, blurred=True))
            c1 = ex.debug_candidates(l); c2 = o.level_candidates(l)
            k1 = ex.debug_level_keys(l); k2 = o.level_keypoints(l)
            k2a = np.stack([k2['x']-16, k2['y']-16, k2['response']],1).astype(np.int32) if len(k2) else np.zeros((0,3),np.int32)
            print('  L',l,'pyr',pi,'blur',pb,'cand',len(c1),len(c2),np.array_equal(c1,c2),'qt',len(k1),len(k2),np.array_equal(k1,k2a))
        n=min(len(k),len(ko)); bad=[i for i in range(n) if k[i]!=ko[i] or not np.array_equal(d[i],do[i])]
        print('  nbad',len(bad),'first',bad[:3]); 
        for i in bad[:3]: print('   ',k[i],ko[i], (d[i]!=do[i]).sum())
bf, b = 458.654*0.110074, 0.110074
L,R = synth.stereo_pair(seed=0)
ex = ORBextractor(1200,1.2,8,20,7)
res = ex.extract_batch(np.stack([L,R]), (0,0))
u,dd,n = M.ComputeStereoMatches(ex, ex, bf, b, 0, 1, 1)
oL = ol.OracleExtractor(1200); oR = ol.OracleExtractor(1200)
mL,kL,dL = oL.extract(L); mR,kR,dR = oR.extract(R)
uo,do,no = ol.oracle_stereo(oL,oR,kL,dL,kR,dR,bf,b); N=len(kL)
s_ok = n[0]==no and np.array_equal(u[0,:N].view(np.uint32),uo.view(np.uint32)) and np.array_equal(dd[0,:N].view(np.uint32),do.view(np.uint32))
print('stereo', n[0], no, 'SAME' if s_ok else 'DIFF'); ok_all &= s_ok
L5,R5 = synth.stereo_pair(512,512,seed=5)
ex5 = ORBextractor(1500,1.2,8,20,7)
res = ex5.extract_batch(np.stack([L5,R5]), (0,511))
out = M.StereoFishEyeKnn(ex5, ex5, 0, 1, 1)
ref = ol.oracle_knn2(res[0][2][res[0][0]:], res[1][2][res[1][0]:]); nq=len(res[0][2])-res[0][0]
k_ok = all(np.array_equal(out[kk][0,:nq], ref[kk]) for kk in ref)
print('knn', 'SAME' if k_ok else 'DIFF'); ok_all &= k_ok
print('ALL PARITY', ok_all, flush=True)
# ---- timing ----
for B in (2, 16, 64, 128):
    imgs = []
    for s in range(B//2):
        l,r = synth.stereo_pair(seed=100+s); imgs += [l]
    for s in range(B//2):
        l,r = synth.stereo_pair(seed=100+s); imgs += [r]
    arr = np.stack(imgs)
    ex = ORBextractor(1200,1.2,8,20,7)
    dptr = ex.device_upload(arr)
    ex.profile(True, serial=(os.environ.get('ORBX_SERIAL','1')=='1'))
    for it in range(3):
        ex.enqueue(None, (0,0), device_ptr=dptr, shape=arr.shape)
        lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, B//2, B//2, bf, b))
        ex.sync()
    K=10
    t=time.time()
    for it in range(K):
        ex.enqueue(None, (0,0), device_ptr=dptr, shape=arr.shape)
        lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, B//2, B//2, bf, b))
    ex.sync(); dt=(time.time()-t)/K
    u,dd,n = M.ComputeStereoMatches(ex, ex, bf, b, 0, B//2, B//2)
    qp = np.zeros(16, np.int64); lib.L.orbx_debug_quadtree_profile(ex._h, qp.ctypes.data)
    print('  quadtree L0 phases (us): gather %.1f roots %.1f passes %.1f [last final round: sort %.1f part %.1f ndiv %.1f rebuild %.1f] select %.1f total %.1f  n=%d nodes=%d nexp=%d' % ((qp[1]-qp[0])/100,(qp[2]-qp[1])/100,(qp[3]-qp[2])/100,(qp[5]-qp[4])/100,(qp[6]-qp[5])/100,(qp[7]-qp[6])/100,(qp[8]-qp[7])/100,(qp[9]-qp[8])/100,(qp[9]-qp[0])/100,qp[10],qp[11],qp[12]))
    print('B',B,'ms/batch %.3f'%(dt*1e3),'pairs/s %.0f'%((B//2)/dt), 'stages', {k:round(v,3) for k,v in ex.stage_ms().items()}, 'matches', n[:3], flush=True)

# ---- single-pair latency: eager vs hipGraph replay ----
L, R = synth.stereo_pair(seed=100)
pair = np.stack([L, R])
for graph in (False, True):
    ex = ORBextractor(1200,1.2,8,20,7)
    ex.graph_replay(graph)
    dptr = ex.device_upload(pair)
    for it in range(5):
        ex.enqueue(None, (0,0), device_ptr=dptr, shape=pair.shape); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b)); ex.sync()
    K = 50
    t = time.time()
    for it in range(K):
        ex.enqueue(None, (0,0), device_ptr=dptr, shape=pair.shape); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b)); ex.sync()
    dt = (time.time() - t) / K
    print('single pair, graph=%s: %.3f ms per pair (sync each), %.0f pairs/s' % (graph, dt * 1e3, 1 / dt), flush=True)

# the same with the frames written into pyramid level 0 by the producer (orbx_input_buffer: no import launch), as bench.py keeps its inputs
ex = ORBextractor(1200,1.2,8,20,7)
ptr, shp, strd, istrd = ex.input_upload(pair)
for it in range(5):
    ex.enqueue(None, (0,0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b)); ex.sync()
K = 200
t = time.time()
for it in range(K):
    ex.enqueue(None, (0,0), device_ptr=ptr, shape=shp, stride=strd, image_stride=istrd); lib.check(lib.L.orbm_stereo_match(ex._h, 0, ex._h, 1, 1, bf, b)); ex.sync()
dt = (time.time() - t) / K
print('single pair, zero-copy input: %.3f ms per pair (sync each), %.0f pairs/s' % (dt * 1e3, 1 / dt), flush=True)
