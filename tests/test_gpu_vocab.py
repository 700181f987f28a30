"""Vocabulary transform on the real GPU vs the reference's own DBoW2, including an ORBvoc-shaped tree (k=10, L=4 here: 11 110 nodes;
the shipped ORBvoc.txt has L=6 but is not part of the repository) and a batch of extracted images."""
import numpy as np
import pytest

import oracle_lib as ol
import vocab_scenes as vs
from test_emu_vocab import CONFIGS, check_vocabulary
from orb_slam3_detailed_comments_amd import synth
from orb_slam3_detailed_comments_amd.extractor import ORBextractor
from orb_slam3_detailed_comments_amd.vocabulary import ORBVocabulary

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", CONFIGS + [(10, 4, 0, 0, False, 1, (4, 2))])
def test_vocabulary_transform_gpu(hip_lib, tmp_path, cfg):
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    check_vocabulary(ex, tmp_path, cfg, seed=21, n_desc=5000)


def test_vocabulary_on_extracted_batch_gpu(hip_lib, tmp_path):
    if ol.reference_dbow2() is None:
        pytest.skip("oracle/_ref/libref_dbow2.so not built")
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    rng = np.random.default_rng(5)
    header, parent, leaf, desc, weight = vs.make_vocabulary(rng, 10, 4)
    path = tmp_path / "voc.txt"
    vs.write_text(path, header, parent, leaf, desc, weight)
    ref = ol.RefVocabulary(path)
    voc = ORBVocabulary.loadFromTextFile(ex, path)
    imgs = np.stack([synth.corner_field(752, 480, seed=40 + s) for s in range(8)])
    ex.enqueue(imgs)
    voc.transform_extracted(ex, 0, 8, 2)          # levelsup 2 of L=4 ~ ORB-SLAM3's levelsup 4 of L=6
    res = ex.fetch()
    for b in range(8):
        d = res[b][2]; n = len(d)
        r = voc.fetch(ex, b, n)
        bi, bv, fn, fs, ff = ref.transform(d, 2)
        assert np.array_equal(r.bow_id, bi) and r.bow_val.tobytes() == bv.tobytes()
        assert np.array_equal(r.fv_node, fn) and np.array_equal(r.fv_start, fs) and np.array_equal(r.fv_feat, ff)
        assert n > 1000 and len(bi) > 300


def test_vocabulary_maximum_feature_count_gpu(hip_lib, tmp_path):
    """16384 features in one call (128 KB of LDS for the sort) and the capacity error one above it."""
    from orb_slam3_detailed_comments_amd._lib import OrbxError
    if ol.reference_dbow2() is None:
        pytest.skip("oracle/_ref/libref_dbow2.so not built")
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    rng = np.random.default_rng(9)
    header, parent, leaf, desc, weight = vs.make_vocabulary(rng, 9, 3)
    path = tmp_path / "voc.txt"
    vs.write_text(path, header, parent, leaf, desc, weight)
    ref = ol.RefVocabulary(path)
    voc = ORBVocabulary.loadFromTextFile(ex, path)
    q = vs.descriptors_near(rng, desc, 16384)
    r = voc.transform(q, 1)
    bi, bv, fn, fs, ff = ref.transform(q, 1)
    assert np.array_equal(r.bow_id, bi) and r.bow_val.tobytes() == bv.tobytes()
    assert np.array_equal(r.fv_node, fn) and np.array_equal(r.fv_start, fs) and np.array_equal(r.fv_feat, ff)
    with pytest.raises(OrbxError):
        voc.transform(np.zeros((16385, 32), np.uint8), 1)


def test_vocabulary_orbvoc_scale_gpu(hip_lib, tmp_path):
    """ORBvoc.txt's shape (k = 10, L = 6, 1.1 M nodes) on the GPU: parity with the reference's DBoW2, and the time of a batch of 128 images'
    descriptors (the 36 MB of node descriptors no longer fit one XCD's L2; they live in the MALL / HBM)."""
    import time
    from test_emu_vocab import check_orbvoc_scale
    ex = ORBextractor(1200, 1.2, 8, 20, 7)
    voc, ref, desc, rng = check_orbvoc_scale(ex, tmp_path, 5000)
    q = vs.descriptors_near(rng, desc, 128 * 1213)
    voc.transform(q[:1213], 4)
    t0 = time.perf_counter()
    for b in range(0, 128 * 1213, 16384 - 16384 % 1213):
        voc.transform(q[b:b + 16384 - 16384 % 1213], 4)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter(); ref.transform(q[:1213], 4); dr = time.perf_counter() - t1
    print("ORBvoc-scale transform: %.3f ms per image of 1213 features (host arrays in, vectors out), reference DBoW2 %.3f ms per image" % (dt * 1e3 / 128, dr * 1e3))
    voc.close()
