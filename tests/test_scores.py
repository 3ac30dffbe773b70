"""Eval post-processing / late fusion: host module vs the literal restatement
of the reference lines, the on-disk score format, and (gpu) the K7 fusion kernel."""
import os

import numpy as np
import pytest

import offk_amd  # noqa: F401
from offk_amd import scores
from oracle import eval_oracle as eo


def make_sets(videos=12, crops=10, classes=101, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((videos, crops, classes)).astype(np.float32) for _ in range(6)]


def test_video_score_matches_reference_lines():
    s = make_sets(1)
    got = scores.video_score(s[0][0], s[1][0], s[2][0])
    np.testing.assert_allclose(got, eo.video_score(s[0][0], s[1][0], s[2][0]), rtol=1e-6)
    assert got.shape == (1, 101)


def test_late_fusion_and_accuracy():
    s = make_sets()
    fused = scores.late_fusion(s, scores.FUSION_BEST)
    ref, ref_pred = eo.fused_prediction(*s)
    np.testing.assert_allclose(fused, ref, rtol=1e-5, atol=1e-6)
    assert list(scores.predict(fused)) == ref_pred
    labels = np.array(ref_pred)
    labels[::3] = (labels[::3] + 1) % 101            # make a third of them wrong
    acc, cf = scores.mean_class_accuracy(labels, ref_pred)
    assert abs(acc - eo.mean_class_acc(labels, ref_pred)) < 1e-12
    assert cf.sum() == len(labels)


def test_score_file_format_and_reorder(tmp_path):
    s = make_sets(5)
    label = np.arange(5)
    path = os.path.join(str(tmp_path), "rgb_save_score.npz")
    scores.save_scores(path, s[0], s[1], s[2], label)
    z = np.load(path)
    assert sorted(z.files) == ["label", "scores1", "scores2", "scores3"]      # test_rgb_off.py:236
    assert z["scores1"].shape == (5, 10, 101)
    a, b, c, lab = scores.load_scores(path)
    assert np.array_equal(a, s[0]) and np.array_equal(lab, label)
    rgb_list = ["v%d 100" % i for i in (3, 1, 4, 0, 2)]
    flow_list = ["v%d 100" % i for i in range(5)]
    idx = scores.reorder_index(rgb_list, flow_list)                            # notebook cell 1
    assert idx == [3, 1, 4, 0, 2]
    assert [flow_list[i] for i in idx] == rgb_list


@pytest.mark.gpu
def test_gpu_score_fusion_kernel():
    import torch
    from offk_amd import runtime
    s = make_sets(64)
    fused, pred = runtime.score_fusion([torch.from_numpy(x).cuda() for x in s], scores.FUSION_BEST)
    ref = scores.late_fusion(s, scores.FUSION_BEST)
    np.testing.assert_allclose(fused.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    assert np.array_equal(pred.cpu().numpy(), np.argmax(fused.cpu().numpy(), axis=1))
    # crops = 1: the modality_fuse sum of Flow_OFF.py:881
    a, b, c = (torch.from_numpy(x[:, 0]).cuda() for x in s[:3])
    f2, _ = runtime.score_fusion([a, b, c], (1.0, 1.0, 1.0), want_pred=False)
    np.testing.assert_allclose(f2.cpu().numpy(), (a + b + c).cpu().numpy(), rtol=1e-6, atol=1e-6)
