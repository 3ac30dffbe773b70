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


# ---- pinned to the vectors the reference holds (tests/golden/eval_split1.npz, oracle/gen_eval_golden.py) -----------
NOTEBOOK_WEIGHTS = {8: scores.FUSION_BEST, 9: scores.FUSION_RGB_ONLY, 10: scores.FUSION_FLOW_ONLY, 11: scores.FUSION_TSN_ONLY}


def _split1(golden_dir):
    import hashlib
    z = np.load(os.path.join(golden_dir, "eval_split1.npz"))
    lists = {}
    for name in ("rgb", "flow"):
        lines = ["%s/%s %d %d" % (z["list_prefix"], n, f, l)
                 for n, f, l in zip(z[name + "_names"], z[name + "_frames"], z[name + "_label"])]
        lists[name] = lines
    # the fixture's lists are byte-for-byte the reference's data/ucf101_{rgb,flow}_val_split_1.txt
    assert [hashlib.sha256("\n".join(lists[k]).encode()).hexdigest() for k in ("rgb", "flow")] == list(z["lists_sha256"])
    return z, lists


def test_reorder_index_pinned_to_reference_split_lists(golden_dir, tmp_path):
    """score_fusion.ipynb cell 1 on the reference's real split-1 lists: full index vector, cell 1 / 2 / 5 outputs."""
    z, lists = _split1(golden_dir)
    want = z["convt_list"]
    assert len(want) == 3783 and sorted(want) == list(range(3783))            # cell 1 prints 3783; a permutation
    assert scores.reorder_index(lists["rgb"], lists["flow"]) == list(want)
    assert eo.convt_list(lists["rgb"], lists["flow"]) == list(want)
    # cell 1's probe key lacks the label column: dict.get -> None (printed "None"); the product raises instead
    assert eo.convt_list(["/home/zhufl/Data/UCF101_Frame/v_TaiChi_g04_c04 173"], lists["flow"]) == [None]
    with pytest.raises(KeyError):
        scores.reorder_index(["/home/zhufl/Data/UCF101_Frame/v_TaiChi_g04_c04 173"], lists["flow"])
    # cell 2: entry idx of the rgb list and entry convt[idx] of the flow list are the same video (first twelve printed)
    for idx, name in enumerate(z["first_names"]):
        assert lists["flow"][want[idx]] == lists["rgb"][idx] and name in lists["rgb"][idx]
    # cell 5: flow labels re-ordered are the rgb labels; flow_label_convt[0] printed as 11
    assert z["flow_label"][want[0]] == 11
    assert np.array_equal(z["flow_label"][want], z["rgb_label"])
    # the same through the file reader
    for k in ("rgb", "flow"):
        with open(os.path.join(str(tmp_path), k + ".txt"), "w") as f:
            f.write("\n".join(lists[k]) + "\n")
    rl, rlab = scores.read_split_list(os.path.join(str(tmp_path), "rgb.txt"))
    fl, _ = scores.read_split_list(os.path.join(str(tmp_path), "flow.txt"))
    assert scores.reorder_index(rl, fl) == list(want) and np.array_equal(rlab, z["rgb_label"])


@pytest.mark.parametrize("cell", [7, 8, 9, 10, 11])
def test_mean_class_accuracy_pinned_to_notebook_outputs(golden_dir, cell):
    """Notebook cells 7-11 print cls_hit, cls_cnt, the per-class accuracies and 'Accuracy xx.xx%'.  cls_cnt is the label
    histogram of the rgb list; a prediction vector with the printed confusion diagonal must reproduce every printed
    number through scores.mean_class_accuracy and through the oracle."""
    z, _ = _split1(golden_dir)
    labels = z["rgb_label"].astype(np.int64)
    hit, cnt, acc_print = z["cls_hit_%d" % cell], z["cls_cnt_%d" % cell], z["cls_acc_%d" % cell]
    assert np.array_equal(np.bincount(labels, minlength=101), cnt)
    preds = labels.copy()
    for c in range(101):                      # per class: the first cnt - hit videos are misclassified as class c + 1
        wrong = np.flatnonzero(labels == c)[: int(cnt[c] - hit[c])]
        preds[wrong] = (c + 1) % 101
    acc, cf = scores.mean_class_accuracy(labels, preds, num_classes=101)
    assert np.array_equal(np.diag(cf), hit) and np.array_equal(cf.sum(axis=1), cnt)
    np.testing.assert_allclose(np.diag(cf) / cf.sum(axis=1), acc_print, atol=5.1e-9)      # printed with 8 decimals
    assert "%.02f" % (acc * 100) == "%.02f" % float(z["accuracy_pct_%d" % cell])           # 'Accuracy {:.02f}%'
    assert abs(acc - eo.mean_class_acc(labels, preds)) < 1e-12
    h2, c2, a2 = eo.class_table(labels, preds, 101)
    assert np.array_equal(h2, hit) and np.array_equal(c2, cnt)
    np.testing.assert_allclose(a2, acc_print, atol=5.1e-9)


def test_fusion_weight_sets_are_the_notebooks(golden_dir):
    z, _ = _split1(golden_dir)
    for cell, w in NOTEBOOK_WEIGHTS.items():
        assert np.allclose(w, z["weights_%d" % cell]), cell


@pytest.mark.gpu
def test_gpu_score_fusion_kernel(golden_dir):
    """K7 against the ORACLE (eval_oracle.fused_prediction = notebook cells 8-11 verbatim), 10 crops, all four weight sets,
    on a 64-video batch and on the split-1 size (3783 videos, scores re-ordered with the pinned convt_list as cell 5 does)."""
    import torch
    from offk_amd import runtime
    z = np.load(os.path.join(golden_dir, "eval_split1.npz"))
    convt = z["convt_list"]
    for videos, reorder in ((64, False), (3783, True)):
        s = make_sets(videos, seed=videos)
        if reorder:      # flow sets arrive in flow-list order; cell 5 re-orders them to the rgb list
            s_dev = s[:3] + [x[convt] for x in s[3:]]
        else:
            s_dev = s
        dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in s_dev]
        for cell, w in NOTEBOOK_WEIGHTS.items():
            fused, pred = runtime.score_fusion(dev, w)
            ref, ref_pred = eo.fused_prediction(*s_dev, w=tuple(z["weights_%d" % cell]))
            np.testing.assert_allclose(fused.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
            got_pred = pred.cpu().numpy()
            assert np.array_equal(got_pred, np.argmax(fused.cpu().numpy(), axis=1))
            # argmax agrees with the oracle's except where the two best fused scores tie within fp32 rounding
            diff = np.flatnonzero(got_pred != np.array(ref_pred))
            for v in diff:
                top = np.sort(ref[v])[-2:]
                assert top[1] - top[0] < 1e-5, (cell, v)
            host = scores.late_fusion(s_dev, w)
            np.testing.assert_allclose(host, ref, rtol=1e-5, atol=1e-5)
    s = make_sets(64)
    # crops = 1: the modality_fuse sum of Flow_OFF.py:881
    a, b, c = (torch.from_numpy(x[:, 0]).cuda() for x in s[:3])
    f2, _ = runtime.score_fusion([a, b, c], (1.0, 1.0, 1.0), want_pred=False)
    np.testing.assert_allclose(f2.cpu().numpy(), (a + b + c).cpu().numpy(), rtol=1e-6, atol=1e-6)
