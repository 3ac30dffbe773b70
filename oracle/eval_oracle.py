"""ORACLE -- test infrastructure only.  Literal numpy restatement of the reference's eval
post-processing and late fusion.

Parity status: PINNED where the reference holds vectors, as of round 3 (tests/golden/eval_split1.npz, written by
oracle/gen_eval_golden.py from the reference's real split lists and from the OUTPUT cells score_fusion.ipynb stores):

  convt_list          score_fusion.ipynb cell 1 on data/ucf101_{rgb,flow}_val_split_1.txt: the full 3783-entry index,
                      the ``None`` / ``3783`` of cell 1's output, the twelve name pairs of cell 2, ``11`` of cell 5
  mean_class_acc      test_rgb_off.py:205-219 / notebook cells 7-11: the printed cls_cnt (= label histogram of the rgb
                      list), cls_hit, the per-class accuracies (8 decimals) and the five ``Accuracy xx.xx%`` lines
                      (86.27 / 95.24 / 90.11 / 89.88 / 94.02) are reproduced from a labelling built to the printed
                      confusion diagonal
  fused_prediction    weight sets of cells 8-11 checked against the cell sources by the generator

What stays UNPINNABLE, and why: the numeric value of a fused score / 10-crop mean (video_score, fused_prediction's
arithmetic).  The reference's score files (rgb_save_score_2.npz, flow_save_score.npz; notebook cell 3) are not in the
repository, and the eval scripts that would write them cannot run here (Python-2 print statements, missing `dataset` /
`baseModel` / `models_off` modules, UCF-101 frames behind unreachable paths).  Those two functions restate the published
lines verbatim (a mean over axis 0 and a weighted sum) and are checked against hand-computed cases only.

  video_score        test_rgb_off.py:138-141
  convt_list         score_fusion.ipynb cell 1
  fused_prediction   score_fusion.ipynb cells 8-11 (lines 300-303)
  mean_class_acc     test_rgb_off.py:205-219 (sklearn.confusion_matrix replaced by a loop)
"""
import numpy as np


def video_score(rst1, rst2, rst3):
    rst = np.mean(rst1, axis=0) + 2 * np.mean(rst2, axis=0) + np.mean(rst3, axis=0)
    return rst.reshape(1, 101)


def convt_list(rgb_list, flow_list):
    """score_fusion.ipynb cell 1: position in the flow list of every rgb-list entry (None when absent, dict.get)."""
    flow_dict = {k: v for v, k in enumerate(flow_list)}
    return [flow_dict.get(k) for k in rgb_list]


def fused_prediction(rgb_rst1, rgb_rst2, rgb_rst3, flow_rst1, flow_rst2, flow_rst3,
                     w=(1., 1.5, 1.6, 1.2, .8, 1.7)):
    video_pred = [
        w[0] * x.mean(axis=0) + w[1] * y.mean(axis=0) + w[2] * z.mean(axis=0) +
        w[3] * m.mean(axis=0) + w[4] * n.mean(axis=0) + w[5] * q.mean(axis=0)
        for x, y, z, m, n, q in zip(rgb_rst1, rgb_rst2, rgb_rst3, flow_rst1, flow_rst2, flow_rst3)]
    return np.stack(video_pred), [np.argmax(x) for x in video_pred]


def mean_class_acc(video_labels, video_pred):
    classes = sorted(set(int(v) for v in video_labels) | set(int(v) for v in video_pred))
    idx = {c: i for i, c in enumerate(classes)}
    cf = np.zeros((len(classes), len(classes)))
    for t, p in zip(video_labels, video_pred):
        cf[idx[int(t)], idx[int(p)]] += 1
    cls_cnt = cf.sum(axis=1)
    cls_hit = np.diag(cf)
    keep = cls_cnt > 0
    return np.mean(cls_hit[keep] / cls_cnt[keep])


def class_table(video_labels, video_pred, num_classes):
    """(cls_hit, cls_cnt, cls_acc) as notebook cells 7-11 print them (every class occurs in the split-1 labels)."""
    cf = np.zeros((num_classes, num_classes))
    for t, p in zip(video_labels, video_pred):
        cf[int(t), int(p)] += 1
    cls_cnt = cf.sum(axis=1)
    cls_hit = np.diag(cf)
    return cls_hit, cls_cnt, cls_hit / cls_cnt
