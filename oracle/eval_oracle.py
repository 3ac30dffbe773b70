"""ORACLE -- test infrastructure only.  Literal numpy restatement of the reference's eval
post-processing and late fusion.

Parity status: UNPINNED.  The reference eval scripts cannot run here (Python-2 syntax, missing
`dataset` / `baseModel` / `models_off` modules, dataset and score files behind unreachable links)
and hold no golden vectors for this step, so these functions restate the published lines verbatim
and are only checked against hand-computed cases.

  video_score        test_rgb_off.py:138-141
  fused_prediction   score_fusion.ipynb cell 8 (lines 300-303)
  mean_class_acc     test_rgb_off.py:205-219 (sklearn.confusion_matrix replaced by a loop)
"""
import numpy as np


def video_score(rst1, rst2, rst3):
    rst = np.mean(rst1, axis=0) + 2 * np.mean(rst2, axis=0) + np.mean(rst3, axis=0)
    return rst.reshape(1, 101)


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
