"""Eval post-processing and late score fusion: the data formats either side of the OFF path
(SURVEY.md section 8(f) rank 2).  Host-side numpy on [videos, 10, 101]-sized arrays, exactly
where the reference does it; the per-batch weighted fusion also exists on the GPU as
``runtime.score_fusion`` (K7) for the two-stream fused forward.

Reference (paths relative to the reference repository root):
  * per-video crop mean + weighted sum: test_rgb_off.py:138 (``np.mean(rst1,0) + 2*np.mean(rst2,0)
    + np.mean(rst3,0)``), test_flow_off.py:377 (OFF 7x7 only);
  * prediction + mean per-class accuracy: test_rgb_off.py:205-219, score_fusion.ipynb cells 7-8;
  * on-disk score format: test_rgb_off.py:236 / test_flow_off.py:450 -- ``np.savez(path,
    scores1=.., scores2=.., scores3=.., label=..)``, scores1 = OFF 7x7, scores2 = TSN backbone,
    scores3 = OFF 14x14, each [videos, crops(10), classes(101)];
  * flow -> rgb list re-ordering: score_fusion.ipynb cells 1 and 5;
  * late-fusion weights: score_fusion.ipynb cell 8 (lines 300-301) and cells 9-11.
"""
import numpy as np

# (rgb 7x7, rgb TSN, rgb 14x14, flow 7x7, flow TSN, flow 14x14)
FUSION_BEST = (1.0, 1.5, 1.6, 1.2, 0.8, 1.7)        # 95.24 %  (notebook cell 8)
FUSION_RGB_ONLY = (1.6, 1.5, 2.0, 0.0, 0.0, 0.0)    # 90.11 %  (cell 9)
FUSION_FLOW_ONLY = (0.0, 0.0, 0.0, 3.1, 1.0, 1.2)   # 89.88 %  (cell 10)
FUSION_TSN_ONLY = (0.0, 1.1, 0.0, 0.0, 1.0, 0.0)    # 94.02 %  (cell 11)
RGB_EVAL_WEIGHTS = (1.0, 2.0, 1.0)                   # test_rgb_off.py:138


def video_score(rst1, rst2, rst3, weights=RGB_EVAL_WEIGHTS):
    """One video: three [crops, classes] score blocks -> [1, classes] (test_rgb_off.py:138-141)."""
    r = sum(w * np.mean(np.asarray(x), axis=0) for w, x in zip(weights, (rst1, rst2, rst3)))
    return r.reshape(1, -1)


def save_scores(path, scores1, scores2, scores3, label):
    """test_rgb_off.py:236 format."""
    np.savez(path, scores1=np.asarray(scores1), scores2=np.asarray(scores2), scores3=np.asarray(scores3),
             label=np.asarray(label))


def load_scores(path):
    z = np.load(path)
    return z["scores1"], z["scores2"], z["scores3"], z["label"]


def reorder_index(target_list, source_list):
    """Index vector that re-orders arrays stored in ``source_list`` order into ``target_list`` order
    (score_fusion.ipynb cell 1: flow list -> rgb list; entries are the split-file lines)."""
    pos = {k: i for i, k in enumerate(source_list)}
    missing = [k for k in target_list if k not in pos]
    if missing:   # the notebook's dict.get would put None there and fail later, at the indexing in cell 5
        raise KeyError("%d entries of the target list are not in the source list, first: %r" % (len(missing), missing[0]))
    return [pos[k] for k in target_list]


def read_split_list(path):
    """A TSN split file (data/ucf101_*_split_*.txt: ``<frame dir> <num frames> <label>`` per line) as the notebook's cell 1
    reads it: one string per line without the newline.  Returns (lines, labels)."""
    with open(path, "r") as f:
        lines = [x.rstrip("\n") for x in f.readlines()]
    return lines, np.array([int(x.split()[-1]) for x in lines], dtype=np.int64)


def late_fusion(score_sets, weights):
    """score_sets: arrays [videos, crops, classes]; returns fused [videos, classes]
    = sum_i w_i * mean over crops (score_fusion.ipynb cell 8)."""
    out = None
    for w, s in zip(weights, score_sets):
        t = w * np.asarray(s, dtype=np.float32).mean(axis=1)
        out = t if out is None else out + t
    return out


def predict(fused):
    return np.argmax(fused, axis=-1)


def mean_class_accuracy(labels, preds, num_classes=None):
    """Confusion-matrix mean per-class accuracy (test_rgb_off.py:208-219), without sklearn.
    Classes that never occur as a label are skipped (sklearn's confusion_matrix only spans
    the labels present)."""
    labels = np.asarray(labels).astype(np.int64)
    preds = np.asarray(preds).astype(np.int64)
    n = int(max(labels.max(), preds.max())) + 1 if num_classes is None else num_classes
    cf = np.zeros((n, n), dtype=np.float64)
    np.add.at(cf, (labels, preds), 1.0)
    cnt = cf.sum(axis=1)
    hit = np.diag(cf)
    keep = cnt > 0
    return float(np.mean(hit[keep] / cnt[keep])), cf
