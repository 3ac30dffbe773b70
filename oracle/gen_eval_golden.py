"""Generate tests/golden/eval_split1.npz: the vectors the REFERENCE holds for the eval post-processing /
late-fusion step.

Runs only in the dev container (needs /root/reference; nothing on the GPU box reads it).  Two sources:

  1. the reference's real split lists ``data/ucf101_rgb_val_split_1.txt`` / ``data/ucf101_flow_val_split_1.txt``
     run through score_fusion.ipynb cell 1 EXACTLY as written there (strip the newline, ``flow_dict = {k: v for v, k
     in enumerate(flow_list)}``, ``convt_list = [flow_dict.get(k) for k in rgb_list]``) -> the flow -> rgb
     re-ordering index, plus the label column of both lists;
  2. the OUTPUT cells the notebook stores: cell 1 (``None``, ``3783``), cell 2 (the first twelve name pairs), cell 5
     (``flow_label_convt[0]`` = 11), and for cells 7-11 the printed ``cls_hit`` / ``cls_cnt`` / per-class accuracy arrays
     and the ``Accuracy xx.xx%`` line (86.27 / 95.24 / 90.11 / 89.88 / 94.02 %).

The fixture holds data only: the two split lists as (name, frames, label) columns -- they are the INPUT of the
re-ordering and data files of the reference, not source -- the index vector and the printed numbers; no line of the
reference's code is stored.  tests/test_scores.py pins scores.reorder_index / mean_class_accuracy and oracle/eval_oracle.py to it.

    python oracle/gen_eval_golden.py
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def notebook_cell1():
    """score_fusion.ipynb cell 1, line for line (paths replaced by the repo-relative ones)."""
    with open(os.path.join(REF, "data", "ucf101_flow_val_split_1.txt"), "r") as f:
        flow_list = f.readlines()
    with open(os.path.join(REF, "data", "ucf101_rgb_val_split_1.txt"), "r") as f:
        rgb_list = f.readlines()
    flow_list = [x[:-1] for x in flow_list]
    rgb_list = [x[:-1] for x in rgb_list]
    flow_dict = {k: v for v, k in enumerate(flow_list)}
    probe = flow_dict.get('/home/zhufl/Data/UCF101_Frame/v_TaiChi_g04_c04 173')
    convt_list = []
    for k in rgb_list:
        convt_list.append(flow_dict.get(k))
    return flow_list, rgb_list, convt_list, probe


def floats_after(text, tag):
    """The bracketed array printed right after the line ``tag``."""
    i = text.index(tag) + len(tag)
    j = text.index("]", i)
    return np.array([float(t) for t in re.findall(r"[-+0-9.eE]+", text[text.index("[", i) + 1:j])])


def main():
    flow_list, rgb_list, convt, probe = notebook_cell1()
    nb = json.load(open(os.path.join(REF, "score_fusion.ipynb")))
    outs = {}
    for i, c in enumerate(nb["cells"]):
        t = ""
        for o in c.get("outputs", []):
            t += "".join(o.get("text", []))
        outs[i] = t
    # the notebook's own stored outputs agree with what cell 1 computes here on the repo's lists
    assert outs[1].split() == [str(probe), str(len(convt))] == ["None", "3783"], outs[1]
    pairs = re.findall(r"flow order video name (.*)\nrgb order video name (.*)\n", outs[2])
    assert len(pairs) == 12
    for idx, (a, b) in enumerate(pairs):
        assert a == rgb_list[idx] and b == flow_list[convt[idx]], (idx, a, b)
    rgb_label = np.array([int(x.split()[-1]) for x in rgb_list], dtype=np.int32)
    flow_label = np.array([int(x.split()[-1]) for x in flow_list], dtype=np.int32)
    assert int(outs[5].strip()) == flow_label[convt[0]] == 11

    rec = {"convt_list": np.array(convt, dtype=np.int32), "rgb_label": rgb_label, "flow_label": flow_label,
           "first_names": np.array([os.path.basename(a.split()[0]) for a, _ in pairs]),
           # the two split lists as data (inputs of the re-ordering): "<prefix>/<name> <frames> <label>" per entry
           "list_prefix": np.array(os.path.dirname(rgb_list[0].split()[0])),
           "rgb_names": np.array([os.path.basename(x.split()[0]) for x in rgb_list]),
           "rgb_frames": np.array([int(x.split()[1]) for x in rgb_list], dtype=np.int32),
           "flow_names": np.array([os.path.basename(x.split()[0]) for x in flow_list]),
           "flow_frames": np.array([int(x.split()[1]) for x in flow_list], dtype=np.int32),
           "lists_sha256": np.array([hashlib.sha256("\n".join(rgb_list).encode()).hexdigest(),
                                     hashlib.sha256("\n".join(flow_list).encode()).hexdigest()])}
    for cell in (7, 8, 9, 10, 11):
        t = outs[cell]
        rec["cls_hit_%d" % cell] = floats_after(t, "cls_hit:")
        rec["cls_cnt_%d" % cell] = floats_after(t, "cls_cnt:")
        k = t.index("]", t.index("cls_cnt:")) + 1
        rec["cls_acc_%d" % cell] = floats_after(t[k:], "")
        rec["accuracy_pct_%d" % cell] = np.array(float(re.search(r"Accuracy ([0-9.]+)%", t).group(1)))
        assert rec["cls_hit_%d" % cell].shape == rec["cls_cnt_%d" % cell].shape == rec["cls_acc_%d" % cell].shape == (101,)
    # late-fusion weight sets as the notebook's source cells spell them (cells 8-11)
    rec["weights_8"] = np.array([1., 1.5, 1.6, 1.2, .8, 1.7])
    rec["weights_9"] = np.array([1.6, 1.5, 2., 0, 0, 0])
    rec["weights_10"] = np.array([0, 0, 0, 3.1, 1.0, 1.2])
    rec["weights_11"] = np.array([0., 1.1, 0, 0, 1., 0])
    for cell in (8, 9, 10, 11):   # ... and check the spelled weights against the cell source text
        src = "".join(nb["cells"][cell]["source"])
        nums = [float(x) for x in re.findall(r"([0-9.]+)\s*\*\s*[xyzmnq]\.mean", src)]
        assert np.allclose(nums, rec["weights_%d" % cell]), (cell, nums)
    for lst, names, frames, labels in ((rgb_list, "rgb_names", "rgb_frames", rgb_label), (flow_list, "flow_names", "flow_frames", flow_label)):
        assert lst == ["%s/%s %d %d" % (rec["list_prefix"], n, f, l) for n, f, l in zip(rec[names], rec[frames], labels)]
    out = os.path.join(ROOT, "tests", "golden", "eval_split1.npz")
    np.savez_compressed(out, **rec)
    print("wrote", out, os.path.getsize(out), "bytes;", len(convt), "videos")


if __name__ == "__main__":
    sys.exit(main())
