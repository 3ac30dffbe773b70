"""Two-stream fused OFF forward + late score fusion (BASELINE config 5).

The RGB-OFF and Flow-OFF sub-networks of one clip batch are independent until the score level
(score_fusion.ipynb cell 8), so they run concurrently on two HIP streams of the same GPU, each
with its own liboffk handle and workspace; their per-clip consensus scores (and the backbone TSN
scores when the caller has them) are fused by K7 on the device.  Across GPUs the clips shard as in
``dist.py`` and the fused scores are gathered once.
"""
import torch

from . import dist as odist
from . import runtime, scores, spec


class TwoStreamOFF:
    def __init__(self, batch, length, precision="fp32", slice_mode=spec.SLICE_FLAT, device=None):
        """precision: "fp32" (default: the fp32 MFMA pipe) or "f32split" (fp32 operands as three bf16 planes on the bf16 pipe)."""
        self.batch, self.length = batch, length
        self.rgb = runtime.OffForward(batch, length, spec.VARIANT_RGB, slice_mode, consensus=True,
                                      device=device, precision=precision)
        self.flow = runtime.OffForward(batch, length, spec.VARIANT_FLOW, slice_mode, consensus=True,
                                       device=device, precision=precision)
        self.device = self.rgb.device
        self._streams = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))

    def load_state_dicts(self, rgb_state_dict, flow_state_dict):
        self.rgb.load_state_dict(rgb_state_dict)
        self.flow.load_state_dict(flow_state_dict)

    def forward(self, rgb_feats, flow_feats, rgb_tsn=None, flow_tsn=None, weights=scores.FUSION_BEST,
                gather=False):
        """rgb_feats / flow_feats: nine feature maps each.  rgb_tsn / flow_tsn: optional [B, 101]
        backbone consensus scores (Feature_Generation_Score after consensus).  Returns
        (fused [B, 101], pred [B]) -- or [world*B, ...] on every rank with gather=True."""
        cur = torch.cuda.current_stream(self.device)
        outs = []
        for st, h, feats in ((self._streams[0], self.rgb, rgb_feats), (self._streams[1], self.flow, flow_feats)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(h.forward(feats, want28=False))
        for st in self._streams:
            cur.wait_stream(st)
        (r7, r14, _), (f7, f14, _) = outs
        sets, ws = [], []
        for s, w in zip((r7, rgb_tsn, r14, f7, flow_tsn, f14), weights):
            if s is not None:
                sets.append(s)
                ws.append(w)
        fused, pred = runtime.score_fusion(sets, ws)
        if gather:
            fused = odist.gather_scores(fused)
            pred = fused.argmax(dim=1).to(torch.int32) if fused.shape[0] != pred.shape[0] else pred
        return fused, pred
