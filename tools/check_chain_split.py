"""tools: quick check of the split chains: B = 12 (P = 72: chains by default), stage tensor sum_28c (fusion_14 channels 800..) vs oracle, both modes; timing at B = 64"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import offk_amd
from offk_amd import runtime, spec, synth
from oracle import off_oracle as orc
B, L = 12, 7
feats_np = synth.make_features(B, L, 2)
w = synth.make_weights(0)
with torch.no_grad():
    want, st = orc.off_forward([torch.from_numpy(f) for f in feats_np], orc.to_torch_weights(w), B, L, 0, orc.SLICE_FLAT, return_stages=True)
P = B * (L - 1)
for prec in ("fp32", "f32split"):
    h = runtime.OffForward(B, L, 0, precision=prec); h.load_state_dict(w)
    out = h.forward([torch.from_numpy(f).cuda() for f in feats_np]); torch.cuda.synchronize()
    f14 = h.region("fusion_14", 1056).view(P, 14, 14, 1056)[..., 800:].permute(0, 3, 1, 2).double().cpu()
    ref = st["sum_28c"].double()
    e = (f14 - ref).abs()
    print(prec, "sum_28c max err / max", (e.max() / ref.abs().max()).item(), "rms", ((e**2).mean().sqrt() / ref.abs().max()).item(),
          "logits", max(((o.cpu().double() - r.double()).abs().max() / r.double().abs().max()).item() for o, r in zip(out, want)))
# timing at B = 64: the chain launches in both modes
for prec in ("fp32", "f32split"):
    B = 64
    h = runtime.OffForward(B, L, 0, precision=prec); h.load_state_dict(w)
    feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
    arr = h._feat_array(feats)
    out = [torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]
    for _ in range(5):
        h.forward_into(arr, *out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        h.forward_into(arr, *out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    h.set_profiling(2); h.launch_times(reset=True)
    for _ in range(20):
        h.forward_into(arr, *out)
    torch.cuda.synchronize()
    print(prec, "%.4f ms / forward" % ms, " ".join("%s %.1f" % (n[:9], t / c * 1e3) for n, (t, c) in h.launch_times().items() if n.startswith("chain")))
