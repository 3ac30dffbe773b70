import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import offk_amd
from offk_amd import runtime, spec, synth
B, L = 64, 7
N, P = B*L, B*(L-1)
h = runtime.OffForward(B, L, spec.VARIANT_RGB)
h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
for si, H in ((0, 28), (2, 14)):
    G = torch.relu(torch.randn(N*H*H, 128, device="cuda")); D = torch.randn(P*H*H, 32, device="cuda")
    for cs in (160, 320, 1056):
        M = torch.empty(P*H*H, cs, device="cuda")
        for algo in (2, 5, 3, 0, 4):
            for _ in range(3): h.sobel_tdiff(si, G, D, M, 0, algo)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): h.sobel_tdiff(si, G, D, M, 0, algo)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1)/20*1e3
            print("site H=%d m_cs=%4d algo=%d  %7.1f us" % (H, cs, algo, us), flush=True)
        del M
