"""Per-item error map of the split-fp32 batched GEMM: python tools/debug_gemm_split.py batch M K Co"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
PREC = os.environ.get("DEBUG_PREC", "f32split")

import offk_amd  # noqa: F401
from offk_amd import runtime

batch, M, K, Co = (int(a) for a in sys.argv[1:5])
g = torch.Generator().manual_seed(3)
x = torch.randn(batch, M, K, generator=g)
if len(sys.argv) > 5 and sys.argv[5] == "relu":
    x = torch.relu(x)
w = torch.randn(batch, Co, K, generator=g) / K ** 0.5
ref = torch.einsum("bmk,bnk->bmn", x.double(), w.double())
y = runtime.batched_gemm_nt(x.cuda(), w.cuda(), PREC).double().cpu()
y2 = runtime.batched_gemm_nt(x.cuda(), w.cuda(), PREC).double().cpu()
print("run-to-run identical:", bool(torch.equal(y, y2)))
err = (y - ref).abs()
print("max err", err.max().item(), "of", ref.abs().max().item())
gm, gn = (M + 63) // 64, Co // 128
bad = []
for b in range(batch):
    for mt in range(gm):
        for nt in range(gn):
            e = err[b, mt * 64:(mt + 1) * 64, nt * 128:(nt + 1) * 128].max().item()
            if e > 1e-4:
                bad.append((b, mt, nt, e))
print(len(bad), "bad items of", batch * gm * gn)
for t in bad[:40]:
    b, mt, nt, e = t
    blk = err[b, mt * 64:(mt + 1) * 64, nt * 128:(nt + 1) * 128]
    rows = (blk.max(dim=1).values > 1e-4).nonzero().flatten().tolist()
    cols = (blk.max(dim=0).values > 1e-4).nonzero().flatten().tolist()
    print("item l=%d" % ((b * gm + mt) * gn + nt), t, "rows", rows[:6], "..", len(rows), "cols", cols[:6], "..", len(cols))
