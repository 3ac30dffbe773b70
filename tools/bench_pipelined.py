"""Two forwards in flight on two HIP streams (two handles / workspaces, same weights and features): the HBM-bound unit
kernels of one batch run beside the L2-bound fusion convs of the other.  Serving-throughput experiment; bench.py's
`value` stays the sequential single-stream figure.   python tools/bench_pipelined.py [--batch 64] [--streams 2]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import offk_amd  # noqa: F401
from offk_amd import runtime, spec, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--length", type=int, default=7)
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--steps", type=int, default=60)
a = ap.parse_args()
B, L = a.batch, a.length
w = synth.make_weights(spec.VARIANT_RGB)
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
hs, outs, streams = [], [], []
for i in range(a.streams):
    h = runtime.OffForward(B, L, spec.VARIANT_RGB, precision="bf16x3"); h.load_state_dict(w)
    hs.append(h); outs.append([torch.empty(h.out_rows(), 101, device="cuda") for _ in range(3)]); streams.append(torch.cuda.Stream())
arr = hs[0]._feat_array(feats)


def run(n, nstreams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            hs[k].forward_into(arr, *outs[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for ns in (1, a.streams):
    run(10, ns)
    dt = run(a.steps, ns)
    print("%d stream(s): %.3f ms per forward, %.0f clips/s" % (ns, dt * 1e3, B / dt), flush=True)
ref = [o.clone() for o in outs[0]]
run(4, a.streams)
print("outputs identical across handles:", all(torch.equal(x, y) for x, y in zip(outs[0], outs[1])) and all(torch.equal(x, y) for x, y in zip(ref, outs[0])))
