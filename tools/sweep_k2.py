"""K2 knob sweep in ONE process (tuning build only: tools/_bin/liboffk_tune.so is liboffk compiled with
-DOFFK_TUNING_KNOBS, which re-reads the OFFK_K2_* environment on every launch).
    OFFK_LIB=tools/_bin/liboffk_tune.so python tools/sweep_k2.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

B, L = 64, 7
h = runtime.OffForward(B, L, spec.VARIANT_RGB)
h.load_state_dict(synth.make_weights(spec.VARIANT_RGB))
feats = [torch.from_numpy(f).cuda() for f in synth.make_features(B, L, 2)]
h.off_units(feats)
torch.cuda.synchronize()
full = spec.algorithmic_bytes_sobel_tdiff(B, L)


def timed(algo, iters=20):
    for _ in range(3):
        h.sobel_tdiff_all(algo)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        h.sobel_tdiff_all(algo)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


configs = [{}]
for nt in (0, 1, 2):
    configs.append({"OFFK_K2_NT": nt})
for tp in (4, 16, 32):
    configs.append({"OFFK_K2_TPIX": tp})
configs += [{"OFFK_K2_FLATROWS": 16}, {"OFFK_K2_FLATROWS": 32}, {"OFFK_K2_FLATROWS": 32, "OFFK_K2_NT": 0}, {"OFFK_K2_ROWS28": 4}]
print("config                                   rot+S   flat+S   T-rot  T-flat  S-only   (us; %d algorithmic bytes)" % full)
for rnd in range(2):
    for cfg in configs:
        for k in ("OFFK_K2_TPIX", "OFFK_K2_FLATROWS", "OFFK_K2_ROWS28", "OFFK_K2_ROWS14", "OFFK_K2_NT"):
            os.environ.pop(k, None)
        for k, v in cfg.items():
            os.environ[k] = str(v)
        t = [timed(a) for a in (0, 4, 2, 5, 3)]
        print("%-40s %6.1f  %6.1f  %6.1f  %6.1f  %6.1f   best %.0f GB/s" % (str(cfg), t[0], t[1], t[2], t[3], t[4], full / min(t[0], t[1]) / 1e3), flush=True)
