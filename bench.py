"""OFF-forward benchmark (BASELINE.json metric: OFF-forward clips/sec, 7-seg 224x224).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the OFF sub-network forward (liboffk: nine OFF units, fusion
@28/@14/@7, three heads) over one batch of synthetic BN-Inception feature maps already
resident in HBM.  N = 1 is BASELINE config 2 (RGB_OFF, B = 64 clips x 7 segments).  N > 1
is config 4: every rank owns 64 clips (weak scaling, no data-path collective), takes the
SegmentConsensus average per clip and the per-clip scores are exchanged once per step with
one RCCL all-gather over xGMI.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import offk_amd  # noqa: E402,F401
from offk_amd import runtime, spec, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA peak


def cpu_baseline(feats_np, weights, length, variant, clips):
    """The oracle (a port of the reference's op sequence, bit-exact against it in the dev
    container) timed on this box's host cores -- reported beside the GPU number only.
    torch's intra-op threading does not scale to every logical CPU on these small convs, so a
    few thread counts are tried (one warm-up + one run each) and the fastest is measured
    properly: the baseline is the best the host does, not a strawman."""
    from oracle import off_oracle as orc
    w = orc.to_torch_weights(weights)
    x = [torch.from_numpy(f[:clips * length]) for f in feats_np]

    def run():
        t0 = time.perf_counter()
        orc.off_forward(x, w, clips, length, variant, orc.SLICE_FLAT)
        return time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    best_t, best_n, sweep = None, default_threads, {}
    with torch.no_grad():
        for n in sorted({default_threads, max(1, ncpu // 2), 64, 32, 16}):
            if n > ncpu:
                continue
            torch.set_num_threads(n)
            run()
            t = run()
            sweep[n] = clips / t
            if best_t is None or t < best_t:
                best_t, best_n = t, n
        torch.set_num_threads(best_n)
        times = [run() for _ in range(2 + 5)][2:]
    med = statistics.median(times)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": clips / med, "unit": "clips/s", "cores": best_n, "kind": "port",
            "sample": "oracle/off_oracle.py (torch CPU ops) on the first %d clips x %d segments of the same "
                      "synthetic maps, 2 warm-ups, median of 5" % (clips, length),
            "cpu_model": model, "host_logical_cpus": os.cpu_count(), "sec_per_forward": med,
            "clips_per_s_by_threads": sweep}


def measured_traffic(batch, length, variant):
    """HBM bytes per K2 launch from the rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950
    correction in MI355X_MICROARCH.md + WRITE_SIZE), recorded in profiles/k2_traffic.json for the
    configuration it was collected on; None for any other configuration."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "k2_traffic.json")))
    except (OSError, ValueError):
        return None
    if rec.get("batch") == batch and rec.get("length") == length and rec.get("variant") == variant:
        return rec.get("hbm_bytes_per_launch")
    return None


def units_training(_unused, B, L, variant, weights, feats, dev, precision, iters=10):
    """Secondary figure: train-mode forward of the nine units (dropout 0.8) and their
    backward (dM -> parameter gradients) at the bench size; random dM, same synthetic maps."""
    P = B * (L - 1)
    ht = runtime.OffForward(B, L, variant, spec.SLICE_FLAT, False, device=dev, precision=precision, training=True)
    ht.load_state_dict(weights)
    gen = torch.Generator(device=dev).manual_seed(5)
    bufs = [torch.randn(P, H, H, C, device=dev, generator=gen) for H, C in ((28, 320), (14, 1056), (7, 832))]
    views = [(bufs[0], 0), (bufs[0], 160)] + [(bufs[1], 160 * k) for k in range(5)] + [(bufs[2], 0), (bufs[2], 160)]
    grads = ht.new_unit_grads()

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    fwd = timed(lambda: ht.off_units_train(feats, 21, 0.8))
    bwd = timed(lambda: ht.off_units_backward(feats, views, 21, 0.8, grads=grads))
    hw = sum(H * H for _n, _c, H in spec.SITES)
    k2b = B * hw * 4 * ((160 + 32 + 32) * (L - 1) + 256 * L)
    k1b = sum(B * L * C * H * H * 4 for _n, C, H in spec.SITES) + B * hw * 4 * (128 * L + 32 * (L - 1))
    return {"train_forward_ms": fwd, "backward_ms": bwd, "clips_per_s_forward_plus_backward": B / (fwd + bwd) * 1e3,
            "backward_algorithmic_bytes": k2b + k1b, "backward_hbm_floor_ms_at_8TBs": (k2b + k1b) / 8e12 * 1e3,
            "note": "units only (K1+K2 train mode; K2b + weight-gradient GEMM + reductions); fusion stages / heads train on the caller's autograd"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--length", type=int, default=7)
    ap.add_argument("--variant", choices=("rgb", "flow"), default="rgb")
    ap.add_argument("--precision", choices=("fp32", "bf16x3"), default="bf16x3",
                    help="arithmetic of the contractions; both modes pass the parity tests (fp32: ~4e-7, bf16x3: ~1e-5 rel)")
    ap.add_argument("--cpu-clips", type=int, default=8, help="clips in the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    variant = spec.VARIANT_RGB if args.variant == "rgb" else spec.VARIANT_FLOW
    B, L = args.batch, args.length
    consensus = (variant == spec.VARIANT_FLOW) or world > 1
    weights = synth.make_weights(variant)
    feats_np = synth.make_features(B, L, config_id=2, clip_offset=rank * B)
    feats = [torch.from_numpy(f).to(dev) for f in feats_np]
    h = runtime.OffForward(B, L, variant, spec.SLICE_FLAT, consensus, device=dev, precision=args.precision)
    h.load_state_dict(weights)
    arr = h._feat_array(feats)
    rows = h.out_rows()
    out = [torch.empty(rows, spec.NUM_CLASSES, device=dev) for _ in range(3)]
    # two buffer sets, alternated per step: the collective of step i (RCCL stream) may still be reading
    # its input while the forward of step i+1 is enqueued on the compute stream
    gathered = [torch.empty(world, 3, rows, spec.NUM_CLASSES, device=dev) for _ in range(2)] if world > 1 else None
    local = [torch.empty(3, rows, spec.NUM_CLASSES, device=dev) for _ in range(2)] if world > 1 else None
    counter = [0]

    def step():
        if world > 1:
            i = counter[0] & 1
            counter[0] += 1
            h.forward_into(arr, local[i][0], local[i][1], local[i][2])
            # config 4: the only exchange on the path -- per-clip consensus scores [rank][head][clip][class],
            # one collective (same call offk_amd.dist.gather_scores makes; tests/test_dist_gloo.py)
            dist.all_gather_into_tensor(gathered[i].view(world * 3 * rows, -1), local[i].view(3 * rows, -1))
        else:
            h.forward_into(arr, out[0], out[1], out[2])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    h.set_profiling(True)
    h.stage_times(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stages = h.stage_times(reset=True)
    h.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        clips_s = world * B * args.steps / dt
        k2_ms, k2_calls = stages["sobel_tdiff"]
        k2_avg_s = k2_ms / max(k2_calls, 1) * 1e-3
        algo_bytes = spec.algorithmic_bytes_sobel_tdiff(B, L)
        achieved = algo_bytes / k2_avg_s / 1e9 if k2_avg_s > 0 else 0.0
        unit_f, fus_f = spec.flops_per_clip(L)
        stage_ms = dict((k, v[0] / max(v[1], 1)) for k, v in stages.items())
        gpu_ms = sum(stage_ms.values())
        res = {
            "metric": "OFF-forward clips/sec (7-seg 224x224)", "value": clips_s, "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16x3 (fp32 split into bf16 hi+lo, 3 MFMA products, f32 accumulate)",
            "data": "synthetic (portable counter-based generator: ReLU-like non-negative BN-Inception "
                    "feature maps, fan-in-scaled uniform weights; features resident in HBM)",
            "config": {"workload": "%s_OFF forward, batch=%d clips/GPU x %d segments, nine 224x224-geometry "
                                   "feature maps%s" % (args.variant.upper(), B, L,
                                                        " + SegmentConsensus avg + RCCL all-gather of per-clip scores"
                                                        if world > 1 else ""),
                       "global_batch": world * B, "segments": L, "parallelism": "clip-shard x%d" % world,
                       "slice_mode": "reference_flat"},
            "roofline": {"bound": "hbm", "kernel": "sobel_tdiff_kernel (K2, all nine sites, one launch)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(B, L, args.variant),
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": k2_avg_s * 1e6},
            "stage_ms": stage_ms,
            "mfma": {"flops_per_step": (unit_f + fus_f) * B, "achieved_tflops": (unit_f + fus_f) * B / (gpu_ms * 1e-3) / 1e12
                     if gpu_ms > 0 else 0.0,
                     "peak_tflops": MFMA_F32_PEAK_TFLOPS if args.precision == "fp32" else 2500.0 / 3.0,
                     "note": "algorithmic fp32 FLOPs / summed stage time; bf16x3 peak = dense bf16 MFMA peak / 3 products"},
        }
        if world == 1 and args.precision != "fp32":
            # the exact-fp32 arithmetic mode of the same library, for reference (fewer steps)
            h32 = runtime.OffForward(B, L, variant, spec.SLICE_FLAT, consensus, device=dev, precision="fp32")
            h32.load_state_dict(weights)
            for _ in range(3):
                h32.forward_into(arr, out[0], out[1], out[2])
            torch.cuda.synchronize()
            n32 = max(5, args.steps // 5)
            t1 = time.perf_counter()
            for _ in range(n32):
                h32.forward_into(arr, out[0], out[1], out[2])
            torch.cuda.synchronize()
            d32 = time.perf_counter() - t1
            res["fp32_mode"] = {"value": B * n32 / d32, "unit": "clips/s", "ms_per_step": d32 / n32 * 1e3, "steps": n32,
                                "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}
        if world == 1:
            res["units_training"] = units_training(None, B, L, variant, weights, feats, dev,
                                                   args.precision)
        if world == 1 and args.cpu_clips > 0:
            res["cpu_baseline"] = cpu_baseline(feats_np, weights, L, variant, min(args.cpu_clips, B))
            res["gpu_over_cpu"] = clips_s / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
