"""K2 HBM traffic per launch from rocprofv3 PMC passes -> k2_traffic.json (what bench.py reports as roofline.traffic).

    profiles/collect_pmc.sh pmc_rNN            # on the GPU box: separate --pmc passes of a short bench run
    python tools/measure_k2_traffic.py gpurun_out/pmc_rNN gpurun_out/k2_traffic.json
    cp gpurun_out/k2_traffic.json profiles/k2_traffic.json        # back in the dev container

FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 tallies the 128-B requests of a wide coalesced stream at
64 B), WRITE_SIZE is taken as it is; both counters are in KB.  Only dispatches of the full K2 launch are used (the
S-blocks-only launch of the fused-units path has a different grid size).  The record carries the sha256 of
sobel_tdiff.hip: bench.py drops the figure to null when the kernel text is no longer the one measured."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_mean(root, counter, kernel="sobel_tdiff_kernel"):
    by_grid = {}
    for path in glob.glob(os.path.join(root, counter, "*", "*_counter_collection.csv")):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    by_grid.setdefault(int(row["Grid_Size"]), []).append(float(row["Counter_Value"]))
    if not by_grid:
        raise SystemExit("no %s rows for %s under %s" % (counter, kernel, root))
    grid = max(by_grid)                      # the full launch (T-blocks + S-blocks) is the larger grid
    vals = by_grid[grid]
    return sum(vals) / len(vals), len(vals), grid


def main(root, out_path, batch=64, length=7, variant="rgb"):
    fetch_kb, nf, grid = counter_mean(root, "FETCH_SIZE")
    write_kb, nw, _ = counter_mean(root, "WRITE_SIZE")
    src = open(os.path.join(ROOT, "optical-flow-guided-feature-pytorch_amd", "csrc", "sobel_tdiff.hip"), "rb").read()
    sys.path.insert(0, ROOT)
    import offk_amd  # noqa: F401
    from offk_amd import spec
    rec = {
        "kernel": "offk::sobel_tdiff_kernel (K2, grouped launch over all nine sites, grid %d work-items)" % grid,
        "kernel_source_sha256": hashlib.sha256(src).hexdigest(),
        "batch": batch, "length": length, "variant": variant,
        "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "dispatches": [nf, nw],
        "fetch_correction": "x2 (gfx950 counts 128-B requests as 64 B for wide coalesced streams, MI355X_MICROARCH.md HBM section)",
        "hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)),
        "algorithmic_bytes_per_launch": spec.algorithmic_bytes_sobel_tdiff(batch, length),
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (profiles/collect_pmc.sh), dir %s" % os.path.basename(root.rstrip("/")),
    }
    rec["ratio_to_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    json.dump(rec, open(out_path, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
