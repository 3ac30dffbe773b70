"""N > 1 path on CPU: two gloo processes shard the clips, each produces its shard's scores
(the oracle stands in for the GPU forward -- it is only the checker's arithmetic here) and
offk_amd.dist assembles them.  Checks clip order, the per-shard parity definition (quirk
Q1) and the all-reduce form of late fusion."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, L, out_dir):
    sys.path.insert(0, ROOT)
    import offk_amd  # noqa: F401
    from offk_amd import dist as odist, spec, synth
    from oracle import off_oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    w = orc.to_torch_weights(synth.make_weights(spec.VARIANT_FLOW))
    first, n = odist.shard_range(B, world, rank)
    # a shard can generate exactly its slice of the global batch
    mine = [torch.from_numpy(f) for f in synth.make_features(n, L, 5, clip_offset=first)]
    res = {}
    for mode in (orc.SLICE_FLAT, orc.SLICE_PER_CLIP):
        with torch.no_grad():
            c7, c14, c28 = orc.off_forward(mine, w, n, L, spec.VARIANT_FLOW, mode, consensus=True)
        local = torch.stack((c7, c14, c28), 0)                 # [heads, b, classes]
        res[mode] = odist.gather_scores(local)
        assert res[mode].shape == (3, B, 101)
        # the all-reduce form of the same exchange (zeroed [heads, B, classes] buffer + sum): identical bits
        assert torch.equal(odist.gather_scores_allreduce(local), res[mode])
        assert torch.equal(odist.gather_scores_allreduce(local[0]), res[mode][0])
    fused = odist.fuse_scores_allreduce((1.0 + rank) * res[orc.SLICE_PER_CLIP][0])
    if rank == 0:
        np.savez(os.path.join(out_dir, "out.npz"), flat=res[0].numpy(), per_clip=res[1].numpy(), fused=fused.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    sys.path.insert(0, ROOT)
    import offk_amd  # noqa: F401
    from offk_amd import dist as odist, spec, synth
    from oracle import off_oracle as orc
    B, L, world = 4, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), B, L, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "out.npz"))
    w = orc.to_torch_weights(synth.make_weights(spec.VARIANT_FLOW))
    full = [torch.from_numpy(f) for f in synth.make_features(B, L, 5)]
    with torch.no_grad():
        # per_clip: sharded == unsharded
        ref = torch.stack(orc.off_forward(full, w, B, L, spec.VARIANT_FLOW, orc.SLICE_PER_CLIP, consensus=True), 0)
        np.testing.assert_allclose(got["per_clip"], ref.numpy(), rtol=1e-4, atol=1e-5)
        # reference_flat: sharded == reference called with batch = B/G on each shard
        parts = []
        for r in range(world):
            sh = odist.shard_features(full, B, L, world, r)
            parts.append(torch.stack(orc.off_forward(sh, w, B // world, L, spec.VARIANT_FLOW, orc.SLICE_FLAT,
                                                     consensus=True), 0))
        ref_flat = torch.cat(parts, 1)
        np.testing.assert_allclose(got["flat"], ref_flat.numpy(), rtol=1e-4, atol=1e-5)
        unsharded_flat = torch.stack(orc.off_forward(full, w, B, L, spec.VARIANT_FLOW, orc.SLICE_FLAT, consensus=True), 0)
        assert not np.allclose(got["flat"], unsharded_flat.numpy(), rtol=1e-4, atol=1e-5)   # quirk Q1 is real
    np.testing.assert_allclose(got["fused"], 3.0 * got["per_clip"][0], rtol=1e-6)


def test_shard_range_rejects_ragged():
    sys.path.insert(0, ROOT)
    import offk_amd  # noqa: F401
    from offk_amd import dist as odist
    import pytest
    assert odist.shard_range(512, 8, 3) == (192, 64)
    with pytest.raises(ValueError):
        odist.shard_range(10, 4, 0)
