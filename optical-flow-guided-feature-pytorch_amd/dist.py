"""Clip sharding over the GPUs of one node and the one collective on the path.

The reference only knows nn.DataParallel (Flow_OFF.py:1415, model_utils.py:350) and has
no explicit collective.  The OFF forward shards over clips (temporal coupling never
crosses a clip, RGB_OFF.py:600-603): rank r owns a contiguous block of clips, runs the
whole forward locally with replicated weights, and the only exchange is the per-clip
score tensors at the end -- one RCCL all-gather over xGMI ([B/G, 101] x heads, <= 620 KB:
latency-bound, so a single fused collective rather than one per head).

Parity note (quirk Q1, RGB_OFF.py:609): in reference_flat slice mode the result of a
forward depends on how clips are grouped into a call, so the sharded result equals the
reference called with batch = B/G on each shard (what DataParallel would have had to do),
not the reference called once with batch = B.  In per_clip mode both coincide.
"""
import torch
import torch.distributed as dist


def shard_range(total_clips, world, rank):
    """Contiguous block of clips owned by ``rank``: (first_clip, n_clips).  Requires an even split,
    as every rank's handle is built for the same (batch, length)."""
    if total_clips % world:
        raise ValueError("clips (%d) must divide evenly over %d ranks" % (total_clips, world))
    n = total_clips // world
    return rank * n, n


def shard_features(feats, total_clips, length, world, rank):
    """Slice each [B*L, ...] feature map down to this rank's frames (views, no copy)."""
    first, n = shard_range(total_clips, world, rank)
    return [f[first * length:(first + n) * length] for f in feats]


def _active(group, always):
    """A collective is issued when a process group exists and has more than one rank -- or, ``always``, also on a single-rank group
    (bench.py's single-rank RCCL smoke: the same librccl calls, init included, on the one GPU a dev box has)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return always or dist.get_world_size(group) > 1


def all_gather_scores_into(gathered, local, group=None, async_op=False):
    """One fused all-gather of this rank's [heads, b, classes] scores into ``gathered`` [world, heads, b, classes] (both contiguous,
    caller-owned: the benchmark alternates two buffer sets).  CUDA/HIP tensors: RCCL ``all_gather_into_tensor``.

    ``async_op=True`` returns the collective's work handle instead of waiting on it: RCCL runs the exchange on its own stream
    behind the forward that produced ``local``, and the CALLER's stream is not ordered behind it -- the forward of the next step
    starts while the scores of this one travel.  ``wait()`` the handle in front of whatever reads ``gathered`` or re-uses either
    buffer (``ScoreExchange`` below does that for a loop of steps).  With ``async_op=False`` the current stream waits for the
    exchange at once: one step's wire time in front of every next forward."""
    world, heads, b, c = gathered.shape
    work = dist.all_gather_into_tensor(gathered.view(world * heads * b, c), local.view(heads * b, c), group=group, async_op=async_op)
    return work if async_op else gathered


def all_reduce_scores_inplace(buf, group=None, async_op=False):
    """The literal north-star form: ``buf`` [heads, world * b, classes] is zero except for this rank's rows; ONE sum all-reduce.
    ``async_op`` as in all_gather_scores_into."""
    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else buf


class ScoreExchange:
    """The per-step exchange of a loop of forwards with the collective OFF the compute stream's critical path: ``nbuf`` alternating
    buffer sets; step i launches its collective asynchronously and only waits for the collective that used the same buffer set
    ``nbuf`` steps earlier (long finished).  ``finish()`` waits for everything outstanding -- call it in front of the consumer of
    the gathered scores.  form: "allgather" (RCCL all_gather_into_tensor into [world, heads, b, classes]) or "allreduce" (the
    zero-buffer sum all-reduce of [heads, world * b, classes]; the caller fills its rows, e.g. as the forward's outputs)."""

    def __init__(self, nbuf=2, group=None):
        self.group, self.pending = group, [None] * nbuf

    def wait_slot(self, i):
        """Before buffer set ``i`` is written again (by the next forward into it, or by zeroing it)."""
        w = self.pending[i]
        if w is not None:
            w.wait()
            self.pending[i] = None

    def all_gather(self, i, gathered, local):
        self.pending[i] = all_gather_scores_into(gathered, local, group=self.group, async_op=True)

    def all_reduce(self, i, buf):
        self.pending[i] = all_reduce_scores_inplace(buf, group=self.group, async_op=True)

    def finish(self):
        for i in range(len(self.pending)):
            self.wait_slot(i)


def gather_scores(local, group=None, always=False):
    """local: [heads, b, classes] (or [b, classes]) per-clip consensus scores of this rank.
    Returns the same with b -> B = world*b, clips in global order, on every rank."""
    if not _active(group, always):
        return local
    world = dist.get_world_size(group)
    squeeze = local.dim() == 2
    x = local.unsqueeze(0) if squeeze else local
    x = x.contiguous()
    heads, b, c = x.shape
    buf = torch.empty(world, heads, b, c, dtype=x.dtype, device=x.device)
    if x.is_cuda:
        all_gather_scores_into(buf, x, group=group)
    else:  # gloo (CPU tests): list form
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x, group=group)
        buf = torch.stack(parts, 0)
    out = buf.permute(1, 0, 2, 3).reshape(heads, world * b, c)
    return out[0] if squeeze else out


def gather_scores_allreduce(local, group=None, rank=None, always=False):
    """The literal north-star form of the same exchange ("RCCL ... only for the final consensus all-reduce"; SURVEY.md
    section 8(e)): every rank writes its shard into a zeroed [heads, B, classes] buffer at its clips' rows and ONE sum
    all-reduce assembles the whole batch on every rank.  x + 0 is exact, so the result equals gather_scores bit for bit;
    it moves world x the bytes (still <= 620 KB at 8 GPUs: latency-bound either way)."""
    if not _active(group, always):
        return local
    world = dist.get_world_size(group)
    r = dist.get_rank(group) if rank is None else rank
    squeeze = local.dim() == 2
    x = local.unsqueeze(0) if squeeze else local
    heads, b, c = x.shape
    buf = torch.zeros(heads, world * b, c, dtype=x.dtype, device=x.device)
    buf[:, r * b:(r + 1) * b] = x
    all_reduce_scores_inplace(buf, group=group)
    return buf[0] if squeeze else buf


def fuse_scores_allreduce(weighted_local, group=None, always=False):
    """Late two-stream fusion when the RGB and Flow streams live on different ranks
    (score_fusion.ipynb lines 300-301: sum_i w_i * score_i): every rank passes its own
    pre-weighted [B, classes] contribution; a sum all-reduce yields the fused scores."""
    if not _active(group, always):
        return weighted_local
    out = weighted_local.contiguous().clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out
