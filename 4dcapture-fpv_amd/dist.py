"""Frame sharding across the GPUs of one node and the per-iteration exchange step.

The reference has no distributed code; this is the MI355X design of §8e: rank r
owns a contiguous block of frames, every per-frame kernel is independent, and the only
cross-frame coupling is the 3-frame stencil of loss_smoothing (global_optimization.py:266-267)
and the 2-frame stencil of loss_world_smoothing (:304).  So per iteration each rank needs
  * its neighbours' 2 boundary rows of body_rotation_rec [78] and camera_ext [16]  (halo), and
  * the global sum of d loss / d scale (one float; scale is a shared parameter, :179).
Both travel in ONE collective per iteration: after Adam on its own rows a rank packs [first two | last two
owned rows of (x | camera_ext)] + its d loss / d scale partial into one 1.5 KB message (fdcap_opt_step_rows_and_pack),
`allgather_packed` gathers every rank's message, and fdcap_opt_unpack_and_step_scale copies the neighbours' rows into
the halo rows and sums the partials in rank order (same bits on every rank) before stepping `scale`.  The exchange is
latency-bound; backend "nccl" is RCCL over xGMI on ROCm, the CPU tests use "gloo".  No reverse exchange is needed:
every rank evaluates all loss terms that touch its own frames.  `exchange_halos` (point-to-point) fills the halo rows
once before the first iteration and after each iteration of mode 'local''s second loop; `allreduce_scalars` sums the
logged loss terms.
"""
from __future__ import annotations

HALO = 2


class FrameShard:
    """Contiguous block partition of `n_total` frames over the ranks of `group`."""

    def __init__(self, n_total: int, group=None, rank: int | None = None, world: int | None = None):
        self.n_total = int(n_total)
        self.group = group
        if rank is None or world is None:
            if group is None:
                rank, world = 0, 1
            else:
                import torch.distributed as dist
                rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world = int(rank), int(world)
        base, rem = divmod(self.n_total, self.world)
        if base < HALO and self.world > 1:
            raise ValueError(f"{n_total} frames over {world} ranks leaves fewer than {HALO} frames per rank")
        self.n_local = base + (1 if self.rank < rem else 0)
        self.frame0 = self.rank * base + min(self.rank, rem)

    def bounds(self, rank: int):
        base, rem = divmod(self.n_total, self.world)
        lo = rank * base + min(rank, rem)
        return lo, lo + base + (1 if rank < rem else 0)

    def global_rank(self, r: int) -> int:
        if self.group is None:
            return r
        import torch.distributed as dist
        return dist.get_global_rank(self.group, r)


def exchange_halos(shard: FrameShard, rows_x, rows_cam) -> None:
    """Fill the 2 halo rows each side of `rows_x` [n_local+4,78] / `rows_cam` [n_local+4,16]
    from the neighbouring ranks' boundary rows (chain topology; clip ends keep zeros).  The two
    tensors travel in one packed [2,94] message per direction: 752 B, latency-bound."""
    if shard.world == 1:
        return
    import torch
    import torch.distributed as dist
    n = shard.n_local
    ops, recv = [], []
    # gloo cannot send device tensors: stage through the host (tests run 2 ranks on one GPU this way;
    # production uses backend "nccl" = RCCL and stays on the device)
    host = rows_x.is_cuda and dist.get_backend(shard.group) == "gloo"
    stage = (lambda t: t.cpu()) if host else (lambda t: t)
    if shard.rank > 0:
        left = shard.global_rank(shard.rank - 1)
        send = stage(torch.cat([rows_x[HALO:2 * HALO], rows_cam[HALO:2 * HALO]], dim=1).contiguous())   # my first two owned rows
        buf = torch.empty_like(send)
        ops.append(dist.P2POp(dist.isend, send, left, group=shard.group))
        ops.append(dist.P2POp(dist.irecv, buf, left, group=shard.group))
        recv.append((buf, 0))
    if shard.rank < shard.world - 1:
        right = shard.global_rank(shard.rank + 1)
        send = stage(torch.cat([rows_x[n:n + HALO], rows_cam[n:n + HALO]], dim=1).contiguous())   # my last two owned rows
        buf = torch.empty_like(send)
        ops.append(dist.P2POp(dist.isend, send, right, group=shard.group))
        ops.append(dist.P2POp(dist.irecv, buf, right, group=shard.group))
        recv.append((buf, n + HALO))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    w = rows_x.shape[1]
    for buf, row in recv:
        buf = buf.to(rows_x.device)
        rows_x[row:row + HALO] = buf[:, :w]
        rows_cam[row:row + HALO] = buf[:, w:]


def allreduce_scalars(shard: FrameShard, dscale, losses=None) -> None:
    """Sum the per-rank d loss / d scale (and, when logging, the loss partial sums)."""
    if shard.world == 1:
        return
    import torch.distributed as dist
    dist.all_reduce(dscale, op=dist.ReduceOp.SUM, group=shard.group)      # gloo reduces device tensors through the host itself
    if losses is not None:
        dist.all_reduce(losses, op=dist.ReduceOp.SUM, group=shard.group)


def allgather_packed(shard: FrameShard, send, gathered, overlap=None) -> None:
    """gathered[r] = rank r's `send` (one small fixed-size message per rank and iteration: the iteration's only
    collective).  `gathered` is [world, len(send)] contiguous; it is handed to the backend as the flat concatenation
    (RCCL and gloo both take that form; gloo refuses the 2-D view).

    overlap: a callable that enqueues work which does not need the gathered data (fdcap_opt_forward_ahead).  Over RCCL the
    collective is issued asynchronously -- it runs on the backend's own stream once `send` is complete --, `overlap()` puts
    its launches on the current stream behind the packing launch, and only then is the current stream made to wait for the
    collective: the launches run while the messages travel.  (gloo, tests: the same order of calls, no concurrency.)"""
    import torch.distributed as dist
    if send.is_cuda and dist.get_backend(shard.group) == "gloo":      # tests: gloo cannot gather device tensors
        flat = gathered.new_empty(gathered.numel(), device="cpu")
        host = send.cpu()
        if overlap is not None:
            overlap()
        dist.all_gather_into_tensor(flat, host, group=shard.group)
        gathered.copy_(flat.view_as(gathered))
    elif overlap is not None and send.is_cuda:
        work = dist.all_gather_into_tensor(gathered.view(-1), send, group=shard.group, async_op=True)
        overlap()
        work.wait()                                                   # (stream-level: the host does not block)
    else:
        dist.all_gather_into_tensor(gathered.view(-1), send, group=shard.group)
        if overlap is not None:
            overlap()


def agree_on(shard: FrameShard, value: int) -> int:
    """Rank 0's `value` on every rank (one tiny broadcast).  For per-rank configuration that decides WHETHER a collective is
    issued -- e.g. `verbose`, which makes a fit read its loss history back (an all-reduce) every few logged iterations:
    a rank that disagreed would issue another sequence of collectives than its peers and hang the group."""
    if shard.world == 1:
        return int(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int64)
    if dist.get_backend(shard.group) != "gloo":
        t = t.cuda()
    dist.broadcast(t, src=shard.global_rank(0), group=shard.group)
    return int(t.item())


def same_on_all_ranks(shard: FrameShard, value: int):
    """(min, max) of `value` over the ranks -- equal when every rank holds the same number."""
    if shard.world == 1:
        return int(value), int(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value), -int(value)], dtype=torch.int64)
    if dist.get_backend(shard.group) != "gloo":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=shard.group)
    return int(t[0].item()), -int(t[1].item())


def barrier(shard: FrameShard) -> None:
    if shard.world > 1:
        import torch.distributed as dist
        dist.barrier(group=shard.group)
