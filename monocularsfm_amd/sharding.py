"""Multi-GPU sharding of the all-pairs job: one process per GPU, torch.distributed (RCCL) gather.

The reference is single-process; image pairs are independent
(/root/reference/src/Feature/FeatureMatching.cpp:14), so the pair list is partitioned and the
only exchange step is an all-gather of the per-pair match lists at the end:

  1. every rank holds the full descriptor store (<= 34 GB even for the largest config, vs 288 GB HBM);
  2. the pair list is cut into world_size contiguous ranges of equal total cost sum n_i * n_j;
  3. each rank runs its pairs through the C ABI;
  4. counts: one all_reduce(sum) of a per-pair count vector; payload: rank-padded (queryIdx, trainIdx
     [, distance-bits]) int32 rows, either gathered to the writer rank (gather_to_writer: the CLI flow, only
     the owner of the SQLite handle needs them) or all-gathered (gather_matches).  The payload is MBs, i.e.
     latency-bound on xGMI -- no bucketing needed.

`torch` is used for the process group and the device tensors of the collectives only.
"""
import numpy as np

def partition_pairs(pairs, n_rows, world_size):
    """-> list (per rank) of int64 index arrays into `pairs`: CONTIGUOUS, cost-balanced ranges.

    pairs: P x 2 image ids; n_rows: array image id -> descriptor count.  Rank r owns the pairs
    [cut[r], cut[r+1]) where the cuts split the prefix sum of the per-pair cost n_i * n_j evenly
    (imbalance <= one pair).  Contiguous ranges keep every rank's results in global pair order, so the
    gathered payload needs no reordering pass -- at 8 GPUs the per-rank compute of the South-Building
    job is ~25 ms and a scatter of 2.5 M matches in NumPy would cost as much again."""
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    P = pairs.shape[0]
    if world_size <= 1 or P == 0:
        return [np.arange(P, dtype=np.int64)] + [np.zeros(0, np.int64) for _ in range(max(world_size - 1, 0))]
    n_rows = np.asarray(n_rows, dtype=np.int64)
    cost = (n_rows[pairs[:, 0]] * n_rows[pairs[:, 1]]).astype(np.float64) + 1.0  # +1: empty images still cost a slot
    csum = np.cumsum(cost)
    targets = csum[-1] * np.arange(1, world_size) / world_size
    cuts = np.concatenate([[0], np.searchsorted(csum, targets, side="left") + 1, [P]])
    cuts = np.minimum.accumulate(np.minimum(cuts, P)[::-1])[::-1]  # monotone, clipped
    cuts = np.maximum.accumulate(cuts)
    return [np.arange(cuts[r], cuts[r + 1], dtype=np.int64) for r in range(world_size)]


def gather_matches(local_idx, local_offs, local_qt, local_dist, n_pairs, group=None, device=None):
    """All-gather the per-pair match lists.  Every rank returns the full CSR
    (offsets int64[P+1], qt int32[M,2], dist float32[M]) in global pair order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    local_idx = np.asarray(local_idx, np.int64)
    local_counts = np.diff(np.asarray(local_offs, np.int64))
    if world == 1:
        offs = np.zeros(n_pairs + 1, np.int64)
        cnt = np.zeros(n_pairs, np.int64)
        cnt[local_idx] = local_counts
        np.cumsum(cnt, out=offs[1:])
        # local results are in local_idx order; local_idx ascending => already global order
        return offs, np.asarray(local_qt, np.int32).reshape(-1, 2), np.asarray(local_dist, np.float32)
    dev = device if device is not None else torch.device("cpu")
    rank = dist.get_rank(group)

    # (1) counts + owner of every pair
    counts = torch.zeros(n_pairs, dtype=torch.int32)
    owner = torch.zeros(n_pairs, dtype=torch.int32)
    counts[torch.from_numpy(local_idx)] = torch.from_numpy(local_counts.astype(np.int32))
    owner[torch.from_numpy(local_idx)] = rank
    co = torch.stack([counts, owner]).to(dev)
    dist.all_reduce(co, op=dist.ReduceOp.SUM, group=group)
    co = co.cpu().numpy()
    counts_all, owner_all = co[0].astype(np.int64), co[1].astype(np.int64)

    # (2) payload, padded to the largest rank
    per_rank_total = np.bincount(owner_all, weights=counts_all, minlength=world).astype(np.int64)
    max_m = int(per_rank_total.max()) if world > 0 else 0
    send = torch.zeros((max(max_m, 1), 3), dtype=torch.int32)
    m_local = int(local_counts.sum())
    if m_local:
        send[:m_local, 0:2] = torch.from_numpy(np.ascontiguousarray(local_qt, dtype=np.int32).reshape(-1, 2))
        send[:m_local, 2] = torch.from_numpy(np.ascontiguousarray(local_dist, dtype=np.float32).view(np.int32))
    send = send.to(dev)
    recv = torch.empty((world,) + tuple(send.shape), dtype=torch.int32, device=dev)
    # list form: supported by both RCCL ("nccl") and gloo (the CPU tests)
    dist.all_gather([recv[r] for r in range(world)], send, group=group)
    recv = recv.cpu().numpy()

    # (3) reassemble in global pair order.  With contiguous per-rank ranges (partition_pairs) this is a
    # plain concatenation; an arbitrary ownership pattern falls back to a scatter.
    offs = np.zeros(n_pairs + 1, np.int64)
    np.cumsum(counts_all, out=offs[1:])
    M = int(offs[-1])
    contiguous = bool(np.all(np.diff(owner_all) >= 0))
    if contiguous:
        parts = [recv[r, :int(per_rank_total[r])] for r in range(world)]
        allm = np.concatenate(parts) if parts else np.zeros((0, 3), np.int32)
        return offs, np.ascontiguousarray(allm[:, 0:2]), np.ascontiguousarray(allm[:, 2]).view(np.float32)
    qt = np.empty((M, 2), np.int32)
    dd = np.empty(M, np.int32)
    for r in range(world):
        idx = np.nonzero(owner_all == r)[0]  # ascending == that rank's local order
        if len(idx) == 0:
            continue
        c = counts_all[idx]
        src_off = np.concatenate([[0], np.cumsum(c)])
        dst = np.repeat(offs[idx] - src_off[:-1], c) + np.arange(int(src_off[-1]))
        qt[dst] = recv[r, :int(src_off[-1]), 0:2]
        dd[dst] = recv[r, :int(src_off[-1]), 2]
    return offs, qt, dd.view(np.float32)


_staging = {}


def _host_buffer(rows, cols, pinned, slot="recv"):
    """Host staging buffers of the exchange step; pinned (page-locked) when they talk to a GPU.
    Pinning 100+ MB costs tens of milliseconds, so the buffers are kept and only grown."""
    import torch
    key = (slot, cols, bool(pinned))
    buf = _staging.get(key)
    if buf is None or buf.shape[0] < rows:
        buf = torch.empty((max(rows, 1) * 5 // 4 + 1024, cols), dtype=torch.int32, pin_memory=bool(pinned))
        _staging[key] = buf
    return buf[:rows]


def gather_to_writer(local_idx, local_offs, local_qt, local_dist, n_pairs, dst=0, group=None, device=None,
                     with_dist=True, force_collectives=False):
    """The exchange step of the CLI flow: only the rank that owns the SQLite handle needs the match lists.

    Every rank returns the global CSR offsets (one all_reduce of the per-pair counts); rank `dst` also returns
    qt int32[M,2] (and dist float32[M] if with_dist) in global pair order, the other ranks return None for
    them.  With with_dist=False the returned qt is a view of a reused staging buffer: valid until the next call.  The `matches` table stores index pairs only (Database.cpp:631-654), so with_dist=False is what the
    writer needs; the payload then is 8 bytes per match.  Per-rank pair ranges must be contiguous and in rank
    order (partition_pairs), so the payload is reassembled by concatenation."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    local_idx = np.asarray(local_idx, np.int64)
    local_counts = np.diff(np.asarray(local_offs, np.int64))
    qt_local = np.ascontiguousarray(local_qt, dtype=np.int32).reshape(-1, 2)
    if world == 1 and not (force_collectives and dist.is_initialized()):  # force: single-rank check of the RCCL path
        offs = np.zeros(n_pairs + 1, np.int64)
        cnt = np.zeros(n_pairs, np.int64)
        cnt[local_idx] = local_counts
        np.cumsum(cnt, out=offs[1:])
        return offs, qt_local, (np.asarray(local_dist, np.float32) if with_dist else None)
    dev = device if device is not None else torch.device("cpu")
    rank = dist.get_rank(group)
    if len(local_idx) > 1 and not (np.diff(local_idx) == 1).all():
        raise ValueError("gather_to_writer needs contiguous per-rank pair ranges (partition_pairs)")

    counts = torch.zeros(n_pairs, dtype=torch.int32)
    counts[torch.from_numpy(local_idx)] = torch.from_numpy(local_counts.astype(np.int32))
    starts = torch.full((world,), n_pairs, dtype=torch.int32)
    starts[rank] = int(local_idx[0]) if len(local_idx) else n_pairs
    cs = torch.cat([counts, starts]).to(dev)
    # counts: disjoint supports -> SUM is a concatenation; starts: every rank contributes its own slot on top
    # of the n_pairs fill of the others -> subtract (world - 1) * n_pairs afterwards
    dist.all_reduce(cs, op=dist.ReduceOp.SUM, group=group)
    cs = cs.cpu().numpy().astype(np.int64)
    counts_all, starts_all = cs[:n_pairs], cs[n_pairs:] - (world - 1) * n_pairs
    if not (np.diff(starts_all) >= 0).all():
        raise ValueError("gather_to_writer: rank ranges are not in rank order")
    offs = np.zeros(n_pairs + 1, np.int64)
    np.cumsum(counts_all, out=offs[1:])
    bounds = np.concatenate([starts_all, [n_pairs]])
    per_rank_total = np.array([offs[bounds[r + 1]] - offs[bounds[r]] for r in range(world)], np.int64)

    cols = 3 if with_dist else 2
    max_m = max(int(per_rank_total.max()), 1)
    m_local = int(local_counts.sum())
    pin = dev.type == "cuda"
    # rank-padded payload; the padding rows are never read (receivers slice by the counts), so no zero fill
    stage = _host_buffer(max_m, cols, pin, slot="send")
    if m_local:
        sv = stage.numpy()
        sv[:m_local, 0:2] = qt_local
        if with_dist:
            sv[:m_local, 2] = np.ascontiguousarray(local_dist, dtype=np.float32).view(np.int32)
    send = stage.to(dev, non_blocking=pin) if pin else stage.clone()
    if rank == dst:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, recv, dst=dst, group=group)
        M = int(offs[-1])
        out = _host_buffer(M, cols, pin, slot="recv")
        at = 0
        for r in range(world):
            k = int(per_rank_total[r])
            out[at:at + k].copy_(recv[r][:k], non_blocking=pin)
            at += k
        if pin:
            torch.cuda.synchronize(dev)
        allm = out.numpy()   # a view of the reused staging buffer
        if with_dist:
            return offs, np.array(allm[:, 0:2], dtype=np.int32, order="C"), np.array(allm[:, 2], dtype=np.int32).view(np.float32)
        # (q, t) only: the staging buffer IS the result -- no copy of 100+ MB; valid until the next call
        return offs, allm, None
    dist.gather(send, None, dst=dst, group=group)
    return offs, None, None


class ShardedMatcher:
    """All-pairs matching over the ranks of a torch.distributed process group.

    match_fn(pairs_subset) -> (offsets, qt, dist) defaults to the GPU context's match_pairs;
    the CPU (gloo) tests inject a stand-in to exercise partition + gather without a GPU."""

    def __init__(self, ctx=None, match_fn=None, group=None, device=None, force_collectives=False, **match_kw):
        if ctx is None and match_fn is None:
            raise ValueError("need a GPU context or a match_fn")
        self.ctx = ctx
        self.group = group
        self.device = device
        self.match_kw = match_kw
        self.force_collectives = force_collectives
        self.match_fn = match_fn if match_fn is not None else (lambda p: ctx.match_pairs(p, **match_kw))

    def _rank_world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def match_local(self, pairs, n_rows):
        rank, world = self._rank_world()
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        mine = partition_pairs(pairs, n_rows, world)[rank]
        offs, qt, d = self.match_fn(pairs[mine])
        return mine, offs, qt, d

    def match_all(self, pairs, n_rows):
        """Every rank returns the complete result for `pairs` (global order)."""
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        mine, offs, qt, d = self.match_local(pairs, n_rows)
        return gather_matches(mine, offs, qt, d, len(pairs), group=self.group, device=self.device)

    def match_to_writer(self, pairs, n_rows, dst=0, with_dist=False):
        """The CLI flow: global offsets everywhere, the match lists only on rank `dst` (the SQLite writer)."""
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        mine, offs, qt, d = self.match_local(pairs, n_rows)
        return gather_to_writer(mine, offs, qt, d, len(pairs), dst=dst, group=self.group, device=self.device,
                                with_dist=with_dist, force_collectives=self.force_collectives)
