"""Multi-GPU sharding of the all-pairs job: one process per GPU, torch.distributed (RCCL) exchange at the end.

The reference is single-process; image pairs are independent
(/root/reference/src/Feature/FeatureMatching.cpp:14), so the pair list is partitioned and the
only exchange step moves the per-pair match lists to the rank that writes them:

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
    job is ~7 ms and a scatter of 2.5 M matches in NumPy would cost several times that."""
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


def _device_buffer(rows, cols, dev, slot):
    """Reused device tensors of the exchange step (the writer's receive buffer, a rank's send buffer)."""
    import torch
    key = (slot, cols, str(dev))
    buf = _staging.get(key)
    if buf is None or buf.shape[0] < rows:
        buf = torch.empty((max(rows, 1) * 5 // 4 + 1024, cols), dtype=torch.int32, device=dev)
        _staging[key] = buf
    return buf[:rows]


def range_bounds(parts, n_pairs):
    """partition_pairs' contiguous ranges -> bounds[world + 1]; rank r owns pairs [bounds[r], bounds[r+1]).
    Every rank computes the same partition, so the bounds need no exchange."""
    b = np.zeros(len(parts) + 1, np.int64)
    for r, p in enumerate(parts):
        if len(p) > 1 and not (np.diff(p) == 1).all():
            raise ValueError("exchange needs contiguous per-rank pair ranges (partition_pairs)")
        b[r + 1] = b[r] + len(p)
        if len(p) and int(p[0]) != int(b[r]):
            raise ValueError("exchange needs the rank ranges in rank order (partition_pairs)")
    if int(b[-1]) != int(n_pairs):
        raise ValueError("rank ranges do not cover the pair list")
    return b


def gather_to_writer(bounds, local_offs, local_payload, dst=0, group=None, device=None, force_collectives=False, slot=""):
    """The exchange step of the CLI flow: only the rank that owns the SQLite handle needs the lists.

    bounds        : range_bounds(...) -- the same on every rank, no exchange needed (an empty range in the middle,
                    which partition_pairs produces when one pair outweighs total / world, is just bounds[r] == bounds[r+1]).
    local_offs    : CSR offsets of this rank's pairs (len = pairs of the rank + 1).
    local_payload : int32 tensor [m_local, cols] ON `device` (cols = 2: (queryIdx, trainIdx) -- what the `matches` table
                    stores, Database.cpp:631-654; cols = 3 adds the distance bits).  With RCCL it is the device tensor
                    the library filled (msfm_fetch_matches_device): the lists never visit the host on the sender.
    -> (global offsets int64[P+1] on every rank, payload int32[M, cols] as a NumPy view of a reused page-locked
        host buffer on rank `dst` -- valid until the next call WITH THE SAME `slot` (the pipelined super-batches alternate two) --
        and None elsewhere).

    One all_reduce(SUM) of the per-pair counts (disjoint supports: a concatenation), then every non-writer rank with
    matches SENDS its block and the writer RECEIVES each block at its final position of one [M, cols] buffer: no rank
    padding, no reassembly pass, one device-to-host copy on the writer.  MBs per rank: latency-bound on xGMI."""
    import torch
    import torch.distributed as dist

    n_pairs = int(bounds[-1])
    local_counts = np.diff(np.asarray(local_offs, np.int64))
    dev = device if device is not None else torch.device("cpu")
    multi = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    cols = int(local_payload.shape[1])
    if len(local_counts) != int(bounds[rank + 1] - bounds[rank]):
        raise ValueError("local result does not match this rank's range")
    counts = torch.zeros(n_pairs, dtype=torch.int32)
    counts[int(bounds[rank]):int(bounds[rank + 1])] = torch.from_numpy(local_counts.astype(np.int32))
    if world > 1 or (force_collectives and multi):
        counts = counts.to(dev)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        counts = counts.cpu()
    offs = np.zeros(n_pairs + 1, np.int64)
    np.cumsum(counts.numpy(), out=offs[1:])
    blk = offs[np.asarray(bounds)]                      # block r = rows [blk[r], blk[r+1]) of the global payload
    pin = dev.type == "cuda"
    M = int(offs[-1])
    # P2POp's peer is a GLOBAL rank; r / dst are ranks of `group`
    peer = (lambda r: dist.get_global_rank(group, r)) if (multi and group is not None) else (lambda r: r)
    if rank == dst:
        out = _device_buffer(M, cols, dev, "recv" + slot) if pin else _host_buffer(M, cols, False, "recv" + slot)
        ops = [dist.P2POp(dist.irecv, out[int(blk[r]):int(blk[r + 1])], peer(r), group)
               for r in range(world) if r != dst and blk[r + 1] > blk[r]]
        if ops:
            reqs = dist.batch_isend_irecv(ops)
        if blk[rank + 1] > blk[rank]:
            out[int(blk[rank]):int(blk[rank + 1])].copy_(local_payload, non_blocking=pin)
        if ops:
            for q in reqs:
                q.wait()
        if pin:
            host = _host_buffer(M, cols, True, "host" + slot)
            host.copy_(out, non_blocking=True)
            torch.cuda.synchronize(dev)
            return offs, host.numpy()
        return offs, out.numpy()
    if blk[rank + 1] > blk[rank]:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_payload, peer(dst), group)]):
            q.wait()
    return offs, None


class ShardedMatcher:
    """All-pairs matching over the ranks of a torch.distributed process group.

    With a GPU context the match lists stay in HBM between the matcher and the exchange (exchange on `device` =
    the context's GPU, backend nccl = RCCL); `device` = cpu (gloo: the CPU tests, or several ranks sharing one GPU)
    stages them through the library's page-locked host buffers instead.  match_fn(pairs_subset) -> (offsets, qt,
    dist) replaces the GPU context in the CPU tests."""

    def __init__(self, ctx=None, match_fn=None, group=None, device=None, force_collectives=False, **match_kw):
        if ctx is None and match_fn is None:
            raise ValueError("need a GPU context or a match_fn")
        self.ctx = ctx
        self.group = group
        self.device = device
        self.match_kw = match_kw
        self.force_collectives = force_collectives
        self._own_fn = match_fn is None   # the GPU context's own matcher: the lists can stay in HBM for the exchange
        self.match_fn = match_fn if match_fn is not None else (lambda p: ctx.match_pairs(p, **match_kw))
        self.last = {}   # per-rank timing of the last match_to_writer call (ms): compute, exchange

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

    def _compute_local(self, pairs, n_rows, with_dist, slot=""):
        """This rank's share of `pairs` through the matcher, its lists as the send tensor of the exchange.
        -> (bounds, local offsets, payload int32[m, cols] on the exchange device, timing dict)"""
        import time
        import torch
        rank, world = self._rank_world()
        parts = partition_pairs(pairs, n_rows, world)
        bounds = range_bounds(parts, len(pairs))
        mine = parts[rank]
        dev = self.device if self.device is not None else torch.device("cpu")
        cols = 3 if with_dist else 2
        t0 = time.perf_counter()
        fetch_ms = 0.0
        if self._own_fn and dev.type == "cuda":
            # lists stay in HBM: the library copies them device-to-device into the send tensor
            kw = dict(self.match_kw)
            kw.pop("fetch", None)
            offs, _, _ = self.ctx.match_pairs(pairs[mine], fetch=False, **kw)
            m = int(offs[-1])
            tf = time.perf_counter()
            if with_dist:
                qt_t = _device_buffer(m, 2, dev, "send_qt" + slot)
                d_t = _device_buffer(m, 1, dev, "send_d" + slot)
                self.ctx.fetch_matches_device(qt_t.data_ptr() if m else 0, d_t.data_ptr() if m else 0)
                payload = torch.cat([qt_t, d_t], dim=1)
            else:
                payload = _device_buffer(m, 2, dev, "send_qt" + slot)
                self.ctx.fetch_matches_device(payload.data_ptr() if m else 0, 0)
            # (the send tensor is torch's allocation, filled by the library through the system HIP runtime: the copy's rate is
            # recorded -- msfm_fetch_matches_device returns when it has completed -- so that a slow cross-runtime path shows up)
            fetch_ms = (time.perf_counter() - tf) * 1e3
        else:
            offs, qt, d = self.match_fn(pairs[mine])
            m = int(offs[-1])
            host = np.empty((m, cols), np.int32)
            host[:, 0:2] = np.asarray(qt, np.int32).reshape(-1, 2)
            if with_dist:
                host[:, 2] = np.ascontiguousarray(d, dtype=np.float32).view(np.int32)
            payload = torch.from_numpy(host).to(dev)
        t1 = time.perf_counter()
        info = {"compute_ms": (t1 - t0) * 1e3, "local_pairs": int(len(mine)), "local_matches": m,
                "fetch_device_ms": fetch_ms, "fetch_device_bytes": m * 4 * cols if fetch_ms else 0}
        return bounds, offs, payload, info

    def match_to_writer(self, pairs, n_rows, dst=0, with_dist=False):
        """The CLI flow: global offsets everywhere, the match lists only on rank `dst` (the SQLite writer).
        -> (offsets, qt int32[M, 2] or None, dist float32[M] or None); qt / dist are views of a reused host buffer."""
        import time
        import torch
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        dev = self.device if self.device is not None else torch.device("cpu")
        bounds, offs, payload, info = self._compute_local(pairs, n_rows, with_dist)
        t1 = time.perf_counter()
        goffs, allm = gather_to_writer(bounds, offs, payload, dst=dst, group=self.group, device=dev,
                                       force_collectives=self.force_collectives)
        info["exchange_ms"] = (time.perf_counter() - t1) * 1e3
        self.last = info
        if allm is None:
            return goffs, None, None
        if with_dist:
            return goffs, allm[:, 0:2], allm[:, 2].view(np.float32)
        return goffs, allm, None

    def match_to_writer_batches(self, pairs, n_rows, batch_pairs=65536, dst=0, sink=None, with_dist=False, pipelined=True):
        """The same flow in SUPER-BATCHES of at most `batch_pairs` pairs, for jobs whose lists do not fit memory (BASELINE's largest
        config: 6.9e9 matches = 83 GB): every super-batch is partitioned over the ranks, matched, exchanged, handed to
        `sink(first_pair, offsets, qt, dist)` on the writer rank -- offsets relative to the super-batch, the arrays views valid until
        the super-batch after the next -- and dropped.  What is resident at any time is two super-batches' lists: on a rank its
        share (in the library + one send tensor), on the writer the whole super-batch.  The reference streams the same way, one
        transaction per <= 100 pairs (/root/reference/src/Feature/FeatureMatching.cpp:13, 70-72, 118-139; the C++ drop-in:
        host/FeatureMatching.cpp).
        `pipelined` (default): the exchange of super-batch k -- all_reduce of the counts, sends to the writer, the writer's copy to
        the host -- runs on a second thread WHILE super-batch k + 1 is being matched (two sets of send / receive buffers, taken in
        turn); every rank issues its exchanges in the same order on that thread, so the collectives pair up as before.  The wall
        clock of a step is then ~ sum(compute) + the last exchange instead of sum(compute + exchange).
        -> per-pair match counts int64[P] on every rank; self.last sums the phases over the super-batches (`exchange_ms`: time
        the exchanges took on their thread; `exchange_wait_ms`: what of it the matching thread actually waited for)."""
        import time
        import torch
        from concurrent.futures import ThreadPoolExecutor
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        P = len(pairs)
        counts = np.zeros(P, np.int64)
        rank, _ = self._rank_world()
        dev = self.device if self.device is not None else torch.device("cpu")
        tot = {"compute_ms": 0.0, "exchange_ms": 0.0, "exchange_wait_ms": 0.0, "local_pairs": 0, "local_matches": 0, "super_batches": 0,
               "max_batch_matches": 0, "fetch_device_ms": 0.0, "fetch_device_bytes": 0, "pipelined": bool(pipelined)}
        step = max(1, int(batch_pairs))

        def exchange(k, bounds, offs, payload):
            if dev.type == "cuda":
                torch.cuda.set_device(dev)   # (the exchange thread's own current device)
            t = time.perf_counter()
            goffs, allm = gather_to_writer(bounds, offs, payload, dst=dst, group=self.group, device=dev,
                                           force_collectives=self.force_collectives, slot=str(k & 1))
            return goffs, allm, (time.perf_counter() - t) * 1e3

        def deliver(b0, n_sub, res):
            goffs, allm, ms = res
            counts[b0:b0 + n_sub] = np.diff(goffs)
            tot["exchange_ms"] += ms
            tot["max_batch_matches"] = max(tot["max_batch_matches"], int(goffs[-1]))
            if sink is not None and rank == dst:
                if with_dist:
                    sink(b0, goffs, allm[:, 0:2], allm[:, 2].view(np.float32))
                else:
                    sink(b0, goffs, allm, None)

        pool = ThreadPoolExecutor(1) if pipelined else None
        pending = None   # (first pair, pairs, future) of the exchange in flight
        try:
            for k, b0 in enumerate(range(0, P, step)):
                sub = pairs[b0:b0 + step]
                bounds, offs, payload, info = self._compute_local(sub, n_rows, with_dist, slot=str(k & 1))
                for key in ("compute_ms", "local_pairs", "local_matches", "fetch_device_ms", "fetch_device_bytes"):
                    tot[key] += info[key]
                tot["super_batches"] += 1
                if pool is None:
                    res = exchange(k, bounds, offs, payload)
                    tot["exchange_wait_ms"] += res[2]
                    deliver(b0, len(sub), res)
                    continue
                if pending is not None:
                    t = time.perf_counter()
                    res = pending[2].result()
                    tot["exchange_wait_ms"] += (time.perf_counter() - t) * 1e3
                    deliver(pending[0], pending[1], res)
                pending = (b0, len(sub), pool.submit(exchange, k, bounds, offs, payload))
            if pending is not None:
                t = time.perf_counter()
                res = pending[2].result()
                tot["exchange_wait_ms"] += (time.perf_counter() - t) * 1e3
                deliver(pending[0], pending[1], res)
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
        self.last = tot
        return counts
