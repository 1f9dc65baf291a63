"""Row-sharded flat search across the GPUs of one node.

Two exchange paths, same result:

* device path (`enable_device_exchange`, used on GPUs): every shard's id_rank
  column is its slice of ONE ordering of all ids (computed once, after loading),
  so the shards' u64 candidate keys compare directly.  Per query the library
  enqueues scan + select into a device block without waiting, ONE all_gather
  queued on the same HIP stream collects the blocks of all shards, a small merge
  kernel picks the global top-k and the host waits once.  While the scan runs the
  host is already enqueueing the collective, so the exchange adds only the
  collective's own device time.
* host path (always available; CPU/gloo tests): per-shard hits travel as 64-byte
  records and are merged by (rank key, id bytes).


One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  Each
rank owns a contiguous block of rows in its own flat index.  A query runs on
every shard; the per-shard top-k lists -- already in the reference order
(rank.total_cmp, id bytes) -- are exchanged with ONE all_gather of fixed-size
records and merged identically on every rank.  The merge compares the f32 rank
key first and the id bytes second (flat.rs:34-40), so the result equals a
single index over all rows, ties included; no global id-rank is needed.

The payload is world * limit * 64 B (5 KiB at 8 GPUs, limit 10): latency-bound,
so a single collective per query and no second round trip in the common case.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np

REC = 64          # bytes per record
MAX_ID = REC - 12  # id bytes carried inline


def pack_hits(hits: List[Tuple[bytes, float, int]], limit: int) -> np.ndarray:
    """[(id, raw, rank_key)] -> uint8[(limit + 1) * REC]; record 0 is the header
    (count, overflow flag)."""
    buf = np.zeros((limit + 1, REC), dtype=np.uint8)
    head = buf[0].view(np.uint32)
    head[0] = len(hits)
    for i, (id_, raw, key) in enumerate(hits):
        rec = buf[i + 1]
        w = rec[:12].view(np.uint32)
        w[0] = key
        w[1] = np.float32(raw).view(np.uint32)
        w[2] = len(id_)
        if len(id_) > MAX_ID:
            head[1] = 1  # ids travel in a second (object) exchange
        else:
            rec[12:12 + len(id_)] = np.frombuffer(id_, dtype=np.uint8)
    return buf.reshape(-1)


def unpack_hits(buf: np.ndarray, limit: int):
    buf = buf.reshape(limit + 1, REC)
    head = buf[0].view(np.uint32)
    out = []
    for i in range(int(head[0])):
        rec = buf[i + 1]
        w = rec[:12].view(np.uint32)
        n = int(w[2])
        id_ = bytes(rec[12:12 + n]) if n <= MAX_ID else None
        out.append((id_, float(w[1:2].view(np.float32)[0]), int(w[0])))
    return out, bool(head[1])


def merge_shards(per_rank: List[List[Tuple[bytes, float, int]]], limit: int) -> List[Tuple[bytes, float]]:
    """k-way merge by (rank key, id bytes) == FlatHit::cmp (flat.rs:34-40)."""
    allhits = [h for hits in per_rank for h in hits]
    allhits.sort(key=lambda h: (h[2], h[0]))
    return [(h[0], h[1]) for h in allhits[:limit]]


def gather_global_ranks(dist, world: int, local_ids_packed):
    """Collective: every rank contributes the ids of its rows (row order); returns
    (all ids blob, offsets, first global row of each rank, rank of every global row
    in bytewise id order -- FlatHit::cmp's tie-break, flat.rs:34-40)."""
    from . import nifs
    blob, off = local_ids_packed
    parts = [None] * world
    dist.all_gather_object(parts, (bytes(blob), np.asarray(off, dtype=np.uint64)))
    counts = [len(p[1]) - 1 for p in parts]
    bases = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    all_blob = b"".join(p[0] for p in parts)
    all_off = np.zeros(int(bases[-1]) + 1, dtype=np.uintp)
    pos, byte0 = 0, 0
    for p in parts:
        o = p[1].astype(np.uintp)
        all_off[pos:pos + len(o)] = o + byte0
        pos += len(o) - 1
        byte0 += len(p[0])
    ranks = nifs.rank_ids((all_blob, all_off))
    return all_blob, all_off, bases, ranks


class ShardedFlat:
    """`ref` is this rank's FlatRef; `dist` is torch.distributed (initialised)
    or None for a single shard.  `local_search(query, limit)` may be injected
    (tests run the exchange on CPU/gloo with a stand-in shard)."""

    def __init__(self, ref, dist=None, device=None,
                 local_search: Optional[Callable] = None, force_exchange: bool = False):
        self.ref, self.dist, self.device = ref, dist, device
        self.force_exchange = force_exchange  # run the collective even on one rank (measures its cost)
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self._local = local_search
        self._torch = None
        self._bufs = {}
        self._dev = None  # state of the device exchange path
        if dist is not None:
            import torch
            self._torch = torch

    # ---------------------------------------------------------- device exchange
    def enable_device_exchange(self, local_ids_packed, max_limit: int = 64):
        """Collective: gathers every shard's ids, ranks them once, installs this
        shard's slice as its id_rank column.  `local_ids_packed` = (bytes, offsets)
        of this rank's rows in row order.  Valid until a shard inserts a new id or deletes a row:
        the library then marks that shard's blocks stale and every rank falls back to the host
        exchange together (see _search_device)."""
        from . import nifs
        torch = self._torch
        all_blob, all_off, bases, ranks = gather_global_ranks(self.dist, self.world, local_ids_packed)
        mine = ranks[bases[self.rank]:bases[self.rank + 1]]
        res = nifs.flat_set_id_ranks(self.ref, mine)
        if res != "ok":
            raise RuntimeError(res)
        block_bytes = 16 + max_limit * 16
        stream = torch.cuda.ExternalStream(nifs.flat_stream(self.ref), device=self.device)
        self._dev = {
            "blob": all_blob, "off": all_off, "bases": bases, "max_limit": max_limit, "block_bytes": block_bytes,
            "stream": stream,
            "local": torch.zeros(block_bytes, dtype=torch.uint8, device=self.device),
            "gathered": torch.zeros(self.world * block_bytes, dtype=torch.uint8, device=self.device),
            "bufs": nifs.MergeBuffers(),
        }

    def _search_device(self, query, limit):
        from . import nifs
        torch, d = self._torch, self._dev
        res = nifs.flat_search_begin(self.ref, query, limit, d["local"].data_ptr())
        if res != "ok":
            raise RuntimeError(res[1])
        with torch.cuda.stream(d["stream"]):  # the collective queues behind the shard's kernels
            self.dist.all_gather_into_tensor(d["gathered"], d["local"])
        res = nifs.flat_merge_gathered(self.ref, d["gathered"].data_ptr(), self.world, limit, d["block_bytes"], d["bufs"])
        if res[0] != "ok":
            if "stale id ranks" in res[1]:
                # Some shard was mutated after enable_device_exchange: its keys no longer compare with
                # the others' and rows moved under the gathered id table.  The stale shard marked its
                # block, every rank merged the same blocks and lands here together: all of them drop
                # the device path and answer this query (and the following ones) over the host path,
                # which needs no global ranks.  enable_device_exchange may be called again later.
                self._dev = None
                return None
            raise RuntimeError(res[1])
        b, off, blob, bases = d["bufs"], d["off"], d["blob"], d["bases"]
        out = []
        for i in range(res[1]):
            g = int(bases[int(b.shard[i])]) + int(b.rows[i])
            out.append((blob[int(off[g]):int(off[g + 1])], float(b.raw[i])))
        return out

    def _local_search(self, query, limit):
        if self._local is not None:
            return self._local(query, limit)
        from . import nifs
        res = nifs.flat_search_with_keys(self.ref, query, limit)
        if res[0] != "ok":
            raise RuntimeError(res[1])
        return res[1]

    def _buffers(self, limit):
        """Per-limit exchange buffers: a pinned host send block and the gathered block."""
        buf = self._bufs.get(limit)
        if buf is None:
            torch = self._torch
            on_gpu = self.device is not None and getattr(self.device, "type", "cpu") != "cpu"
            send_host = torch.zeros(((limit + 1), REC), dtype=torch.uint8)
            if on_gpu:
                send_host = send_host.pin_memory()
            send_dev = torch.empty_like(send_host, device=self.device) if on_gpu else send_host
            gathered = torch.empty(self.world * (limit + 1) * REC, dtype=torch.uint8, device=send_dev.device)
            buf = self._bufs[limit] = (send_host, send_host.numpy(), send_dev, gathered, on_gpu)
        return buf

    def search_batch(self, queries, limit: int) -> List[List[Tuple[bytes, float]]]:
        """flat_search_batch over all shards (BASELINE configs[3]'s batched leg: 16 x 256 queries):
        every rank answers the whole batch on its rows (one shared pass per <= 256 queries), the ranks'
        hit lists meet in ONE all_gather of fixed-size wire blocks (nq x (limit + 1) x 64 B per rank:
        180 KB at 256 queries, limit 10) and are merged per query by (rank key, id bytes) in the
        library (vt_hit_blocks_merge).  `local_search_batch` may stand in for the shard (CPU tests)."""
        from . import nifs
        q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32))
        nq = q.shape[0]
        if self.dist is None or (self.world == 1 and not self.force_exchange):
            if self._local is not None:
                return [[(h[0], h[1]) for h in self._local(v, limit)] for v in q]
            res = nifs.flat_search_batch(self.ref, q, limit)
            if res[0] != "ok":
                raise RuntimeError(res[1])
            return res[1]
        torch = self._torch
        on_gpu = self.device is not None and getattr(self.device, "type", "cpu") != "cpu"
        key = ("batch", nq, limit)
        buf = self._bufs.get(key)
        if buf is None:
            send_host = torch.zeros((nq, limit + 1, REC), dtype=torch.uint8)
            if on_gpu:
                send_host = send_host.pin_memory()
            send_dev = torch.empty_like(send_host, device=self.device) if on_gpu else send_host
            gathered = torch.empty(self.world * nq * (limit + 1) * REC, dtype=torch.uint8, device=send_dev.device)
            buf = self._bufs[key] = (send_host, send_host.numpy(), send_dev, gathered)
        send_host, send_np, send_dev, gathered = buf
        long_ids = None
        if self._local is None:
            res = nifs.flat_search_batch_blocks(self.ref, q, limit, send_np)
            if res[0] != "ok":
                raise RuntimeError(res[1])
            long_ids = res[1]
        else:
            for i in range(nq):
                hits = self._local(q[i], limit)
                send_np[i] = pack_hits(hits, limit).reshape(limit + 1, REC)
                if any(len(h[0]) > MAX_ID for h in hits):
                    long_ids = long_ids or [None] * nq
                    long_ids[i] = [h[0] for h in hits]
        if on_gpu:
            send_dev.copy_(send_host, non_blocking=True)
        self.dist.all_gather_into_tensor(gathered, send_dev.reshape(-1))
        host = gathered.cpu().numpy().reshape(self.world, nq, limit + 1, REC)
        any_long = bool(host[:, :, 0, 4:8].view(np.uint32).any())
        if not any_long:
            merged = nifs.hit_blocks_merge(host, self.world, nq, limit)
            return [nifs.unpack_block(merged[i]) for i in range(nq)]
        # ids longer than a record: a second (object) exchange carries them whole, the merge runs here
        objs = [None] * self.world
        if long_ids is None:
            long_ids = [None] * nq
        mine = [long_ids[i] if long_ids[i] is not None else [h[0] for h in unpack_hits(send_np[i].reshape(-1), limit)[0]]
                for i in range(nq)]
        self.dist.all_gather_object(objs, mine)
        out = []
        for i in range(nq):
            per_rank = []
            for r in range(self.world):
                h, _ = unpack_hits(host[r, i].reshape(-1), limit)
                per_rank.append([(objs[r][i][j], x[1], x[2]) for j, x in enumerate(h)])
            out.append(merge_shards(per_rank, limit))
        return out

    def search(self, query, limit: int) -> List[Tuple[bytes, float]]:
        if self.dist is None or (self.world == 1 and not self.force_exchange):
            return [(h[0], h[1]) for h in self._local_search(query, limit)]
        if self._dev is not None and 0 < limit <= self._dev["max_limit"]:
            out = self._search_device(query, limit)
            if out is not None:
                return out
        send_host, send_np, send_dev, gathered, on_gpu = self._buffers(limit)
        long_ids = None
        if self._local is None:
            # the library serialises its hits straight into the pinned send block
            from . import nifs
            res = nifs.flat_search_packed(self.ref, query, limit, send_np[1:])
            if res[0] != "ok":
                raise RuntimeError(res[1])
            count, long_ids = res[1]
            head = send_np[0].view(np.uint32)
            head[0] = count
            head[1] = 1 if long_ids is not None else 0
        else:
            hits = self._local(query, limit)
            send_np[:] = pack_hits(hits, limit).reshape(limit + 1, REC)
            if any(len(h[0]) > MAX_ID for h in hits):
                long_ids = [h[0] for h in hits]
        if on_gpu:
            send_dev.copy_(send_host, non_blocking=True)
        self.dist.all_gather_into_tensor(gathered, send_dev.reshape(-1))
        host = gathered.cpu().numpy().reshape(self.world, -1)
        per_rank, any_long = [], False
        for r in range(self.world):
            h, flag = unpack_hits(host[r], limit)
            per_rank.append(h)
            any_long |= flag
        if any_long:  # ids longer than MAX_ID bytes: second exchange carries them whole
            objs = [None] * self.world
            self.dist.all_gather_object(objs, long_ids if long_ids is not None else [h[0] for h in per_rank[self.rank]])
            per_rank = [[(objs[r][i], h[1], h[2]) for i, h in enumerate(per_rank[r])] for r in range(self.world)]
        return merge_shards(per_rank, limit)
