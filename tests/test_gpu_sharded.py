"""Two OS processes, each with its own row shard resident on the GPU (both on cuda:0 -- the
build image's boxes have one GPU), exchanging their per-shard hits over gloo and merging by
(rank key, id bytes): the host exchange path of vettore_amd/sharded.py end to end with real
shard searches.  The merged result must equal the oracle's search over all rows."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outq):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        import torch  # first: shares its HIP runtime with the library
        import torch.distributed as dist
        import oracle
        from vettore_amd import nifs
        from vettore_amd.sharded import ShardedFlat
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ok, checked = True, 0
        for metric, d in ((2, 48), (0, 24), (3, 16)):
            rng = np.random.default_rng(300 + metric)
            n = 3000
            x = np.round(rng.uniform(-1, 1, size=(n, d)) * 16).astype(np.float32) / 16   # exact ties across shards
            x[n // 2 + 7] = x[11]                                                       # a verbatim copy in the other shard
            if metric == 2:
                x = np.stack([oracle.normalize_l2(r) for r in x])
            ids = [b"doc-%d" % (i + 1) for i in range(n)]
            lo, hi = rank * n // world, (rank + 1) * n // world
            ref = nifs._flat_new(metric)
            assert nifs.flat_load_matrix(ref, ids[lo:hi], x[lo:hi]) == ("ok", ())
            sf = ShardedFlat(ref, dist, None)          # CPU tensors through gloo; the shard search runs on the GPU
            packed = oracle.pack_ids(ids)
            for qi in range(5):
                q = x[11] if qi == 0 else rng.uniform(-1, 1, d).astype(np.float32)
                for limit in (1, 10, 40):
                    got = sf.search(q, limit)
                    want = oracle.matrix_search(metric, x, packed, q, limit)
                    ok &= [(g[0], np.float32(g[1]).tobytes()) for g in got] == \
                          [(w[0], np.float32(w[1]).tobytes()) for w in want]
                    checked += 1
            del ref
        outq.put((rank, ok, checked))
    except Exception as e:  # surface the failure instead of letting the parent time out
        outq.put((rank, False, repr(e)))
        raise
    finally:
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass


def test_two_processes_two_gpu_shards_merge_to_the_single_index():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(o[0] for o in outs) == [0, 1]
    assert all(o[1] is True for o in outs) and all(o[2] == 45 for o in outs), outs
