#!/usr/bin/env python3
"""Stand-in for bench.py's supervised measurement (tests/test_bench_supervisor.py): takes the
arguments the supervisor passes its child and behaves as FAKE_CHILD says, per rank --
  ok            prints a result line (rank 0) and exits 0
  rccl_fails    exits 3 under --exchange rccl (FAKE_FAIL_RANKS: only on those ranks), fine under host
  rccl_hangs    sleeps under --exchange rccl (the supervisor's timeout must end it), fine under host
  always_fails  exits 4 whatever the exchange
  meet          like ok, but the ranks first MEET the way the real child's do: a process group (gloo here) over the
                MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE the supervisor handed down, one all_gather of the ranks;
                FAKE_FAIL_RANKS then names the ranks whose first (rccl) attempt exits 3 before meeting
No GPU, no library: the supervisor's logic is what runs."""
import json
import os
import sys
import time

args = sys.argv[1:]
assert "--child" in args, args
exchange = [args[i + 1] for i, v in enumerate(args) if v == "--exchange"][-1]
note = ([args[i + 1] for i, v in enumerate(args) if v == "--exchange-note"] or [None])[-1]
rank = int(os.environ.get("RANK", "0"))
mode = os.environ.get("FAKE_CHILD", "ok")
fail_ranks = [int(v) for v in os.environ.get("FAKE_FAIL_RANKS", "").split(",") if v] or None
hit = fail_ranks is None or rank in fail_ranks
if mode == "always_fails":
    sys.exit(4)
if exchange == "rccl" and hit:
    if mode == "rccl_fails":
        sys.stderr.write("fake child: RCCL exchange timed out on shard 0\n")
        sys.exit(3)
    if mode == "rccl_hangs":
        time.sleep(120)
met = None
if mode == "meet":
    if exchange == "rccl" and fail_ranks is not None and rank in fail_ranks:
        sys.stderr.write("fake child: RCCL exchange timed out on shard 0\n")
        sys.exit(3)
    import datetime
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (a rank of the first attempt that died leaves its peers waiting here: the short timeout is what a wedged collective
    # looks like to the supervisor -- a child that exits non-zero)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(os.environ.get("FAKE_MEET_TIMEOUT", "20"))))
    mine = torch.tensor([rank], dtype=torch.int64)
    everyone = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(everyone, mine)
    met = sorted(int(t.item()) for t in everyone)
    dist.destroy_process_group()
print("some banner a library prints on stdout")
if rank == 0:
    gpus = ([int(args[i + 1]) for i, v in enumerate(args) if v == "--gpus"] or [1])[-1]
    scaling = ([args[i + 1] for i, v in enumerate(args) if v == "--scaling"] or ["strong"])[-1]
    print(json.dumps({"metric": "fake", "value": 1.0, "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
                      "gpus_argument": gpus, "scaling": scaling, "rccl_ranks": len(met) if met is not None else None, "met": met,
                      "steps": ([int(args[i + 1]) for i, v in enumerate(args) if v == "--steps"] or [None])[-1],
                      "warmup": ([int(args[i + 1]) for i, v in enumerate(args) if v == "--warmup"] or [None])[-1],
                      "config": {"exchange": exchange, "exchange_note": note, "master_port": os.environ.get("MASTER_PORT"),
                                 "agent_store": os.environ.get("TORCHELASTIC_USE_AGENT_STORE")}}))
