import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
import oracle
from vettore_amd import nifs
oracle.build()

def bits(h): return [(x[0], np.float32(x[1]).tobytes()) for x in h]
rng = np.random.default_rng(5)
d = 16
m = 0
ref = nifs._flat_new(m)
nifs.flat_set_reduce_order(ref, 3)
o = oracle.FlatIndex(m)
live = []
log = []
for step in range(400):
    op = rng.integers(0, 10)
    if op < 5 or not live:
        cnt = int(rng.integers(1, 40))
        items = [("id-%d" % rng.integers(0, 600), rng.uniform(-1, 1, d).astype(np.float32)) for _ in range(cnt)]
        assert nifs.flat_insert_many(ref, items)[0] == "ok"
        o.insert_many(items)
        live = list({*live, *[i for i, _ in items]})
        log.append(("ins", [i for i, _ in items]))
    elif op < 8:
        victim = live.pop(int(rng.integers(0, len(live))))
        nifs.flat_delete(ref, victim); o.delete(victim)
        log.append(("del", victim))
    else:
        nifs.flat_delete(ref, "missing-%d" % step); o.delete("missing-%d" % step)
        log.append(("delmiss",))
    q = rng.uniform(-1, 1, d).astype(np.float32)
    k = int(rng.integers(1, 30))
    if len(o) == 0:
        continue
    got = nifs.flat_search(ref, q, k)[1]
    want = o.search(q, k)
    if bits(got) != bits(want):
        print("MISMATCH step", step, "k", k, "n", len(o))
        full_g = nifs.flat_search(ref, q, len(o))[1]
        full_o = o.search(q, len(o))
        print("full equal:", bits(full_g) == bits(full_o))
        gd = dict(full_g); od = dict(full_o)
        for i, (a, b) in enumerate(zip(got, want)):
            if bits([a]) != bits([b]):
                print(" first diff at", i, a, b, "gpu-full raw of wanted id:", gd.get(b[0]), "oracle raw of gpu id", od.get(a[0]))
                break
        print("last ops:", log[-4:])
        again = nifs.flat_search(ref, q, k)[1]
        print("again equal:", bits(again) == bits(want))
        break
else:
    print("no mismatch")
