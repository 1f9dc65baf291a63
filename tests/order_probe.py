"""Prints what `dot(a, b).to_bits()` of INTEGRATION.md section 4's probe gives
under each candidate lane order of wide::f32x8::reduce_add (computed by the
oracle).  A maintainer compares the reference build's output with this table.

    python tests/order_probe.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PROBE_A = [1.0e8, 1.0, -1.0e8, 3.0, 1.0e-3, 7.0, -9.0, 0.125]
PROBE_B = [1.0] * 8


def table():
    import oracle
    out = {}
    for name, order in (("PAIR", oracle.ORDER_PAIR), ("AVX", oracle.ORDER_AVX), ("SEQ", oracle.ORDER_SEQ),
                        ("SSE2", oracle.ORDER_SSE2)):
        oracle.set_reduce_order(order)
        v = oracle.compute(oracle.METRIC_CODE["inner_product"], PROBE_A, PROBE_B)
        out[name] = (int(np.float32(v).view(np.uint32)), float(v))
    oracle.set_reduce_order(oracle.DEFAULT_ORDER)
    return out


if __name__ == "__main__":
    for name, (bits, val) in table().items():
        print("%-4s bits = %#010x  value = %r" % (name, bits, val))
