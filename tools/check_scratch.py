#!/usr/bin/env python3
"""tools/check_scratch.py [--no-scratch] RESOURCES PATTERN [PATTERN ...] -- build-time guard (Makefile).

RESOURCES is what hipcc printed under -Rpass-analysis=kernel-resource-usage for one translation
unit.  Every kernel whose (mangled) name contains one of the PATTERNs must report `VGPRs Spill: 0`:
those kernels issue global loads from inline asm whose data lands in the named registers only
at a hand-placed s_waitcnt (K1m's FAST path) or keep 128 accumulators beside a DMA ring whose
waits are hand-counted (K2b) -- a vector register the allocator spilled in between would be
silently wrong, and its scratch traffic would sit in the counted queue.  (Scratch
as such is not the test: K1m's 16 bytes per lane hold spilled SGPRs and no vector data.)  Exit status 1 names the offenders; a pattern that matches
no kernel at all is an error too (the guard must not rot).

--no-scratch: the matching kernels must also report `ScratchSize [bytes/lane]: 0` (K1, K1m, K1p since r05: their
only scratch user was the out-parameter of the overflow recovery -- 16 bytes per lane that made every launch set up
a scratch segment).
"""
import re
import sys


def main():
    args = sys.argv[1:]
    no_scratch = bool(args) and args[0] == "--no-scratch"
    if no_scratch:
        args = args[1:]
    text = open(args[0], errors="replace").read()
    patterns = args[1:]
    kernels = {}
    scratch_bytes = {}
    name = None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and name:
            kernels[name] = int(m.group(1))
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            scratch_bytes[name] = int(m.group(1))
    bad, seen = [], {p: 0 for p in patterns}
    for k, scratch in kernels.items():
        for p in patterns:
            if p in k:
                seen[p] += 1
                if scratch:
                    bad.append((k, scratch))
                if no_scratch and scratch_bytes.get(k, 0):
                    print("check_scratch: %s uses %d bytes of scratch per lane" % (k, scratch_bytes[k]), file=sys.stderr)
                    bad.append((k, 0))
    for p, n in seen.items():
        if n == 0:
            print("check_scratch: no kernel matches %r in %s" % (p, args[0]), file=sys.stderr)
            return 1
    for k, scratch in bad:
        if scratch:
            print("check_scratch: %s spills %d vector registers" % (k, scratch), file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
