#!/usr/bin/env python3
"""tools/env_table.py [--check] -- DESIGN_APPENDIX A.10's table of settings, generated from the three key lists of
vettore_amd/csrc/vt_env.h (name, how the environment string is read, default, what it does) and written between

    <!-- env:begin --> ... <!-- env:end -->

so that the document cannot name a switch the library does not have.  --check: exit 1 if the file is out
of date, or if the product list has grown past 30 entries (tests/test_docs.py runs that on the CPU box)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV_H = os.path.join(ROOT, "vettore_amd", "csrc", "vt_env.h")
DOC = os.path.join(ROOT, "DESIGN_APPENDIX.md")
MAX_PRODUCT_KEYS = 30

PARSE = {
    "P_FLAG": "flag (unset, empty or `0`: off)", "P_INT": "integer", "P_ORDER": "`pair` \\| `avx` \\| `seq` \\| `sse2`",
    "P_NOMINATE": "`f32` \\| `bf16`", "P_SHADOW": "`0` / `off` \\| anything else", "P_SLAB": "`malloc` \\| anything else",
    "P_EXCHANGE": "`host` \\| `rccl`", "P_NONE": "no variable: `vt_debug_set` only",
}


def keys_of(text, macro):
    m = re.search(r"#define %s\(X\)(.*?)\n\n" % macro, text, flags=re.S)
    if not m:
        raise SystemExit("vt_env.h has no %s list" % macro)
    out = []
    for key, name, parse, dflt, doc in re.findall(r'X\(([A-Z0-9_]+), "([a-z0-9_]+)", (P_[A-Z]+), (-?\d+), "((?:[^"\\]|\\.)*)"\)', m.group(1)):
        out.append({"key": key, "name": name, "parse": parse, "default": int(dflt), "doc": doc})
    return out


def lists():
    text = open(ENV_H).read()
    return keys_of(text, "VT_ENV_PRODUCT_KEYS"), keys_of(text, "VT_ENV_EXPERIMENT_KEYS"), keys_of(text, "VT_ENV_HOOK_KEYS")


def table():
    product, experiments, hooks = lists()
    rows = ["| Setting (`vt_debug_set` name) | Environment variable | Read as | Default | Effect |", "|---|---|---|---|---|"]

    def add(k, where):
        var = "—" if k["parse"] == "P_NONE" or where == "hooks" else "`VT_%s`" % k["name"].upper()
        rows.append("| `%s` | %s | %s | %d | %s |" % (k["name"], var, PARSE[k["parse"]] if where != "hooks" else "`vt_debug_set` only",
                                                   k["default"], k["doc"]))

    rows.append("| **`libvettore_hip.so` (%d settings)** | | | | |" % len(product))
    for k in product:
        add(k, "product")
    rows.append("| **`make experiments` only (`vettore_amd/lib/experiments/libvettore_hip.so`: wrong results on purpose)** | | | | |")
    for k in experiments:
        add(k, "experiments")
    rows.append("| **`libvettore_hip_hooks.so` only (fault injection for tests)** | | | | |")
    for k in hooks:
        add(k, "hooks")
    return "\n".join(rows)


def main():
    product, _, _ = lists()
    text = open(DOC).read()
    m = re.search(r"(<!-- env:begin -->\n)(.*?)(\n<!-- env:end -->)", text, flags=re.S)
    if not m:
        raise SystemExit("DESIGN_APPENDIX.md has no <!-- env:begin --> ... <!-- env:end --> markers")
    want = table()
    if "--check" in sys.argv:
        if len(product) > MAX_PRODUCT_KEYS:
            print("VT_ENV_PRODUCT_KEYS has %d entries (at most %d: every one is a path a maintainer must trust)" % (len(product), MAX_PRODUCT_KEYS))
            sys.exit(1)
        if m.group(2).strip() != want.strip():
            print("DESIGN_APPENDIX.md A.10 is out of date: run tools/env_table.py")
            sys.exit(1)
        return
    open(DOC, "w").write(text[:m.start(2)] + want + text[m.end(2):])
    print("%d product settings written to DESIGN_APPENDIX.md A.10" % len(product))


if __name__ == "__main__":
    main()
