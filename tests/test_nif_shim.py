"""The erl_nif shim (integration/c_src/vettore_gpu_nif.c) cannot be built here -- the image has
no Erlang/OTP -- but it must not drift from the C ABI unnoticed:
  * it compiles (-fsyntax-only -Wall -Wextra -Werror) against include/vettore_flat.h and a
    stand-in erl_nif.h that declares the documented erl_nif calls it makes (tests/stubs/);
  * every vt_* function it calls is declared in include/vettore_flat.h and called with the
    declared number of arguments;
  * the Elixir stubs (integration/lib/vettore/gpu/nifs.ex) and the ErlNifFunc table agree on
    names and arities.
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "c_src", "vettore_gpu_nif.c")
HEADER = os.path.join(ROOT, "include", "vettore_flat.h")
STUBS = os.path.join(ROOT, "tests", "stubs")
EX_NIFS = os.path.join(ROOT, "integration", "lib", "vettore", "gpu", "nifs.ex")


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def split_args(argtext):
    depth, args, cur = 0, [], ""
    for ch in argtext:
        if ch == "," and depth == 0:
            args.append(cur)
            cur = ""
            continue
        depth += ch in "([{"
        depth -= ch in ")]}"
        cur += ch
    if cur.strip():
        args.append(cur)
    return args


def call_sites(text, name):
    out = []
    for m in re.finditer(r"\b%s\s*\(" % re.escape(name), text):
        i, depth = m.end(), 1
        while depth:
            depth += text[i] == "("
            depth -= text[i] == ")"
            i += 1
        out.append(split_args(text[m.end():i - 1]))
    return out


def test_shim_compiles_against_the_header():
    res = subprocess.run(["cc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + STUBS,
                          "-I" + os.path.join(ROOT, "include"), SHIM], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_every_library_call_matches_its_declaration():
    header = strip_comments(open(HEADER).read())
    shim = strip_comments(open(SHIM).read())
    declared = {}
    for m in re.finditer(r"\b(vt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        args = m.group(2).strip()
        declared[m.group(1)] = 0 if args in ("", "void") else len(split_args(args))
    used = sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", shim)))
    assert len(used) >= 18
    for name in used:
        assert name in declared, name + " is not declared in include/vettore_flat.h"
        for args in call_sites(shim, name):
            assert len(args) == declared[name], (name, len(args), declared[name])


def test_elixir_stubs_and_nif_table_agree():
    shim = strip_comments(open(SHIM).read())
    table = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"([a-z0-9_]+)",\s*(\d+),\s*[a-z0-9_]+,', shim)}
    ex = open(EX_NIFS).read()
    stubs = {}
    for m in re.finditer(r"def ([a-z0-9_]+)\(([^)]*)\), do: :erlang\.nif_error", ex):
        stubs[m.group(1)] = len([a for a in m.group(2).split(",") if a.strip()])
    assert table == stubs and len(table) >= 14
