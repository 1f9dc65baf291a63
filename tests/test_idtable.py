"""The host id table (vettore_amd/csrc/host/vt_idtable.h) is plain C++: built here with g++ and
driven against std::unordered_map through the shard's own sequence of calls (find-or-insert of
appended rows, swap-delete, growth), with AddressSanitizer and UBSan on."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_id_table_against_unordered_map():
    exe = os.path.join(tempfile.mkdtemp(), "idtable_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                           os.path.join(ROOT, "tests", "idtable_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.stdout, out.stderr[-2000:])
