"""The numbers the documents carry that can be checked without a GPU."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_quotes_the_collected_test_counts():
    """DESIGN.md's test counts are generated (tools/design_counts.py), not typed: VERDICT r4 found 261 where 262 ran."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_counts.py"), "--check"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr


def test_the_settings_table_is_the_librarys_and_stays_short():
    """DESIGN_APPENDIX A.10 is generated from csrc/vt_env.h (tools/env_table.py), and the product's list holds at most 30
    settings (every kept knob is a path a maintainer must trust; nifs.rs has none)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "env_table.py"), "--check"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr


def test_quoted_headline_ranges_include_the_drivers_own_runs():
    """README.md and DESIGN.md quote the headline as a range; every BENCH_rNN.json the driver has written must lie inside
    it (the documents quoted builder-box numbers no driver run had reached)."""
    # (only records that are part of the tree the documents were written against -- tracked files: a record the driver
    # writes AFTER the last commit is news for the next edit of the documents, not a unit-test failure -- ADVICE r5)
    tracked = subprocess.run(["git", "ls-files", "BENCH_r*.json"], cwd=ROOT, capture_output=True, text=True).stdout.split()
    values = []
    for name in sorted(tracked):
        if re.fullmatch(r"BENCH_r\d+\.json", name):
            rec = json.load(open(os.path.join(ROOT, name)))
            parsed = rec.get("parsed") or {}
            if parsed.get("unit") == "queries/s" and "value" in parsed:
                values.append((name, float(parsed["value"])))
    assert values, "no BENCH_rNN.json with a parsed headline"
    for doc in ("README.md", "DESIGN.md"):
        text = open(os.path.join(ROOT, doc)).read()
        m = re.search(r"<!-- headline:begin -->\s*\**(\d+(?:\.\d+)?)\D{1,3}(\d+(?:\.\d+)?)\s+queries/s", text)
        assert m, doc + " has no <!-- headline:begin --> LOW-HIGH queries/s marker"
        lo, hi = float(m.group(1)), float(m.group(2))
        for name, v in values:
            assert lo <= v <= hi, (doc, name, v, lo, hi)
