"""tools/ holds at most 40 files and every one of them answers --help under today's library, without a GPU
(the one-off A/B scripts of rounds 1-5 live in tools/attic/, unmaintained)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

TOOLS = os.path.join(ROOT, "tools")


def kept():
    out = []
    for base, dirs, files in os.walk(TOOLS):
        dirs[:] = [d for d in dirs if d not in ("attic", "__pycache__")]
        out += [os.path.join(base, f) for f in files]
    return sorted(out)


def test_tools_directory_is_small():
    files = kept()
    assert len(files) <= 40, len(files)
    assert all(f.endswith((".py", ".sh", ".hip")) for f in files), files


@pytest.mark.parametrize("path", [f for f in kept() if f.endswith((".py", ".sh"))], ids=lambda p: os.path.basename(p))
def test_every_tool_answers_help(path):
    cmd = [sys.executable, path, "--help"] if path.endswith(".py") else ["bash", path, "--help"]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (path, r.stderr[-400:])
    assert len(r.stdout.strip()) > 40, (path, r.stdout)


def test_profile_summary_reports_the_busy_time_per_launch_of_overlapping_kernels(tmp_path):
    """tools/summarize_prof.py: with independent frames in flight on two streams every launch of the frame kernel lasts about twice the
    per-frame time; `busy_union_ns_per_call` (time during which at least one launch runs / launches) is the trace's counterpart of the
    bench line's launch_us.  Synthetic trace: 4 launches of 50 ns, pairwise overlapping by half, and one kernel that never overlaps."""
    import csv
    import json
    trace = tmp_path / "trace"
    trace.mkdir()
    with open(trace / "out_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        for t0 in (0, 25, 50, 75):
            w.writerow({"Kernel_Name": "k_overlapping", "Start_Timestamp": 1000 + t0, "End_Timestamp": 1000 + t0 + 50})
        for t0 in (0, 100):
            w.writerow({"Kernel_Name": "k_alone", "Start_Timestamp": 5000 + t0, "End_Timestamp": 5000 + t0 + 40})
    r = subprocess.run([sys.executable, os.path.join(TOOLS, "summarize_prof.py"), str(tmp_path)], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-400:]
    k = json.load(open(tmp_path / "summary.json"))["kernels"]
    assert k["k_overlapping"]["trace"]["avg_ns"] == 50 and k["k_overlapping"]["trace"]["busy_union_ns_per_call"] == 125 / 4
    assert abs(k["k_overlapping"]["trace"]["overlap_factor"] - 200 / 125) < 1e-12
    assert k["k_alone"]["trace"]["busy_union_ns_per_call"] == 40 and k["k_alone"]["trace"]["overlap_factor"] == 1.0
