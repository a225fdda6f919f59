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
