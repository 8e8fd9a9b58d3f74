"""tools/ and tools/experiments/ are measurement scripts run by hand on the GPU box (profiles/README.md says which file came from which); nothing
else in the suite imports them.  This keeps them from rotting silently: every Python script compiles, every shell script
parses, and every script a README / DESIGN / profile names exists."""
import glob
import os
import py_compile
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "experiments", "*.py")))
SH = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")) + glob.glob(os.path.join(ROOT, "tools", "experiments", "*.sh")))


@pytest.mark.parametrize("path", PY, ids=[os.path.relpath(p, ROOT) for p in PY])
def test_python_tool_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


@pytest.mark.parametrize("path", SH, ids=[os.path.relpath(p, ROOT) for p in SH])
def test_shell_tool_parses(path):
    assert subprocess.run(["bash", "-n", path], capture_output=True, text=True).returncode == 0


def test_tools_named_in_the_docs_exist():
    named = set()
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md", os.path.join("profiles", "README.md"), os.path.join("tools", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        named.update(re.findall(r"tools/(?:experiments/)?[A-Za-z0-9_]+\.(?:py|sh|hip)", text))
    missing = sorted(n for n in named if not os.path.exists(os.path.join(ROOT, n)))
    assert not missing, missing
    assert len(named) > 20
