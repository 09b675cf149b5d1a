"""The host binaries read their environment in ONE place (ganon_amd/host/tunables.hpp): round 5 had 36 getenv()
sites, 15 of them in classify.cpp.  Checked here: no other site reads a knob of ours; every GANON_* name the host sources mention is
in the table; `--verbose` says under which knobs a run was started (the oracle-backed twin of the binary shares all of the host code,
so this runs without a GPU); a knob still does what it did (a batch size of 3 reads changes no output byte)."""
import glob
import os
import re
import subprocess

import pytest

import cli_util as cu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "ganon_amd", "host")
OTHER_TOOLS = {"LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY", "GCOV_PREFIX", "LLVM_PROFILE_FILE", "ASAN_OPTIONS", "LSAN_OPTIONS",
               "TSAN_OPTIONS"}   # (startup.hpp fast_exit: variables of profilers / coverage / sanitizers, not knobs of ours)


def _sources():
    return sorted(glob.glob(os.path.join(HOST, "*.cpp")) + glob.glob(os.path.join(HOST, "*.hpp")))


def test_the_environment_is_read_in_one_place():
    table = set(re.findall(r'\{ "(GANON_[A-Z0-9_]+)"', open(os.path.join(HOST, "tunables.hpp")).read()))
    assert len(table) >= 25
    for path in _sources():
        text = open(path).read()
        for m in re.finditer(r'getenv\("([A-Z0-9_]+)"\)', text):
            assert os.path.basename(path) == "startup.hpp" and m.group(1) in OTHER_TOOLS, (path, m.group(0))
        if os.path.basename(path) != "tunables.hpp":
            assert "getenv(knob_info" not in text
        for name in re.findall(r"\$?(GANON_(?:HOST|DEVICE|PARTITION|HIP)_?[A-Z0-9_]*)", text):
            if name.rstrip("_") in ("GANON_HOST", "GANON_HIP", "GANON_DEVICE") and name.rstrip("_") not in table:
                continue   # (prose like "$GANON_HOST_*")
            assert name in table, (os.path.basename(path), name)


@pytest.fixture(scope="module")
def twin():
    return cu.build_oracle_binary()


def test_verbose_lists_the_knobs_and_a_knob_still_works(twin, kat, tmp_path):
    files = cu.KatFiles(kat, str(tmp_path / "kat"))
    case = next(c for c in kat["cases"] if c["single"] and len(c["single"]) >= 2) if any(c["single"] and len(c["single"]) >= 2 for c in kat["cases"]) else kat["cases"][0]
    outs = {}
    for tag, env in (("plain", {}), ("small", {"GANON_HOST_BATCH_READS": "3", "GANON_HOST_POST_THREADS": "2"})):
        prefix = str(tmp_path / tag)
        args = [a for a in files.case_args(case, prefix) if a != "--quiet"] + ["--verbose"]
        e = {k: v for k, v in os.environ.items() if not k.startswith("GANON_")}
        e.update(env)
        p = subprocess.run([twin] + args, capture_output=True, text=True, timeout=300, env=e)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stderr.splitlines() if ln.startswith("[host tunables]")]
        assert len(line) == 1, p.stderr[-2000:]
        if env:
            assert "GANON_HOST_BATCH_READS=3" in line[0] and "GANON_HOST_POST_THREADS=2" in line[0]
        else:
            assert line[0].strip() == "[host tunables] none set (defaults)"
        outs[tag] = {ext: open(prefix + ext).read() for ext in (".rep", ".all", ".unc", ".sta") if os.path.exists(prefix + ext)}
    assert outs["plain"].keys() == outs["small"].keys() and ".rep" in outs["plain"]
    for ext in outs["plain"]:
        if ext != ".sta":   # (.sta carries nothing that depends on batching either, but is compared where it is the subject: test_cli_kat)
            assert outs["plain"][ext] == outs["small"][ext], ext
