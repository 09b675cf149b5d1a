"""The driver keeps ~9 KB of bench.py's stdout tail and parses the LAST line: round 4's line had grown to 22.7 KB and the
record came back `parsed: null`.  bench.py now ends with one compact line (target 4 KB, cap 8 KB); everything
else goes to bench_detail.json and an earlier stdout line.  Checked here without a GPU on real inputs: the full lines the
round-3 and round-4 runs printed (profiles/), and a worst case padded with long strings."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL_LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[34]_bench_default_run*.json")))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _check(line: dict, text: str):
    import bench
    assert len(text) < bench.LINE_TARGET, len(text)
    for k in REQUIRED:
        assert k in line, k
    cfg = line["config"]
    for k in ("workload", "reads_per_gpu", "parallelism", "oracle_mismatching_reads"):
        assert k in cfg, k
    assert len(cfg) <= 24 and all(not isinstance(v, (dict, list)) for v in cfg.values())
    for k in ("bound", "achieved", "peak", "unit", "frac", "frac_fetched", "algo_over_peak", "traffic"):
        assert k in line["roofline"], k
    assert all(not isinstance(v, (dict, list)) for v in line["roofline"].values())
    if line["cpu_baseline"] is not None:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in line["cpu_baseline"], k


@pytest.mark.parametrize("path", FULL_LINES, ids=os.path.basename)
def test_compact_line_of_earlier_rounds_full_lines(path):
    import bench
    full = json.load(open(path))
    assert len(json.dumps(full)) > 9000          # these are the lines the driver could not keep
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    _check(json.loads(text), text)
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["steps"] == full["steps"]
    assert line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert line["config"]["oracle_mismatching_reads"] == 0


def test_compact_line_worst_case_stays_under_the_cap():
    import bench
    full = json.load(open(FULL_LINES[-1]))
    full["config"]["workload"] = "x" * 5000
    full["config"]["parallelism"] = "y" * 500
    full["roofline"].update(frac_fetched=0.73, algo_over_peak=1.1, note="n" * 3000, frac_measured_on="m" * 900, traffic=238413278869,
                            fetched_bytes_per_launch=233138354688)
    full["cpu_baseline"]["sample"] = "s" * 4000
    full["other_workloads"] = full["other_workloads"] + [{"workload": f"w{i}", "error": "e" * 900} for i in range(40)]
    text = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(text) < bench.LINE_TARGET, len(text)
    assert json.loads(text)["roofline"]["traffic_over_fetched"] == pytest.approx(1.0226, abs=1e-3)


def test_bench_prints_detail_first_and_the_compact_record_last():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emit-from", FULL_LINES[-1]], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 2 and lines[0].startswith("bench_detail: ")
    assert json.loads(lines[0][len("bench_detail: "):]) == json.load(open(FULL_LINES[-1]))
    assert len(lines[-1]) < 8192
    _check(json.loads(lines[-1]), lines[-1])
    # what the driver does: keep the tail of stdout, parse the last line
    tail = p.stdout[-9000:]
    assert json.loads(tail.splitlines()[-1])["metric"].startswith("Mreads/s classified")


@pytest.mark.gpu
def test_real_run_ends_with_a_short_parsable_line():
    env = dict(os.environ, GANON_BENCH_EXTRAS="hibf_tiny")   # (one child process stands for the five of a default run)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-e2e", "--cpu-sample", "20000"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.splitlines()
    assert len(lines[-1]) < 8192 and any(ln.startswith("bench_detail: ") for ln in lines[:-1])
    line = json.loads(lines[-1])
    _check(line, lines[-1])
    assert line["value"] > 0 and line["config"]["oracle_mismatching_reads"] == 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] in ("port", "reference")
    r = line["roofline"]
    assert 0 < r["frac"] and 0 < r["frac_fetched"] and r["algo_over_peak"] >= r["frac_fetched"] * 0.99
    assert os.path.exists(os.path.join(ROOT, "bench_detail.json"))


@pytest.mark.gpu
def test_reference_binary_leg_with_our_binary_standing_in():
    # a real ganon-classify on PATH is timed on the same reads and an .ibf of the same bits, its .all diffed with the
    # GPU's matches, and reported as kind "reference".  No SeqAn3 build exists in this image: $GANON_REFERENCE_CLASSIFY points the
    # probe at this repo's binary, which takes the same command line and prints the same timing line.
    import bench_cpu
    assert bench_cpu.find_reference_classify()[0] is None or "mi355x" not in bench_cpu.find_reference_classify()[0]
    env = dict(os.environ, GANON_REFERENCE_CLASSIFY=os.path.join(ROOT, "ganon_amd", "host", "ganon-classify"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-e2e", "--no-extra"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.splitlines()[-1])
    cb = line["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["agrees_with_ours"] is True and cb["value"] > 0 and cb["port_value"] > 0, cb
    assert "--threads" in cb["sample"] and "== the GPU's matches" in cb["sample"]
