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
    assert len(cfg) <= 4 + bench.MAX_EXTRA and all(not isinstance(v, (dict, list)) for v in cfg.values())
    assert line["roofline"]["traffic_measured_in_this_run"] is False and "traffic_from" in line["roofline"]
    if "ranks" in line:
        assert all(not isinstance(v, (dict, list)) for v in line["ranks"].values())
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


def test_compact_line_of_an_eight_rank_run_carries_the_proof_and_the_scalars_of_every_workload():
    """ranks_seen / devices / per-rank times, the headline at the binary's default cutoff, every other
    workload's SURVEY 8(d) fraction and the HIBF levels' line fractions are in the driver's record itself"""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_run9_detail.json")))
    full["n_gpus"] = 8
    full["ranks"] = {"ranks_seen": 8, "backend": "rccl", "launcher": "torchrun/env", "distinct_devices": 8, "shared_gpu_dry_run": False,
                     "devices": ",".join(f"0000:{i:02x}:00/0123456789ab" for i in range(8)), "ms_per_step_min": 41.2, "ms_per_step_max": 45.1,
                     "ms_per_step_by_rank": ",".join(["42.11"] * 8), "exchange_ms_max": 3.2}
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    _check(json.loads(text), text)
    assert line["n_gpus"] == 8 and line["ranks"]["ranks_seen"] == 8 and line["ranks"]["distinct_devices"] == 8
    assert line["ranks"]["devices"].count(",") == 7
    c = line["config"]
    assert c["flat8g_cutoff0.2_mreads_s"] == full["variants"]["rel_cutoff_0.2"]["mreads_per_s"]
    assert c["flat8g_wrapper_defaults_mreads_s"] == full["variants"]["wrapper_defaults_device_filter_matches"]["mreads_per_s"]
    assert 0.2 < c["hibf64k_frac"] < 0.3 and 0.35 < c["hibf64k_skew_frac"] < 0.45
    assert c["hibf64k_skew_level_line_fracs"].count("/") == 2
    assert line["roofline"]["traffic_from"].startswith("profiles/") and " " not in line["roofline"]["traffic_from"]


def _run_bench(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GANON_BENCH_ALLOW_SHARED_GPU")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_gpus_n_without_n_devices_fails_loudly_instead_of_measuring_fewer():
    """`--gpus N` used to be parsed and never read.  Without a launcher it now starts N ranks itself -- and
    refuses when the node has fewer GPUs (this container has none)"""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs here: the refusal cannot be provoked")
    p = _run_bench(["--gpus", "2", "--workload", "tiny"])
    assert p.returncode == 2 and p.stdout == "", (p.returncode, p.stdout[-300:])
    assert "--gpus 2 but" in p.stderr and "refusing" in p.stderr


def test_gpus_n_must_equal_the_launchers_world_size():
    p = _run_bench(["--gpus", "2", "--workload", "tiny"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "4", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert p.returncode == 2 and p.stdout == ""
    assert "--gpus 2 but the launcher started WORLD_SIZE=4" in p.stderr


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


@pytest.mark.gpu
def test_gpus_2_starts_two_ranks_itself_and_proves_them():
    """`python bench.py --gpus 2`, no torchrun: on this one-GPU box only with the explicit dry-run override (two ranks share the
    GPU and talk gloo -- RCCL refuses duplicate devices); the record says n_gpus 2, ranks_seen 2 and that the GPU was shared.
    The in-job extras stand for flat128g / slice1t: the read-sharded and the bin-range partitioned path with the exchange step."""
    args = ["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-e2e"]
    p = _run_bench(args, timeout=600)
    assert p.returncode == 2 and "refusing" in p.stderr and p.stdout.strip() == "", (p.returncode, p.stderr[-1500:])
    p = _run_bench(args, {"GANON_BENCH_ALLOW_SHARED_GPU": "1", "GANON_BENCH_EXTRAS": "tiny,slice_tiny"}, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.splitlines()[-1])
    _check(line, p.stdout.splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["oracle_mismatching_reads"] == 0
    rk = line["ranks"]
    assert rk["ranks_seen"] == 2 and rk["launcher"] == "self" and rk["distinct_devices"] == 1 and rk["shared_gpu_dry_run"] is True
    assert rk["backend"] == "gloo" and rk["ms_per_step_by_rank"].count(",") == 1 and rk["ms_per_step_min"] <= rk["ms_per_step_max"]
    assert line["config"]["slice_tiny_exchange_ms"] > 0 and line["config"]["other_workloads_mismatching_reads"] == 0


@pytest.mark.gpu
def test_gpus_2_under_torchrun_and_a_wrong_gpus_flag():
    """the driver's own launch line (torch.distributed.run, one rank per GPU) with the flag it passes -- and with a flag that
    does not match the launcher's world size, which must fail instead of printing a line for another N"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env.update(GANON_BENCH_ALLOW_SHARED_GPU="1", GANON_BENCH_EXTRAS="tiny")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29611",
            os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-e2e"]
    p = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"]["ranks_seen"] == 2 and line["ranks"]["launcher"] == "torchrun/env"
    p = subprocess.run(base + ["--gpus", "3"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode != 0 and "--gpus 3 but the launcher started WORLD_SIZE=2" in p.stderr
    assert not any(ln.startswith('{"metric"') for ln in p.stdout.splitlines())
