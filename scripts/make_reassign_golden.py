"""Golden vectors for f-4 (`ganon reassign`, the EM over .all) made by the REFERENCE'S OWN code.

Build container only (it reads /root/reference): loads /root/reference/src/ganon/reassign.py with importlib -- its only
import is ganon.util, which imports nothing but the standard library; both modules are executed from where they lie, nothing
of them is copied -- and runs reassign(cfg) on .rep/.all inputs.  Inputs and outputs are committed under
tests/golden/reassign/<case>/ as data: in.rep, in[.<hierarchy>].all, cfg.json, out.rep, out[.<hierarchy>].one, log.txt.

Inputs come from two places:
  * this repo's ganon-classify (the oracle-backend twin, which runs without a GPU and is byte-identical to the HIP binary:
    tests/test_cli_kat.py) on the reference's own 98-pair read fixture against the synthetic 40-target database of the tests,
    in the default line order and with --reference-order, one and two hierarchy levels, --output-single;
  * seeded synthetic .all/.rep texts that reach what those do not: ties, no unique read at all, reads listed in two places,
    a threshold above zero, --max-iter 0 / 1, targets only in .rep, three-digit counts.

usage: python scripts/make_reassign_golden.py        (rewrites tests/golden/reassign/)
"""
import contextlib
import importlib.util
import io
import json
import os
import shutil
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/ganon"
OUT = os.path.join(ROOT, "tests", "golden", "reassign")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def load_reference_reassign():
    pkg = types.ModuleType("ganon")
    pkg.__path__ = []  # a namespace to hang ganon.util on; ganon/__init__.py needs installed package metadata
    sys.modules["ganon"] = pkg
    for name in ("util", "reassign"):
        spec = importlib.util.spec_from_file_location(f"ganon.{name}", os.path.join(REF, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"ganon.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
    return sys.modules["ganon.reassign"]


def run_reference(ref, case_dir, cfg):
    """runs in case_dir with relative prefixes so that the log holds no absolute path"""
    ns = types.SimpleNamespace(input_prefix=["in"], output_prefix="out", max_iter=cfg.get("max_iter", 10), threshold=cfg.get("threshold", 0),
                               remove_all=False, skip_one=False, skip_rep=False, quiet=False, verbose=False)
    cwd = os.getcwd()
    os.chdir(case_dir)
    err = io.StringIO()
    try:
        with contextlib.redirect_stderr(err):
            ok = ref.reassign(ns)
    finally:
        os.chdir(cwd)
    with open(os.path.join(case_dir, "log.txt"), "w") as f:
        f.write(err.getvalue())
    with open(os.path.join(case_dir, "cfg.json"), "w") as f:
        json.dump(dict(max_iter=ns.max_iter, threshold=ns.threshold, returned=bool(ok)), f)
    return ok


def synthetic(seed, n_reads, n_targets, p_unique, max_matches, hier=None, ties=False, split_reads=False, rep_only=2, big_counts=False):
    """-> {filename: text}.  .all lines `read \\t target \\t count`, .rep rows for every target that occurs (+ a few that do not)"""
    rng = np.random.default_rng(seed)
    hier = hier or [""]
    files = {}
    rep_rows = []
    total = 0
    for hname in hier:
        names = [f"{hname or 'T'}{rng.integers(100, 999)}.{i}" for i in range(n_targets)]
        weights = np.ones(n_targets) if ties else rng.random(n_targets) ** 3 + 0.01
        weights /= weights.sum()
        lines, later = [], []
        direct = {n: 0 for n in names}
        unique = {n: 0 for n in names}
        lca = {n: 0 for n in names}
        for r in range(n_reads):
            m = 1 if rng.random() < p_unique else int(rng.integers(2, max_matches + 1))
            ts = rng.choice(n_targets, size=min(m, n_targets), replace=False, p=weights)
            rid = f"{hname}read{r}" + ("/1" if r % 7 == 0 else "")
            cs = [int(rng.integers(1, 400 if big_counts else 30)) for _ in ts]
            rec = [f"{rid}\t{names[t]}\t{c}\n" for t, c in zip(ts, cs)]
            for t in ts:
                direct[names[t]] += 1
            if len(ts) == 1:
                unique[names[ts[0]]] += 1
            else:
                lca[names[ts[0]]] += 1  # (any node would do: reassign only carries `unique` over)
            if split_reads and len(rec) > 1 and r % 5 == 0:
                lines.append(rec[0])
                later.extend(rec[1:])
            else:
                lines.extend(rec)
        lines.extend(later)
        total += n_reads
        files["in.all" if hname == "" else f"in.{hname}.all"] = "".join(lines)
        label = hname or "H1"
        for i, n in enumerate(names):
            if direct[n] or i % 3 == 0:
                row = [label, n, str(direct[n]), str(unique[n]), str(lca[n])]
                if i % 2 == 0:
                    row += ["species", f"name of {n}"]
                elif i % 5 == 0:
                    row += ["assembly"]
                rep_rows.append("\t".join(row) + "\n")
        for j in range(rep_only):
            rep_rows.append(f"{label}\tLCA{j}\t0\t0\t{j + 1}\tgenus\tnode {j}\n")
    files["in.rep"] = "".join(rep_rows) + f"#total_classified\t{total}\n#total_unclassified\t{7}\n"
    return files


def classify_cases(tmp):
    """inputs produced by this repo's ganon-classify (oracle backend) from tests/golden/sim.{1,2}.fq.gz"""
    import cli_util as cu
    import ganon_fixtures as gf
    import test_cli_kat as tk
    binary = cu.build_oracle_binary()
    os.makedirs(os.path.join(tmp, "db"))
    db = tk.make_sim_db(os.path.join(tmp, "db"))
    pairs = db["fq1"] + "," + db["fq2"]
    base = ["--tax", db["tax"], "--paired-reads", pairs, "--output-all", "--output-lca", "--quiet", "--rel-cutoff", "0.25", "--rel-filter", "0.1"]
    out = {}

    def grab(name, args, prefix):
        cu.run(binary, args)
        d = os.path.dirname(prefix)
        files = {}
        for fn in sorted(os.listdir(d)):
            if fn.startswith(os.path.basename(prefix)) and (fn.endswith(".rep") or fn.endswith(".all")):
                files["in" + fn[len(os.path.basename(prefix)):]] = open(os.path.join(d, fn)).read()
        out[name] = files

    for name, extra in (("sim_default", []), ("sim_reference_order", ["--reference-order", "--threads", "1"])):
        p = os.path.join(tmp, name, "x")
        os.makedirs(os.path.dirname(p))
        grab(name, ["--ibf", db["ibf"], "-o", p] + base + extra, p)
    # what `ganon classify --multiple-matches em` (the wrapper's default) runs the binary with: classify.py:50-52
    p = os.path.join(tmp, "sim_em_mode", "x")
    os.makedirs(os.path.dirname(p))
    grab("sim_em_mode", ["--ibf", db["ibf"], "-o", p, "--skip-lca"] + [a for a in base if a != "--output-lca"], p)
    # two hierarchy levels: the database cut in two filters (targets 0-7 / 8-39), second level sees what the first left
    import oracle
    halves = []
    names = list(db["targets"])
    for h, sel in enumerate((names[:8], names[8:])):
        built = gf.build_ibf({n: db["targets"][n] for n in sel}, 19, 31, max_fp=0.05, filter_size=0.0)
        path = os.path.join(tmp, f"half{h}.ibf")
        gf.write_ibf(path, built)
        halves.append(path)
    for name, extra in (("sim_two_levels", []), ("sim_two_levels_single", ["--output-single"])):
        p = os.path.join(tmp, name, "x")
        os.makedirs(os.path.dirname(p))
        grab(name, ["--ibf", ",".join(halves), "--tax", db["tax"] + "," + db["tax"], "--hierarchy-labels", "1_first,2_second", "-o", p] +
             [a for a in base if a not in ("--tax", db["tax"])] + extra, p)
    return out


def main():
    import tempfile
    ref = load_reference_reassign()
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    cases = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, files in classify_cases(tmp).items():
            cases[name] = (files, {})
    cases["syn_skewed"] = (synthetic(1, 400, 12, 0.4, 4), {})
    cases["syn_ties"] = (synthetic(2, 300, 6, 0.3, 5, ties=True), {})
    cases["syn_no_unique"] = (synthetic(3, 120, 8, 0.0, 3), {})
    cases["syn_split_reads"] = (synthetic(4, 250, 10, 0.5, 4, split_reads=True), {})
    cases["syn_threshold"] = (synthetic(5, 500, 20, 0.3, 6), {"threshold": 0.05})
    cases["syn_max_iter_1"] = (synthetic(5, 500, 20, 0.3, 6), {"max_iter": 1})
    cases["syn_unbounded"] = (synthetic(6, 600, 25, 0.2, 8, big_counts=True), {"max_iter": 0})
    cases["syn_two_levels"] = (synthetic(7, 200, 9, 0.4, 4, hier=["lvlA", "lvlB"]), {})
    cases["syn_all_unique"] = (synthetic(8, 80, 5, 1.0, 2), {})
    cases["syn_many_targets"] = (synthetic(9, 3000, 400, 0.35, 12), {"max_iter": 25})
    for name, (files, cfg) in cases.items():
        d = os.path.join(OUT, name)
        os.makedirs(d)
        for fn, text in files.items():
            with open(os.path.join(d, fn), "w") as f:
                f.write(text)
        ok = run_reference(ref, d, cfg)
        print(name, "ok" if ok else "returned False", sorted(os.listdir(d)))


if __name__ == "__main__":
    main()
