"""Golden vectors for the `.rep` consumer contract (second half): what the REFERENCE'S OWN `parse_rep`
(/root/reference/src/ganon/report.py:163-209, the first thing `ganon report` does with a classify run) makes of `.rep` files this
repository's ganon-classify wrote.

Build container only (reads /root/reference).  report.py cannot be imported as a module here (it imports multitax, which is not
installed), so the one function is taken out of the file's syntax tree and executed from there -- nothing of it is copied: what gets
committed under tests/golden/report/ is data, `<case>.rep` (our output) and `<case>.parse_rep.json` (the reference's reading of it,
with normalize False and True).

The .rep files come from the oracle-backend twin of the binary (same host code, runs without a GPU, byte-identical to the HIP binary:
tests/test_cli_kat.py) on the reference's 98-pair fixture against the tests' 40-target database: flat IBF with tax, HIBF, two hierarchy
levels, --skip-lca, and --output-single.

usage: python scripts/make_report_golden.py        (rewrites tests/golden/report/)"""
import ast
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/ganon/report.py"
OUT = os.path.join(ROOT, "tests", "golden", "report")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def reference_parse_rep():
    tree = ast.parse(open(REF).read(), REF)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "parse_rep"]
    assert len(fn) == 1, "report.py has no parse_rep"
    ns = {}
    exec(compile(ast.Module(body=fn, type_ignores=[]), REF, "exec"), ns)
    return ns["parse_rep"]


def main():
    import cli_util as cu
    import test_cli_kat as tk
    parse_rep = reference_parse_rep()
    binary = cu.build_oracle_binary()
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    with tempfile.TemporaryDirectory() as d:
        db = tk.make_sim_db(d)
        reads = ["--paired-reads", db["fq1"] + "," + db["fq2"]]
        common = ["--output-all", "--output-lca", "--quiet", "--rel-cutoff", "0.25", "--rel-filter", "0.1"]
        cases = {
            "flat_tax": ["--ibf", db["ibf"], "--tax", db["tax"]] + reads + common,
            "flat_skip_lca": ["--ibf", db["ibf"], "--skip-lca", "--output-all", "--quiet", "--rel-cutoff", "0.25"] + reads,
            "hibf_tax": ["--ibf", db["hibf"], "--hibf", "--tax", db["tax"]] + reads + common,
            "two_levels": ["--ibf", db["ibf"] + "," + db["ibf"], "--tax", db["tax"] + "," + db["tax"], "--hierarchy-labels", "1_first,2_second",
                           "--rel-cutoff", "0.6,0.25", "--output-all", "--output-lca", "--quiet"] + reads,
            "two_levels_single_output": ["--ibf", db["ibf"] + "," + db["ibf"], "--tax", db["tax"] + "," + db["tax"], "--hierarchy-labels", "1_first,2_second",
                                         "--rel-cutoff", "0.6,0.25", "--output-all", "--output-lca", "--output-single", "--quiet"] + reads,
            "single_end": ["--ibf", db["ibf"], "--tax", db["tax"], "--single-reads", db["fq1"]] + common,
        }
        for name, args in cases.items():
            prefix = os.path.join(d, name)
            cu.run(binary, args + ["-o", prefix])
            rep = os.path.join(OUT, name + ".rep")
            shutil.copy(prefix + ".rep", rep)
            out = {}
            for normalize in (False, True):
                reports, counts = parse_rep(rep, normalize)
                out["normalize_" + str(normalize).lower()] = {"reports": reports, "counts": counts}
            with open(os.path.join(OUT, name + ".parse_rep.json"), "w") as f:
                json.dump(out, f, indent=1, sort_keys=True)
            print(name, "levels", [k for k in out["normalize_false"]["counts"] if k != "total"], out["normalize_false"]["counts"]["total"])


if __name__ == "__main__":
    main()
