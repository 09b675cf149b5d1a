"""Summarises gpurun_out/calib (scripts/calib_gather.sh) into profiles/r03_calib_gather.json: the gather rate per table size
and row width, and per counter pass the counter total per launch of gather_kernel against the known algorithmic bytes."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "calib")


def main():
    timing = [json.loads(l) for l in open(os.path.join(OUT, "timing.jsonl")) if l.strip().startswith("{")]
    res = {"timing": timing, "counters": []}
    for d in sorted(glob.glob(os.path.join(OUT, "pmc_*_*_*"))):
        if not os.path.isdir(d):
            continue
        m = re.match(r"pmc_(.+)_(\d+)_(\d+)$", os.path.basename(d))
        if not m:
            continue
        counters, table, row = m.group(1).split("+"), int(m.group(2)), int(m.group(3))
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        tot, launches = {}, set()
        for f in files:
            for r in csv.DictReader(open(f)):
                if "gather_kernel" not in r.get("Kernel_Name", ""):
                    continue
                launches.add(r.get("Dispatch_Id"))
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        log = os.path.join(OUT, os.path.basename(d) + ".log")
        line = None
        if os.path.exists(log):
            for l in open(log):
                if l.strip().startswith("{"):
                    line = json.loads(l)
        n = max(1, len(launches))
        entry = {"table_mib": table, "row_bytes": row, "launches": len(launches), "per_launch": {k: v / n for k, v in tot.items()}}
        if line:
            entry["algorithmic_bytes_per_launch"] = line["algorithmic_bytes_per_launch"]
            entry["row_requests_per_launch"] = line["row_requests_per_launch"]
            if "FETCH_SIZE" in tot:
                kib = tot["FETCH_SIZE"] / n
                entry["FETCH_SIZE_bytes_x1"] = kib * 1024
                entry["FETCH_SIZE_x1_over_algorithmic"] = kib * 1024 / line["algorithmic_bytes_per_launch"]
                entry["FETCH_SIZE_x1_over_128B_lines"] = kib * 1024 / (line["row_requests_per_launch"] * max(128, row))
        res["counters"].append(entry)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r03_calib_gather.json"), "w") as f:
        json.dump(res, f, indent=1)
    for t in timing:
        print(f"table {t['table_mib']:>5} MiB row {t['row_bytes']:>3} B U={t['loads_in_flight_per_lane']:>2} bpc={t['blocks_per_cu']}: "
              f"{t['ms_avg']:8.3f} ms  {t['algorithmic_gbs']:8.1f} GB/s algorithmic  {t['line128_gbs']:8.1f} GB/s in 128-B lines")
    for c in res["counters"]:
        print(json.dumps(c))


if __name__ == "__main__":
    sys.exit(main())
