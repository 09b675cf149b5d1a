"""One BASELINE workload at full size through the C ABI in THIS process, with the HIP runtime this process maps -- printed.
The GPU tests and bench.py run inside a PyTorch process, whose bundled libamdhip64 (7.0.2) is mapped first and therefore serves
libganon_hip.so too; the product binaries map /opt/rocm's (7.2.0), which is what the library is linked against (RUNPATH).  Round 3's
silent zero came from exactly that difference.  Without --with-torch this script never imports torch, so the library gets the runtime
it was linked against; tests/test_runtime72.py runs it both ways and compares the match checksums.

  python scripts/runtime_check.py <flat8g|hibf64k|flat128g|tiny|hibf_tiny|...> [--with-torch] [--reads N]   -> one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mapped(name: str):
    out = []
    for ln in open("/proc/self/maps"):
        p = ln.split()[-1]
        if name in os.path.basename(p) and p not in out:
            out.append(p)
    return out


def main() -> int:
    args = sys.argv[1:]
    with_torch = "--with-torch" in args
    reads = int(args[args.index("--reads") + 1]) if "--reads" in args else 0
    name = [a for a in args if not a.startswith("--") and not a.isdigit()][0]
    if with_torch:
        import torch  # noqa: F401 -- maps torch/lib/libamdhip64.so first, as pytest and bench.py do
    import numpy as np

    import bench
    import bench_workload as bw
    import ganon_amd
    assert with_torch or "torch" not in sys.modules, "something imported torch"
    spec = dict(bench.WORKLOADS[name])
    n = reads or spec["reads"]
    if spec["kind"] == "hibf":
        fill = ganon_amd.FILL_3_OF_16 if spec.get("fill") == "3/16" else 0
        if spec.get("skew"):
            wl, flt = bw.make_hibf_skew_device_workload(ganon_amd, name, spec["user_bins"], spec["h"], n, seed=42, rows_scale=spec.get("rows_scale", 1.0), fill=fill)
        else:
            wl, flt = bw.make_hibf_device_workload(ganon_amd, name, spec["user_bins"], spec["tmax"], spec.get("rows_top", spec["rows"]), spec["rows"],
                                                   spec["h"], n, seed=42, fill=fill or 1)
        off2 = None
    else:
        wl = bw.make_device_flat_workload(name, spec["bins"], spec["rows"], spec["h"], n, spec["paired"], seed=42, read_len=spec.get("read_len", 150))
        flt, _ = bw.device_filter(ganon_amd, wl, 0)
        off2 = wl.off2
    st = ganon_amd.HipStream(flt, n, wl.bases.size, max_matches=n * 2)
    st.upload(wl.bases, wl.off, off2)
    st.classify(wl.k, wl.w, 0.75)
    nh, status, mo, m = st.fetch()
    tm = st.timings()
    hip = mapped("libamdhip64")
    print(json.dumps({"workload": name, "reads": n, "with_torch": with_torch, "libamdhip64": hip, "libganon_hip": mapped("libganon_hip"),
                      "checksum": f"{bw.checksum_matches(m):016x}", "matches": int(len(m)), "n_hashes": int(nh.sum(dtype=np.uint64)),
                      "ms_count": round(tm["ms_count"], 3)}), flush=True)
    st.destroy()
    flt.free()
    return 0


if __name__ == "__main__":
    sys.exit(main())
