#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (tools/profile_bench.sh) to two small files:
  <tag>_kernel_stats.csv : per kernel name: calls, total / average / min / max duration (ns), share
  <tag>_traffic.json     : per kernel name: FETCH_SIZE and WRITE_SIZE per launch (raw KB), and the corrected
                           HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md section HBM:
                           on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes; WRITE_SIZE is exact),
                           with the grid size of the launches it was measured on."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def main(out, tag):
    stats = defaultdict(list)
    for r in rows(os.path.join(out, "trace", "**", "*kernel_trace.csv")):
        stats[re.sub(r"\s+", " ", r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in stats.values()) or 1
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
        for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % (name, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / total))
    pmc = defaultdict(lambda: defaultdict(list))
    grid = {}
    for sub in ("fetch", "write"):
        for r in rows(os.path.join(out, sub, "**", "*counter_collection.csv")):
            name = re.sub(r"\s+", " ", r["Kernel_Name"])
            pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid[name] = "%s x %s" % (r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
    traffic = {}
    for name, c in pmc.items():
        f, w = c.get("FETCH_SIZE", []), c.get("WRITE_SIZE", [])
        if not f or not w:
            continue
        # the largest launches of a kernel are the benchmark-sized ones (small ones come from plan building)
        fk, wk = sorted(f)[len(f) // 2:], sorted(w)[len(w) // 2:]
        fa, wa = sum(fk) / len(fk), sum(wk) / len(wk)
        traffic[name] = {"launches_fetch": len(f), "launches_write": len(w), "grid": grid[name],
                         "FETCH_SIZE_KB_upper_half_mean": round(fa, 1), "WRITE_SIZE_KB_upper_half_mean": round(wa, 1),
                         "FETCH_SIZE_KB_min_max": [round(min(f), 1), round(max(f), 1)],
                         "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
    # what was run: the bench line of the traced pass says which sizes these per-launch figures belong to
    meta = {"tag": tag}
    try:
        with open(os.path.join(out, "trace.stdout")) as f:
            line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        meta.update({"n": b["config"]["n"], "nnz": b["config"]["nnz"], "kernel": b["config"]["kernel"],
                     "n_gpus": b["n_gpus"], "step_ms_hip_events_in_traced_run": b["roofline"]["step_ms_hip_events"]})
        if "cholsol" in b:
            meta["nrhs_per_gpu"] = b["cholsol"]["nrhs_per_gpu"]
    except Exception:  # not a bench.py run (a tools/time_*.py script, a bench_configs.py section): say what was run instead
        try:
            with open(os.path.join(out, "trace.stdout")) as f:
                lines = [l for l in f.read().splitlines() if l.strip()]
            meta["run"] = "not a bench.py headline run; last line of the traced command's output follows"
            meta["last_output_line"] = lines[-1][:400] if lines else ""
        except OSError:
            meta["run"] = "output of the traced command not kept"
    with open(os.path.join(out, tag + "_traffic.json"), "w") as fjs:
        json.dump({"_meta": meta, "kernels": traffic}, fjs, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
