"""Aggregate rocprofv3 --pmc passes (run with --output-format csv) into the JSON committed under profiles/.

Usage: [PMC_STEPS=5] python tools/pmc_summary.py <out.json> <kernel-substring>[,<kernel-substring>...] <dir-or-csv> [<dir-or-csv> ...]
PMC_STEPS = steps (timed + warm-up) of the profiled bench command: written as "_meta" so that bench.py can turn "per launch" into "per step".
Every *_counter_collection.csv below the given paths is read; per kernel (matched by substring, reported under
the substring) and per counter: the mean over dispatches of the counter value (FETCH_SIZE / WRITE_SIZE are KB).
"""
import csv
import glob
import json
import os
import sys


def main():
    out_path, keys, paths = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    files = []
    for p in paths:
        if os.path.isdir(p):
            files += glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
        else:
            files.append(p)
    acc = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("kernel_name") or ""
                ctr = row.get("Counter_Name") or row.get("counter_name")
                val = float(row.get("Counter_Value") or row.get("counter_value") or 0.0)
                disp = row.get("Dispatch_Id") or row.get("dispatch_id")
                for k in keys:
                    if k in name:
                        acc.setdefault(k, {}).setdefault(ctr, {}).setdefault((f, disp), 0.0)
                        acc[k][ctr][(f, disp)] += val  # one row per (dispatch, counter[, dimension])
    res = {}
    for k, ctrs in acc.items():
        res[k] = {}
        for c, d in ctrs.items():
            vals = list(d.values())
            res[k][c] = {"per_launch_KB_mean" if c.endswith("_SIZE") else "per_launch_mean": sum(vals) / len(vals),
                         "launches": len(vals)}
    if os.environ.get("PMC_STEPS"):
        res["_meta"] = {"steps": int(os.environ["PMC_STEPS"])}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
