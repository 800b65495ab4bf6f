"""Reduce the counter passes of tools/profile_issue.sh: per kernel (launches of at least 0.1 ms worth of work: the big ones), the mean of
every counter over its launches.  usage: profile_issue_reduce.py <dir> <tag>"""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name") or row.get("Kernel Name")
        c = row.get("Counter_Name") or row.get("Counter Name")
        v = float(row.get("Counter_Value") or row.get("Counter Value") or 0.0)
        d = acc.setdefault(k, {}).setdefault(c, [0.0, 0])
        d[0] += v
        d[1] += 1
res = {}
for k, cs in acc.items():
    if not k or "rocclr" in k:
        continue
    res[k[:110]] = {c: {"mean_per_launch": s / n, "launches": n} for c, (s, n) in cs.items()}
json.dump({"_meta": {"tag": tag, "note": "rocprofv3 --pmc, two passes; values are sums over the device per launch unless the counter is a derived percentage"},
           "kernels": res}, open(os.path.join(out, tag + "_issue.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: round(v["mean_per_launch"], 3) for c, v in cs.items()} for k, cs in res.items()
                  if any(x in k for x in ("cholsol", "chol_clique", "chol_block", "rag_mfma", "gaxpy_tiled"))}, indent=1))
