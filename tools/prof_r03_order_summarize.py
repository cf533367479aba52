"""Summarises tools/prof_r03_order.sh: per view and configuration the L2 / L1 counters and the average duration of the two timed k_trace2
instantiations (primary, bounce) -> <out>/summary/r03_layout_order_pmc.json."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
summ = os.path.join(out, "summary"); os.makedirs(summ, exist_ok=True)


def find(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


def which(kernel_name):
    if "k_trace2<" not in kernel_name:
        return None
    args = kernel_name.split("k_trace2<")[1].split(">")[0].replace(" ", "").split(",")
    if args[1] != "false":
        return None                      # the counting instantiation (untimed counter pass of bench.py)
    return "primary" if args[0] == "true" else "bounce"


res = {}
for view in ("headline", "interior"):
    for cfg in ("ref_queue", "couples_queue", "ref_order", "couples_order"):
        e = {}
        f = find(f"{view}_{cfg}_l2/**/*counter_collection.csv")
        if f:
            acc = {}
            for r in csv.DictReader(open(f)):
                k = which(r["Kernel_Name"])
                if k:
                    acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, c in acc.items():
                last = {n: sum(v[-4:]) / len(v[-4:]) for n, v in c.items()}          # the timed launches (2 repetitions x 2 batches)
                h, m = last.get("TCC_HIT_sum"), last.get("TCC_MISS_sum")
                rq, ac = last.get("TCP_TCC_READ_REQ_sum"), last.get("TCP_TOTAL_CACHE_ACCESSES_sum")
                e[k] = {"TCC_HIT": h, "TCC_MISS": m, "l2_hit_rate": h / (h + m) if h is not None and m else None, "TCP_TCC_READ_REQ": rq, "TCP_TOTAL_CACHE_ACCESSES": ac,
                        "l1_hit_rate": 1.0 - rq / ac if rq is not None and ac else None}
        f = find(f"{view}_{cfg}_stats/**/*kernel_stats.csv")
        if f:
            for r in csv.DictReader(open(f)):
                k = which(r["Name"])
                if k:
                    e.setdefault(k, {})["avg_us_all_launches"] = float(r["AverageNs"]) / 1e3; e[k]["calls"] = int(r["Calls"])
            tot = {}
            for r in csv.DictReader(open(f)):
                n = r["Name"]
                if any(t in n for t in ("k_sort_", "k_order_")):
                    tot["order_and_sort_kernels_total_ms"] = tot.get("order_and_sort_kernels_total_ms", 0.0) + float(r["TotalDurationNs"]) / 1e6
            e.update(tot)
        lg = os.path.join(out, f"{view}_{cfg}_stats.log")
        if os.path.exists(lg):
            for line in open(lg):
                if line.startswith("{") and '"metric"' in line:
                    j = json.loads(line); e["mray_s_under_profiler"] = j["value"]; e["ms_per_step_under_profiler"] = j["ms_per_step"]
        res[f"{view}/{cfg}"] = e
json.dump(res, open(os.path.join(summ, "r03_layout_order_pmc.json"), "w"), indent=1)
for k, v in res.items():
    print(k, json.dumps({a: (b if not isinstance(b, dict) else {x: (round(y, 4) if isinstance(y, float) else y) for x, y in b.items() if x in ("l2_hit_rate", "l1_hit_rate", "avg_us_all_launches")}) for a, b in v.items()}))
