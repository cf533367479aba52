"""Summarises the rocprofv3 passes of tools/prof_r02.sh into gpurun_out/<tag>/summary/{r02_bench_kernel_stats.csv, r02_bench_pmc_summary.json, r02_traffic.json}."""
import csv
import glob
import json
import os
import shutil
import sys

out = sys.argv[1]
summ = os.path.join(out, "summary"); os.makedirs(summ, exist_ok=True)


def find(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


st = find("stats/**/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(summ, "r02_bench_kernel_stats.csv"))


def counters(name):
    """{kernel short name: {counter: [value per dispatch]}} for the two non-counting k_trace2 instantiations."""
    f = find(name + "/**/*counter_collection.csv")
    res = {}
    if not f:
        return res
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trace2<" not in k:
            continue
        args = k.split("k_trace2<")[1].split(">")[0].replace(" ", "").split(",")
        if args[1] != "false":          # counting build = the untimed counter pass
            continue
        key = "primary" if args[0] == "true" else "bounce"
        res.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return res


def mean_of_batches(vals, samples):
    """The timed launches: repeats x batches, all with the same sample count; warm-up launches carry 32 samples as well (warmup 32), so
    for --steps 20 the 20-sample launches are the LAST `repeats` ones."""
    if not vals:
        return None
    v = vals[-2:] if samples == 20 else vals[-4:]
    return sum(v) / len(v)


summary, traffic = {}, {"n1": {"headline": {}, "interior": {}}}
for name, view, samples in (("head_s32", "headline", 32), ("head_s20", "headline", 20), ("int_s32", "interior", 32)):
    fe, wr, l2 = counters(name + "_fetch"), counters(name + "_write"), counters(name + "_l2")
    entry = {}
    tot_bytes, hits, miss = 0.0, 0.0, 0.0
    for k in ("primary", "bounce"):
        f = mean_of_batches(fe.get(k, {}).get("FETCH_SIZE"), samples); w = mean_of_batches(wr.get(k, {}).get("WRITE_SIZE"), samples)
        h = mean_of_batches(l2.get(k, {}).get("TCC_HIT_sum"), samples); m = mean_of_batches(l2.get(k, {}).get("TCC_MISS_sum"), samples)
        rq = mean_of_batches(l2.get(k, {}).get("TCP_TCC_READ_REQ_sum"), samples); ac = mean_of_batches(l2.get(k, {}).get("TCP_TOTAL_CACHE_ACCESSES_sum"), samples)
        entry[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "TCC_HIT": h, "TCC_MISS": m, "TCP_TCC_READ_REQ": rq, "TCP_TOTAL_CACHE_ACCESSES": ac,
                    "l2_hit_rate": (h / (h + m)) if h is not None and m is not None and h + m > 0 else None,
                    "l1_hit_rate": (1.0 - rq / ac) if rq is not None and ac else None,
                    "hbm_side_bytes_per_launch": ((f or 0) + (w or 0)) * 1024.0}
        tot_bytes += entry[k]["hbm_side_bytes_per_launch"]; hits += h or 0.0; miss += m or 0.0
    summary[name] = entry
    l1m = sum((entry[k]["TCP_TCC_READ_REQ"] or 0.0) for k in ("primary", "bounce"))
    traffic["n1"][view][f"s{samples}"] = {"traversal_hbm_bytes_per_launch": int(tot_bytes / 2), "l2_hit_rate": round(hits / (hits + miss), 4) if hits + miss > 0 else None,
                                          "l1_miss_requests_per_launch": int(l1m / 2), "l2_miss_requests_per_launch": int(miss / 2),
                                          "note": "mean over the two k_trace2 launches (primary + bounce) of a batch: (FETCH_SIZE + WRITE_SIZE) KiB x 1024, separate --pmc passes; FETCH_SIZE calibrated at 1.03 on this access pattern in round 1 (profiles/r01_bench_pmc_summary.json)"}
sq = counters("head_s32_sq")
for k, d in sq.items():
    e = {c: mean_of_batches(v, 32) for c, v in d.items()}
    if e.get("SQ_ACTIVE_INST_VALU") and e.get("SQ_THREAD_CYCLES_VALU"):
        e["valu_lane_utilisation"] = e["SQ_THREAD_CYCLES_VALU"] / (64.0 * e["SQ_ACTIVE_INST_VALU"])
    summary.setdefault("head_s32_sq", {})[k] = e
json.dump({"command": "python bench.py --steps {64|20} [--view interior] --warmup 32 --repeats 2 --no-extras --no-cpu-baseline under rocprofv3 --kernel-trace --pmc <set> (tools/prof_r02.sh)",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB; per-launch means over the timed launches of each non-counting k_trace2 instantiation", "passes": summary},
          open(os.path.join(summ, "r02_bench_pmc_summary.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(summ, "r02_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
