#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one bench configuration into a profiles/*.json summary.

    python scripts/profile_summary.py --out profiles/r02_fly_step_pmc.json --round 2 --kernel fly_split_kernel \
        --trace gpurun_out/prof/c3_trace --fetch gpurun_out/prof/c3_fetch --write gpurun_out/prof/c3_write \
        --bench gpurun_out/prof/c3_trace.json --cmd "python3 bench.py --steps 3 --warmup 1 --no-extras"

--trace / --fetch / --write are the -d directories of three separate rocprofv3 runs of the SAME bench command
(--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE), --bench the JSON line the traced run printed.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on wide coalesced
reads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact."""
import os, argparse, csv, glob, json, statistics, sys

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True); ap.add_argument("--round", type=int, default=2)
ap.add_argument("--kernel", required=True, help="substring of the kernel name")
ap.add_argument("--trace"); ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--bench")
ap.add_argument("--tcc", help="-d directory of the --pmc TCC_HIT_sum TCC_MISS_sum pass"); ap.add_argument("--sq", help="-d directory of the SQ counter pass")
ap.add_argument("--lds", help="-d directory of the SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT pass")
ap.add_argument("--cmd", default=""); ap.add_argument("--pmc-cmd", default="")
a = ap.parse_args()

def rows(d, suffix):
    out = []
    for f in glob.glob(f"{d}/**/*{suffix}", recursive=True):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out

doc = {"round": a.round, "kernel_match": a.kernel, "command_trace": a.cmd, "command_pmc": a.pmc_cmd}
bench = None
if a.bench:
    lines = [l for l in open(a.bench).read().splitlines() if l.startswith("{")]
    bench = json.loads(lines[-1])
    r = bench["roofline"]
    doc["config"] = {"drones": bench["config"]["population"] // bench["n_gpus"], "hidden": bench["config"]["hidden"],
                     "coins": bench["config"]["coins"], "timesteps": bench["config"]["timesteps"],
                     "timesteps_per_launch": r["timesteps_per_launch"], "wavefronts_per_block": r["wavefronts_per_block"],
                     "drones_per_launch": r.get("drones_per_launch", bench["config"]["population"] // bench["n_gpus"]),
                     "kept_words_per_drone": r.get("kept_words_per_drone", 0), "streamed_words_per_drone": r.get("streamed_words_per_drone")}
    doc["bench_line_of_the_traced_run"] = {"launch_us_from_hip_events": r["launch_us"], "algorithmic_gbs": r.get("algorithmic_gbs"),
                                           "bound": r.get("bound"), "frac": r["frac"], "value": bench["value"], "ms_per_step": bench["ms_per_step"]}
    doc["algorithmic_bytes_per_launch"] = r["bytes_per_launch"]
if a.trace:
    ks = [r for r in rows(a.trace, "kernel_stats.csv") if a.kernel in r["Name"]]
    if not ks:
        sys.exit(f"no kernel matching {a.kernel} in {a.trace}")
    k = max(ks, key=lambda r: float(r["TotalDurationNs"]))
    doc["kernel"] = k["Name"][:200]
    doc["kernel_trace"] = {"calls": int(k["Calls"]), "average_ns": float(k["AverageNs"]), "min_ns": float(k["MinNs"]),
                           "max_ns": float(k["MaxNs"]), "percentage_of_gpu_time": float(k["Percentage"])}
    doc["launch_us_profiled"] = float(k["AverageNs"]) / 1e3
    if bench:
        doc["algorithmic_gbs_from_trace"] = doc["algorithmic_bytes_per_launch"] / float(k["AverageNs"])
    doc["all_kernels"] = [{"name": r["Name"][:120], "calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]),
                           "percentage": float(r["Percentage"])} for r in sorted(rows(a.trace, "kernel_stats.csv"),
                                                                                 key=lambda r: -float(r["TotalDurationNs"]))[:12]]
pmc = {}
for name, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
    if not d:
        continue
    per_kernel = {}
    for r in rows(d, "counter_collection.csv"):
        if a.kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
            per_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    vals = [v for vs in per_kernel.values() for v in vs]
    if vals:
        # the first launches of a run include cold effects; keep them all, report the spread
        pmc[name] = {"dispatches": len(vals), "mean_kb": statistics.fmean(vals), "min_kb": min(vals), "max_kb": max(vals)}
if pmc:
    doc["pmc"] = pmc
    doc["gfx950_correction"] = ("FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: read bytes = 2 x FETCH_SIZE x 1024 "
                                "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE x 1024 is exact")
    rd = 2 * pmc["FETCH_SIZE"]["mean_kb"] * 1024 if "FETCH_SIZE" in pmc else None
    wr = pmc["WRITE_SIZE"]["mean_kb"] * 1024 if "WRITE_SIZE" in pmc else None
    doc["hbm_read_bytes_per_launch"], doc["hbm_write_bytes_per_launch"] = rd, wr
    if rd is not None and wr is not None:
        doc["hbm_traffic_bytes_per_launch"] = rd + wr
        if bench:
            doc["traffic_over_algorithmic"] = (rd + wr) / doc["algorithmic_bytes_per_launch"]
# any other counters, mean per dispatch of the matching kernel
def counter_means(d):
    acc = {}
    for r in rows(d, "counter_collection.csv"):
        if a.kernel in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: statistics.fmean(v) for k, v in acc.items()}

per = {"fabric_bytes": doc.get("hbm_traffic_bytes_per_launch")}
if a.tcc:
    t = counter_means(a.tcc)
    if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t:
        doc["tcc"] = t
        per["l2_hit_rate"] = t["TCC_HIT_sum"] / max(t["TCC_HIT_sum"] + t["TCC_MISS_sum"], 1.0)
if a.sq:
    q = counter_means(a.sq)
    doc["sq"] = q
    if "SQ_INSTS_VALU" in q:
        per["valu_wave_instructions"] = q["SQ_INSTS_VALU"]
    if q.get("SQ_WAVE_CYCLES"):
        per["sq_issue_fraction"] = q.get("SQ_ACTIVE_INST_ANY", 0.0) / q["SQ_WAVE_CYCLES"]
        per["sq_wait_fraction"] = q.get("SQ_WAIT_ANY", 0.0) / q["SQ_WAVE_CYCLES"]
        per["sq_issue_stall_fraction"] = q.get("SQ_WAIT_INST_ANY", 0.0) / q["SQ_WAVE_CYCLES"]
if a.lds and os.path.isdir(a.lds):
    q = counter_means(a.lds)
    if "SQ_LDS_IDX_ACTIVE" in q:
        doc["lds_counters"] = q
        per["lds_array_cycles"] = q["SQ_LDS_IDX_ACTIVE"]              # summed over the chip's CUs
        per["lds_bank_conflict_cycles"] = q.get("SQ_LDS_BANK_CONFLICT")
        if doc.get("launch_us_profiled"):
            # share of the launch's CU-cycles (256 CUs at the nominal 2.4 GHz) in which a CU's LDS array was busy
            per["lds_busy_fraction"] = q["SQ_LDS_IDX_ACTIVE"] / (256 * 2.4e3 * doc["launch_us_profiled"])
doc["per_launch"] = per
if "hbm_traffic_bytes_per_launch" in doc:
    doc["counter_note"] = ("FETCH_SIZE / WRITE_SIZE sit on the L2s' memory side: they count what the L2s pass on to the fabric, "
                           "Infinity-Cache hits included (MI355X_MICROARCH.md), not HBM bytes alone")
json.dump(doc, open(a.out, "w"), indent=1)
print(json.dumps({k: doc.get(k) for k in ("launch_us_profiled", "achieved_gbs_from_trace", "traffic_over_algorithmic")}))
