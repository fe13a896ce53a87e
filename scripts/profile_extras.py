#!/usr/bin/env python3
"""Summaries of the non-PMC-traffic passes of scripts/profile_all.sh: effective shader clock of the headline
kernel (GRBM_GUI_ACTIVE), SQ counters of the default bench's kernels, kernel stats of the default bench and of
the traced configurations.  Run from the repo root after profile_all.sh's outputs were merged into gpurun_out/prof."""
import collections, csv, glob, json, os, statistics

ROUND = int(os.environ.get("ROUND", "3"))
R = f"r{ROUND:02d}"

P = "gpurun_out/prof"

def load(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows

def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])

# ---- clock of the profiled headline kernel
rows = [r for r in load(f"{P}/c3_clock/**/*counter_collection.csv") if any(k in r["Kernel_Name"] for k in ("fly_split_kernel", "fly_pair_kernel", "fly_wide_kernel")) and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
clocks = [float(r["Counter_Value"]) / 8 / d for r, d in zip(rows, durs)]
un, tr = last_json(f"{P}/c3_unprofiled.json"), json.load(open(f"profiles/{R}_c3_pmc.json"))
json.dump({"round": ROUND, "what": "effective shader clock of the headline kernel while rocprofv3 collects counters: GRBM_GUI_ACTIVE / 8 / kernel duration (MI355X_MICROARCH.md, DVFS give-back)",
           "command": "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-extras",
           "kernel": tr["kernel"][:80], "dispatches": len(rows),
           "effective_clock_ghz": {"mean": statistics.fmean(clocks), "min": min(clocks), "max": max(clocks)},
           "launch_us_in_this_pass": statistics.fmean(durs) / 1e3, "launch_us_kernel_trace_pass": tr["launch_us_profiled"],
           "launch_us_unprofiled_same_box": un["roofline"]["launch_us"],
           "note": "profiled and un-profiled launch durations of one box agree within its run-to-run noise (a few per cent); boxes of the "
                   "pool differ by a few per cent among themselves"},
          open(f"profiles/{R}_c3_clock.json", "w"), indent=1)

# ---- SQ counters of the default bench
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in load(f"{P}/full_sq/**/*counter_collection.csv"):
    name = r["Kernel_Name"]
    key = next((k for k in ("fly_wide_kernel", "fly_pair_kernel", "fly_split_kernel", "fly_resident_kernel", "ga_breed_kernel", "ga_select_kernel", "finalize_kernel", "reset_kernel") if k in name), None)
    if key is None:
        continue
    if key == "fly_wide_kernel":
        key += " regs=" + name.split("fly_wide_kernel<")[1].split(",")[0].strip()
    if key == "fly_pair_kernel":
        args = name.split("fly_pair_kernel<")[1].split(",")
        key += f" keep={args[0].strip()} regs={args[1].strip()}"
    if key == "fly_split_kernel":
        args = name.split("fly_split_kernel<float, false, ")[1].split(",")
        key += f" L={args[0].strip()} S={args[1].strip()} keep={args[2].strip()}"
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for key, cs in agg.items():
    n, top = len(cs["SQ_WAVES"]), max(cs["SQ_WAVE_CYCLES"])
    sel = [i for i in range(n) if cs["SQ_WAVE_CYCLES"][i] > 0.8 * top]
    d = {c: statistics.fmean([v[i] for i in sel]) for c, v in cs.items()}
    wc = d["SQ_WAVE_CYCLES"]
    d.update(dispatches_averaged=len(sel), dispatches_total=n, frac_active_inst=d["SQ_ACTIVE_INST_ANY"] / wc, frac_wait_any=d["SQ_WAIT_ANY"] / wc,
             frac_wait_inst=d["SQ_WAIT_INST_ANY"] / wc, valu_per_wave=d["SQ_INSTS_VALU"] / d["SQ_WAVES"], lds_per_wave=d["SQ_INSTS_LDS"] / d["SQ_WAVES"],
             salu_per_wave=d["SQ_INSTS_SALU"] / d["SQ_WAVES"])
    out[key] = d
json.dump({"round": ROUND, "command": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
           "note": "per-dispatch means over the dispatches within 20 % of the kernel's largest SQ_WAVE_CYCLES (full live-population launches: the bench also flies random-genome populations whose launches are mostly frozen); SQ_* cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md); fly_split L = wavefronts per workgroup, S = workgroups per 64-drone block (3: workgroups of 48 drones), keep = KiB of genes per workgroup kept in LDS; fly_resident: a whole 2 000-step episode of 65 536 drones per launch",
           "kernels": out}, open(f"profiles/{R}_sq_counters.json", "w"), indent=1)

# ---- memory-side traffic of the generation step's breed kernel
breed = {}
for name, d in (("FETCH_SIZE", "full_fetch"), ("WRITE_SIZE", "full_write")):
    vals = [float(r["Counter_Value"]) for r in load(f"{P}/{d}/**/*counter_collection.csv") if "ga_breed_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    if vals:
        breed[name + "_mean_kb"] = statistics.fmean(vals)
        breed["dispatches"] = len(vals)
if breed:
    bench = last_json(f"{P}/full_trace.json")
    P4 = 4 * sum(a * b for a, b in zip([32, 32, 4], [16, 33, 33]))            # config 3: 4P bytes per drone
    n = bench["config"]["population"] // bench["n_gpus"]
    breed.update(algorithmic_bytes=2 * P4 * n, read_bytes=2 * breed.get("FETCH_SIZE_mean_kb", 0) * 1024, write_bytes=breed.get("WRITE_SIZE_mean_kb", 0) * 1024,
                 select_breed_ms=bench["generation"]["select_breed_ms"],
                 note="ga_breed_kernel at config 3: every lane reads one 16-byte word of ITS child's parent, 64 different parents per wavefront: "
                      "64 sectors of 128 bytes for 1 KiB of genes; read_bytes = 2 x FETCH_SIZE (gfx950 correction) is what the L2s fetched for it")
    breed["read_over_algorithmic_read"] = breed["read_bytes"] / (P4 * n)
    json.dump({"round": ROUND, "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
               "ga_breed_kernel": breed}, open(f"profiles/{R}_ga_breed_pmc.json", "w"), indent=1)
    print("ga_breed_kernel", breed)

# ---- kernel stats
def stats(src_glob, dst, top):
    rows = load(src_glob)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(dst, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows[:top]:
            r = dict(r); r["Name"] = r["Name"][:140]; w.writerow(r)
stats(f"{P}/full_trace/**/*kernel_stats.csv", f"profiles/{R}_bench_kernel_stats.csv", 20)
for t in ("c3", "c3_hbm", "c3_k1", "shard8", "c4", "c5"):
    stats(f"{P}/{t}_trace/**/*kernel_stats.csv", f"profiles/{R}_{t}_kernel_stats.csv", 8)
print(json.load(open(f"profiles/{R}_c3_clock.json"))["effective_clock_ghz"])
for k, d in out.items():
    print(k, {c: round(v, 3) for c, v in d.items() if c.startswith("frac") or c.endswith("per_wave")})
