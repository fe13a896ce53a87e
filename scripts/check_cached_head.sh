#!/bin/bash
# Default rollout (tile head kept in L2 where the library chooses to) vs the same tiles flown plainly, by layer spec.
cd "$(dirname "$0")/.."
run() {  # hidden drones
  python3 - "$1" "$2" <<'PY'
import sys, json, subprocess
hidden, n = sys.argv[1], sys.argv[2]
def bench(extra):
    out = subprocess.run([sys.executable, "bench.py", "--no-extras", "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--hidden", hidden, "--drones", n, "--timesteps", "1000"] + extra, capture_output=True, text=True).stdout
    return json.loads(out.strip().splitlines()[-1])
d = bench([])
r = d["roofline"]
p = bench(["--tile-drones", str(r["drones_per_launch"])])
print(hidden, n, "tile", r["drones_per_launch"], "cached", r["cached_drones_per_launch"], "default %.4g" % d["value"], "plain %.4g" % p["value"], "ratio %.3f" % (d["value"] / p["value"]), flush=True)
PY
}
run 32,32 65536
run 16,16 131072
run 24,24 65536
run 64 131072
run 32 131072
run 16 262144
run 8,8 262144
