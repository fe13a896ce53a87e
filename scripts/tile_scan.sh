#!/bin/bash
# Config 3 by tile size and wavefronts per block: drone-steps/s of the step engine (bench.py --no-extras).
cd "$(dirname "$0")/.."
out=gpurun_out/tile_scan.log; : > $out
for tile in 4096 6144 8192 10240 12288 16384; do
  for waves in 8 16 32; do
    python3 bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --tile-drones $tile --waves-per-block $waves 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tile', $tile, 'waves', $waves, 'value %.4g' % d['value'], 'ms %.2f' % d['ms_per_step'], 'launch_us %.1f' % d['roofline']['launch_us'], flush=True)" >> $out || echo "tile $tile waves $waves failed" >> $out
  done
done
cat $out
