#!/bin/bash
# Config 5 shard (131 072 x 64-64-64) by wavefronts per block.
cd "$(dirname "$0")/.."
out=gpurun_out/c5_scan.log; : > $out
for waves in 1 2 4 8; do
  python3 bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 --drones 131072 --timesteps 400 --hidden 64,64,64 --coins 5 --waves-per-block $waves 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('waves', $waves, 'value %.4g' % d['value'], 'ms %.2f' % d['ms_per_step'], 'launch_us %.1f' % d['roofline']['launch_us'], 'GB/s %.0f' % d['roofline']['achieved'], flush=True)" >> $out || echo "waves $waves failed" >> $out
done
cat $out
