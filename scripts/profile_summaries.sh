#!/bin/bash
# After scripts/profile_all.sh's outputs were merged into gpurun_out/prof: write every profiles/r0<ROUND>_<tag>_pmc.json
# (+ kernel stats, clock, SQ counters).  The bench arguments per tag are the ones profile_all.sh used.
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof
declare -A ARGS=(
  [c3]=""
  [c3_hbm]=" --no-lds-keep --tile-drones 65536"
  [c3_tiles]=" --no-lds-keep"
  [c3_k1]=" --steps-per-launch 1 --waves-per-block 1 --tile-drones 65536"
  [shard8]=" --drones 8192"
  [c4]=" --drones 32768 --timesteps 4000"
  [c2]=" --drones 1024 --hidden 16,16 --coins 1"
  [c5]=" --drones 131072 --timesteps 400 --hidden 64,64,64 --coins 5 --walls 4"
  [gen12]=" --hidden 12,12 --no-specialise"
  [jit12]=" --hidden 12,12 --specialise"
)
ROUND=${ROUND:-3}
for tag in c3 c3_hbm c3_tiles c3_k1 shard8 c4 c2 c5 gen12 jit12; do
  kernel=fly_          # fly_pair_kernel / fly_split_kernel / fly_step_kernel: the one with the most time in the run
  python3 scripts/profile_summary.py --out profiles/r0${ROUND}_${tag}_pmc.json --round $ROUND --kernel $kernel \
    --trace $P/${tag}_trace --fetch $P/${tag}_fetch --write $P/${tag}_write --tcc $P/${tag}_tcc --sq $P/${tag}_sq --lds $P/${tag}_lds --bench $P/${tag}_trace.json \
    --cmd "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras${ARGS[$tag]}" \
    --pmc-cmd "rocprofv3 --pmc FETCH_SIZE (and, separately: WRITE_SIZE; TCC_HIT_sum TCC_MISS_sum; the SQ counters) --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-extras${ARGS[$tag]}"
done
python3 scripts/profile_extras.py
