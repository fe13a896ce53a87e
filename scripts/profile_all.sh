#!/bin/bash
# Every profile of the round on one GPU box (rocprofv3 kernel traces, FETCH_SIZE / WRITE_SIZE passes, GRBM clock pass,
# SQ counters), outputs under gpurun_out/prof/.  Summaries are written by scripts/profile_summary.py afterwards.
set -e
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/prof
rm -rf $out          # (rocprofv3 adds a directory per run: old runs would be averaged in)
mkdir -p $out
export TMPDIR=/tmp
# page the image in and bring the GPU to its sustained clocks before anything is recorded (the first run on a fresh box is ~5 % slow)
python3 bench.py --steps 10 --warmup 2 --no-extras > /dev/null 2>&1
scripts/profile_config.sh c3                                                       # BASELINE config 3 (headline: whole-population launches, 136 KiB of genes per workgroup kept in LDS)
scripts/profile_config.sh c3_hbm --no-lds-keep --tile-drones 65536                 # the same with nothing kept on chip: every timestep's genes from HBM
scripts/profile_config.sh c3_tiles --no-lds-keep                                   # the round-2 rollout: Infinity-Cache tiles, nothing in LDS
scripts/profile_config.sh c3_k1 --steps-per-launch 1 --waves-per-block 1 --tile-drones 65536   # the literal contract: one launch per timestep
scripts/profile_config.sh shard8 --drones 8192                                     # 65 536-drone population over 8 GPUs: one shard
scripts/profile_config.sh c4 --drones 32768 --timesteps 4000                       # config 4 shard
scripts/profile_config.sh c2 --drones 1024 --hidden 16,16 --coins 1                # config 2
scripts/profile_config.sh c5 --drones 131072 --timesteps 400 --hidden 64,64,64 --coins 5 --walls 4   # config 5 shard (short episode)
scripts/profile_config.sh gen12 --hidden 12,12 --no-specialise                                  # run-time-spec kernel
scripts/profile_config.sh jit12 --hidden 12,12 --specialise                        # the same spec specialised at run time
# effective shader clock of the profiled headline run: GRBM_GUI_ACTIVE / 8 / kernel duration
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/c3_clock -- python3 bench.py --steps 1 --warmup 0 --no-extras > $out/c3_clock.json 2> $out/c3_clock.err
# the whole default bench (resident engine, GA kernels, random-init population): trace, then SQ counters
rocprofv3 --kernel-trace --stats --output-format csv -d $out/full_trace -- python3 bench.py --no-cpu-baseline > $out/full_trace.json 2> $out/full_trace.err
find $out/full_trace -name '*kernel_trace.csv' -delete
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/full_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/full_sq.json 2> $out/full_sq.err
# memory-side traffic of the generation step's kernels (ga_breed_kernel gathers 16-byte parent words: how many bytes do the L2s fetch for them?)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/full_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/full_fetch.json 2> $out/full_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/full_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/full_write.json 2> $out/full_write.err
echo "all profiled"; du -sh $out
