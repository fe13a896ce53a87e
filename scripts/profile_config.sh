#!/bin/bash
# Profile one bench configuration on the GPU box: kernel trace + FETCH_SIZE + WRITE_SIZE + TCC hit/miss + SQ passes (separate
# rocprofv3 runs: --pmc is never combined with other trace domains), outputs under gpurun_out/prof/<tag>_*.
#   scripts/profile_config.sh <tag> <bench args...>
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --steps 3 --warmup 1 --no-extras "$@" > $out/${tag}_trace.json 2> $out/${tag}_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 bench.py --steps 1 --warmup 0 --no-extras "$@" > $out/${tag}_fetch.json 2> $out/${tag}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 bench.py --steps 1 --warmup 0 --no-extras "$@" > $out/${tag}_write.json 2> $out/${tag}_write.err
# L2 hits and misses (what share of the launch's requests the L2s serve), and the SQ view (vector instructions issued,
# wave-cycles busy / issuing / waiting) -- two more passes of their own
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/${tag}_tcc -- python3 bench.py --steps 1 --warmup 0 --no-extras "$@" > $out/${tag}_tcc.json 2> $out/${tag}_tcc.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $out/${tag}_sq -- python3 bench.py --steps 1 --warmup 0 --no-extras "$@" > $out/${tag}_sq.json 2> $out/${tag}_sq.err
# the LDS array: cycles it was busy and the extra cycles bank conflicts cost (a pass of its own; not fatal where a counter is missing)
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $out/${tag}_lds -- python3 bench.py --steps 1 --warmup 0 --no-extras "$@" > $out/${tag}_lds.json 2> $out/${tag}_lds.err || echo "no LDS counter pass for $tag"
# keep what gets merged back small: only the summaries and the counter files
find $out/${tag}_trace -name '*kernel_trace.csv' -delete
python3 bench.py --steps 3 --warmup 1 --no-extras "$@" > $out/${tag}_unprofiled.json 2>/dev/null
echo "profiled $tag"
