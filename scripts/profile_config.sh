#!/bin/bash
# Profile one bench configuration on the GPU box: kernel trace + FETCH_SIZE + WRITE_SIZE passes (three separate
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
# keep what gets merged back small: only the summaries and the counter files
find $out/${tag}_trace -name '*kernel_trace.csv' -delete
python3 bench.py --steps 3 --warmup 1 --no-extras "$@" > $out/${tag}_unprofiled.json 2>/dev/null
echo "profiled $tag"
