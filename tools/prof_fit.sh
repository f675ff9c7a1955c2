#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_fit.sh <tag> <fit-concurrency>   -> kernel stats of a bench run
tag=$1; R=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fit-concurrency $R > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python tools/kstats.py $tag 16
