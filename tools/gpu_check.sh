#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_check.sh <tag> [bench args...]
# runs the GPU parity suite, then rocprofv3 kernel stats around the headline bench
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/pytest_gpu_$tag.log 2>&1
tail -4 gpurun_out/pytest_gpu_$tag.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
tail -c 900 gpurun_out/bench_$tag.json
