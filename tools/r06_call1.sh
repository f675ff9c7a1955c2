#!/bin/bash
# round 6, GPU call 1: block-solve parity + ladder + A/B timing + config-5 kernel stats
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_refinement.py -x -q > gpurun_out/r06/refinement_tests.log 2>&1 || { tail -30 gpurun_out/r06/refinement_tests.log; exit 1; }
tail -2 gpurun_out/r06/refinement_tests.log
python -m pytest tests/test_gpu_conditioning.py -q > gpurun_out/r06/ladder_tests.log 2>&1; tail -3 gpurun_out/r06/ladder_tests.log
timeout -k 10 600 python tools/solve_block_ab.py headline > gpurun_out/r06/ab.log 2>&1 || { tail -20 gpurun_out/r06/ab.log; exit 1; }
tail -25 gpurun_out/r06/ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/prof_c5 -- python3 tools/config5_run.py seed=7 > gpurun_out/r06/config5_rocprof.txt 2>&1 || exit 1
cp "$(ls gpurun_out/r06/prof_c5/*/*kernel_stats.csv | head -1)" gpurun_out/r06/config5_kernel_stats.csv
rm -rf gpurun_out/r06/prof_c5
tail -2 gpurun_out/r06/config5_rocprof.txt | cut -c1-600
python3 tools/config5_run.py seed=7 2>&1 | tail -1 | cut -c1-600
