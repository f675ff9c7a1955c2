#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_refinement.py -x -q > gpurun_out/r06/refinement_tests.log 2>&1 || { tail -30 gpurun_out/r06/refinement_tests.log; exit 1; }
tail -2 gpurun_out/r06/refinement_tests.log
python -m pytest tests/test_gpu_conditioning.py -q > gpurun_out/r06/ladder_tests.log 2>&1; tail -3 gpurun_out/r06/ladder_tests.log
timeout -k 10 600 python tools/solve_block_ab.py headline > gpurun_out/r06/ab.log 2>&1 || { tail -20 gpurun_out/r06/ab.log; exit 1; }
timeout -k 10 600 python tools/solve_block_ab.py small > gpurun_out/r06/ab_small.log 2>&1 || { tail -20 gpurun_out/r06/ab_small.log; exit 1; }
cat gpurun_out/r06_solve_block_ab_headline.txt gpurun_out/r06_solve_block_ab_small.txt | cut -c1-120
