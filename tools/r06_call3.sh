#!/bin/bash
# round 6, GPU call 3: the whole GPU suite at HEAD, config 5 (default / mult 64 / block env), bench headline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -x -q > gpurun_out/r06/gputest_a.log 2>&1; rc=$?
tail -15 gpurun_out/r06/gputest_a.log
[ $rc -ne 0 ] && exit $rc
for s in 7 1; do python3 tools/config5_run.py seed=$s 2>&1 | tail -1 | cut -c1-700; done > gpurun_out/r06/config5_default.txt
cat gpurun_out/r06/config5_default.txt
for s in 7 1; do BOBE_HMC_MULT=64 python3 tools/config5_run.py seed=$s 2>&1 | tail -1 | cut -c1-700; done > gpurun_out/r06/config5_mult64.txt
cat gpurun_out/r06/config5_mult64.txt
python3 bench.py --steps 10 --warmup 2 > gpurun_out/r06/bench_headline_a.json 2> gpurun_out/r06/bench_headline_a.err || { tail -20 gpurun_out/r06/bench_headline_a.err; exit 1; }
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r06/bench_headline_a.json'))
for k in ('value','ms_per_step','sub_ms','fit_ms','accurate_sweep','reference_noise_cycle','matern_cycle','roofline'):
    print(k, json.dumps(j.get(k))[:900])
print('cpu', json.dumps(j.get('cpu_baseline'))[:600])
PY
