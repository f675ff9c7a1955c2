#!/bin/bash
# usage (GPU box, repo root): bash tools/tune_sweep.sh   - the fit of the bench line under a few tuning switches (A/B material)
run() { env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['fit_ms'].items()}, {k: round(x,2) for k,x in d['sub_ms'].items()})"; }
run A=0
run BOBE_XCD_SHARES=0
run BOBE_FILL_INV=0
run BOBE_FILL_INV=2 BOBE_FILL_INV_CHUNK=8
run BOBE_FILL_INV_CHUNK=2
run BOBE_TRTRI64=200
run A=0
