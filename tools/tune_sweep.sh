#!/bin/bash
# usage (GPU box, repo root): bash tools/tune_sweep.sh   - A/B material: tile-size switches of the inverse and K^-1 launches at N = 8192
run() { env "$@" python bench.py --config large --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],1), {k: round(x,1) for k,x in d['fit_ms'].items()}, {k: round(x,1) for k,x in d['sub_ms'].items()})"; }
run A=0
run BOBE_LAUUM64=3000
run BOBE_TRTRI64=5000
run BOBE_LAUUM64=3000 BOBE_TRTRI64=5000
run BOBE_LAUUM64=3000 BOBE_TRTRI64=1100
run A=0
