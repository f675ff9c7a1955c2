"""The factorisation has several launch sequences (one-launch panel - with three or four row strips per workgroup - or
k_potf2 + k_trsm_panel for batches too wide for one launch; K = 256 update pairs or single panels; XCD tile shares or the
row-major tile order; 32 / 64 / 128-wide tiles; deferred updates in the panel launches; alone, on a slot, in a lock-step
batch, as a graph replay).  They must all produce the same bits: the library reads its tuning environment once per
process, so each variant runs tools/bits_snapshot.py in its own process and the digests of L, the MLL value, its
gradient, the batched values, the predictions and a three-chunk sweep are compared."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digests(extra_env, timeout=900):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bits_snapshot.py"), "print"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])


def _same(base, other, what):
    assert base.keys() == other.keys(), what
    differing = [k for k in base if base[k] != other[k]]
    assert not differing, (what, differing)


def test_every_launch_sequence_gives_the_same_bits():
    base = _digests({"BITS_MAX_N": "1500"})
    assert len(base) >= 30
    for variant in ({"BOBE_XCD_SHARES": "0"}, {"BOBE_PAIR_MIN": "0"}, {"BOBE_SYRK32_BELOW": "0"}, {"BOBE_TRTRI64": "0"},
                    {"BOBE_LOCKSTEP_MIN_N": "1024"}, {"BOBE_LOCKSTEP_MIN_N": "1024", "BOBE_GRAPH_MAX_N": "0"},
                    {"BOBE_LOCKSTEP_MIN_N": "1024", "BOBE_MLL_SLOTS": "2"}, {"BOBE_LOCKSTEP_MIN_N": "200"}):
        _same(base, _digests(dict(variant, BITS_MAX_N="1500")), variant)


def test_the_switches_that_only_act_on_large_matrices():
    """N = 4096 and 8192: XCD tile shares need >= 256 tiles in a launch, K = 256 pairs need B rem^2 > 300, the 128-wide
    tiles of the inverse need >= 600 blocks in a level, the update fillers 20+ block columns; an eight-wide lock-step
    batch takes the k_potf2 + k_trsm_panel pair for its first panels (8 x 62 panel workgroups do not fit 256 CUs)."""
    sizes = {"BITS_SIZES": "4096:8:rbf,8192:8:matern", "BITS_B8": "1"}
    base = _digests(sizes)
    assert len(base) == 2 * 7 + 2 * 2                 # (no sweeps above N = 2048; + the eight-wide batches)
    for variant in ({"BOBE_XCD_SHARES": "0"}, {"BOBE_PAIR_MIN": "0"}, {"BOBE_TRTRI64": "0"}, {"BOBE_TRTRI64": "100000"},
                    {"BOBE_FILL": "0"}, {"BOBE_FILL": "2"}, {"BOBE_SYRK32_BELOW": "0"}):
        _same(base, _digests(dict(variant, **sizes)), variant)


def test_deferred_updates_in_the_panel_shadows_do_not_move_a_bit():
    """potrf() lets trailing-update tiles of far block columns ride in the panel launches as filler workgroups
    (k_chol_panel<true, .>, chol_plan in gp_factor.hip).  Where a tile is computed must not change what it holds: the
    factor, MLL, gradient, batch results and predictions with the fillers forced on everywhere (BOBE_FILL=2: lone and
    lock-step, ragged and full sizes, both kernels) equal, bit for bit, those with every update in its own launch (=0)
    and those of the default rule (=1: fill_pays)."""
    sizes = "300:3:rbf,1500:8:matern,2500:5:rbf,3200:6:matern,4096:8:rbf,5000:4:rbf"
    digests = {mode: _digests({"BOBE_FILL": mode, "BITS_SIZES": sizes, "BOBE_LOCKSTEP_MIN_N": "1024"}) for mode in ("0", "2", "1")}
    assert len(digests["0"]) == 6 * 7 + 2      # (+ the sweeps at N <= 2048)
    _same(digests["0"], digests["2"], "forced on")
    _same(digests["0"], digests["1"], "default rule")
