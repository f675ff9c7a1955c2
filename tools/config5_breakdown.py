"""GPU-busy time per phase of a config-5 run from a rocprofv3 kernel-stats file, beside the run's own phase timers.

    python tools/config5_breakdown.py <kernel_stats.csv> <config5_run output> [<same run without the profiler>]
Kernels are attributed to the phase that launches them (bo.py's timers: 'GP Training', 'Acquisition Optimization',
'MCMC Sampling', 'Nested Sampling'); kernels both the fit and the acquisition launch (coordinate scaling, the alpha solves,
device copies) are split by the launch counts of kernels that belong to one phase only."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
run = open(sys.argv[2]).read()
plain = open(sys.argv[3]).read() if len(sys.argv) > 3 else ""
PHASE = [
    ("MCMC Sampling", ("k_hmc_run", "k_hmc_leapfrog")),
    ("Nested Sampling", ("k_rwalk",)),
    ("GP Training", ("k_chol_panel", "k_syrk_trail", "k_trtri", "k_trti_diag", "k_lauum", "k_mll_grad_reduce", "k_mll_terms",
                     "k_potf2", "k_trsm_panel", "k_kernel_matrix<0, true", "k_kernel_matrix<1, true", "k_fill_diag", "k_diag_")),
    ("Acquisition Optimization", ("k_trimul", "k_blk_step", "k_cross_vv", "k_wip_score", "k_wg_", "k_wip_grad", "k_argmin",
                                  "k_predict_finalize", "k_kernel_matrix<0, false", "k_kernel_matrix<1, false", "k_vec_axpy",
                                  "k_gram_small", "k_append_rows", "k_load_padded", "k_predict_grad", "k_ei")),
]
busy = {p: 0.0 for p, _ in PHASE}
busy["shared (scaling, alpha solves, copies, fills)"] = 0.0
top = {p: [] for p in busy}
for r in rows:
    name, t = r["Name"], float(r["TotalDurationNs"]) / 1e9
    for p, keys in PHASE:
        if any(k in name for k in keys):
            break
    else:
        p = "shared (scaling, alpha solves, copies, fills)"
    busy[p] += t
    top[p].append((t, int(r["Calls"]), name.split("(")[0][:60]))
total = sum(busy.values())


def timers(text):
    m = re.findall(r"timing (\{[^}]*\})", text)
    return eval(m[-1]) if m else {}


def wall(text):
    m = re.findall(r"after (\d+) GP points in ([0-9.]+)s", text)
    return (int(m[-1][0]), float(m[-1][1])) if m else (0, float("nan"))


tp, tq = timers(run), timers(plain)
print("# config 5 (tools/config5_run.py seed=7): where the run's seconds go - rocprofv3 --kernel-trace --stats of the run,")
print("# kernels attributed to the phase that launches them, beside bo.py's phase timers")
n_p, w_p = wall(run)
n_q, w_q = wall(plain)
print(f"# under the profiler: {n_p} GP points in {w_p} s; without it: {n_q} GP points in {w_q} s; GPU busy (sum of kernel "
      f"durations) {total:.2f} s = {total / w_q:.0%} of the unprofiled wall" if plain else f"# {n_p} points, {w_p} s")
print(f"{'phase':<48}{'GPU busy s':>11}{'timer s (profiled)':>20}{'timer s (plain)':>17}{'busy / plain timer':>20}")
for p in busy:
    a, b = tp.get(p), tq.get(p)
    ratio = f"{busy[p] / b:.0%}" if b else "-"
    print(f"{p:<48}{busy[p]:>11.3f}{(f'{a:.1f}' if a is not None else '-'):>20}{(f'{b:.1f}' if b is not None else '-'):>17}{ratio:>20}")
print()
for p in busy:
    print(f"## {p}: top kernels")
    for t, c, n in sorted(top[p], reverse=True)[:6]:
        print(f"   {n:<62}{c:>8} calls {t * 1e3:>9.1f} ms {t / c * 1e6:>8.1f} us each")
