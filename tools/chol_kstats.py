"""Per-kernel totals of the LAST factorisation batch in a rocprofv3 kernel trace (tools/chol_trace.py run ...).
usage: python tools/chol_kstats.py <kernel_trace.csv> N"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2])
nb = (N + 127) // 128
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bobe::", "")) for r in rows)
# last factorisation: from the last k_kernel_matrix (assembly) on
last_asm = max(i for i, e in enumerate(ev) if "k_kernel_matrix" in e[2])
seg = [e for e in ev[last_asm + 1:] if any(s in e[2] for s in ("k_potf2", "k_trsm_panel", "k_syrk_trail", "k_chol_step"))]
tot = collections.defaultdict(float)
cnt = collections.Counter()
busy = 0.0
for s, e, n in seg:
    tot[n] += (e - s) / 1e3
    cnt[n] += 1
    busy += (e - s) / 1e3
span = (seg[-1][1] - seg[0][0]) / 1e3
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{n:60s} {cnt[n]:4d} launches {v:9.1f} us  avg {v / cnt[n]:7.2f} us")
print(f"span {span:.1f} us, kernel time {busy:.1f} us, gaps {span - busy:.1f} us")
