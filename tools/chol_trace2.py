"""Per-block timeline of the strip-lookahead Cholesky from a rocprofv3 kernel trace (run with tools/chol_trace.py run N B)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N, B = int(sys.argv[2]), int(sys.argv[3])
nb = (N + 127) // 128
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bobe::", "").replace("bobe::", "")) for r in rows)
last = max(i for i, e in enumerate(ev) if "k_kernel_matrix" in e[2])
seg = [e for e in ev[last + 1:] if e[2].startswith(("k_chol_strip", "k_strip_update", "k_syrk_trail", "k_copy_diag"))]
print(f"{'k':>3} {'stripA':>8} {'stripupd':>8} {'stripB':>8} {'colupd':>8} | {'upd GF':>7} {'TF/s if bound':>13}")
i = 0
tot = [0.0] * 4
for k in range(nb):
    d = []
    for name in ("k_chol_strip", "k_strip_update", "k_chol_strip", "k_syrk_trail"):
        if i < len(seg) and seg[i][2].startswith(name):
            d.append((seg[i][1] - seg[i][0]) / 1e3)
            i += 1
        else:
            d.append(0.0)
    rem = nb - 1 - k
    fl = B * 128.0 * (rem * 128.0) ** 2 if k > 0 else 0.0
    for j in range(4):
        tot[j] += d[j]
    if k % 2 == 1 or k < 2 or k == nb - 1:
        print(f"{k:3d} {d[0]:8.2f} {d[1]:8.2f} {d[2]:8.2f} {d[3]:8.2f} | {fl / 1e9:7.3f} {fl / ((d[0] + d[2]) * 1e-6) / 1e12 if d[0] + d[2] else 0:13.2f}")
print("totals us: stripA %.1f stripupd %.1f stripB %.1f colupd %.1f  sum %.1f; span %.1f" % (*tot, sum(tot), (seg[-1][1] - seg[0][0]) / 1e3))
