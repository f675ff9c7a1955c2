"""Sequence of factorisation kernels (name, duration) of the last batch in a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bobe::", "").replace("bobe::", ""), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows)
last = max(i for i, e in enumerate(ev) if "k_kernel_matrix" in e[2])
seg = [e for e in ev[last + 1:] if e[2].startswith(("k_chol", "k_strip", "k_syrk", "k_copy_diag", "k_potf2", "k_trsm"))]
t0 = seg[0][0]
for s, e, n, g in seg[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.2f} us  grid {g:6d}  {n}")
print("span", (seg[-1][1] - t0) / 1e3)
