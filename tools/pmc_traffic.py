"""HBM traffic of one kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the bench command.

MI355X_MICROARCH.md section HBM: the counters come in KB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE reads the bytes exactly.
    python tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel substring> N C chunk ["<command>" "<collected: round / head>"]
        > profiles/traffic_k_<name>.json
The two passes:  rocprofv3 --pmc FETCH_SIZE -d <fetch_dir> --output-format csv -- python3 bench.py --no-secondary
                 rocprofv3 --pmc WRITE_SIZE -d <write_dir> --output-format csv -- python3 bench.py --no-secondary
bench.py reads the file back as roofline.traffic (and says so in roofline.traffic_source).
"""
import csv
import glob
import json
import sys


def per_launch(d, counter, kern):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    vals = []
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return sum(vals) / max(len(vals), 1), len(vals)


fetch_dir, write_dir, kern = sys.argv[1:4]
N, C, chunk = map(int, sys.argv[4:7])
command = sys.argv[7] if len(sys.argv) > 7 else "python3 bench.py --no-secondary"
collected = sys.argv[8] if len(sys.argv) > 8 else ""
f_kb, nf = per_launch(fetch_dir, "FETCH_SIZE", kern)
w_kb, nw = per_launch(write_dir, "WRITE_SIZE", kern)
out = {"kernel": kern, "N": N, "C": C, "chunk": chunk, "launches_seen": [nf, nw],
       "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
       "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
       "correction": "gfx950: FETCH_SIZE x2 (counts 128-B requests at 64 B), WRITE_SIZE exact; KB -> bytes",
       "command": command, "collected": collected}
print(json.dumps(out, indent=1))
