import csv, glob, sys
f = glob.glob(f'gpurun_out/prof_{sys.argv[1]}/*/*kernel_stats.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    print("%-52s calls=%6s total_ms=%9.3f avg_us=%9.2f pct=%s" % (r['Name'][:52], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
