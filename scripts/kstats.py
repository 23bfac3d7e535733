"""print a rocprofv3 *_kernel_stats.csv in readable form: python scripts/kstats.py <csv>"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mgta::", "").replace("mgta::", "")
    if float(r["AverageNs"]) * int(r["Calls"]) > 3e5:
        print(f"{n:40s} calls {r['Calls']:>3s} avg_ms {float(r['AverageNs']) / 1e6:8.3f} pct {r['Percentage']}")
