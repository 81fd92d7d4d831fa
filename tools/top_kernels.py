"""Prints the top rows of a rocprofv3 kernel_stats csv: python tools/top_kernels.py <csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:k]:
    print("%-100s calls %5s avg %9.1f us %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                    100 * float(r["TotalDurationNs"]) / tot))
