"""Print a rocprofv3 kernel_stats.csv as one line per kernel (name cut at the template arguments)."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")[:70]
    print("%-70s calls %6s avg %8.1f us min %8.1f max %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
