"""Prints the kernel timeline of the last search in a rocprofv3 --kernel-trace CSV (filter phases + refine)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 17
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ("rescore", "select_candidates", "score_filter", "prep_queries", "scan_topk", "merge_topk")
sel = [r for r in rows if any(s in r["Kernel_Name"] for s in names)]
tail = sel[-n:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{r["Kernel_Name"][:44]:44s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r.get("Grid_Size_X", "")}')
