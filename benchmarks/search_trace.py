#!/usr/bin/env python3
"""One query over a resident store N times (default route), for a rocprofv3 --kernel-trace run; --report DIR prints the last search's
kernels with their durations and the gaps in front of them.  ROWS (default 100000), K (10), NQ (1) from the environment."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(d):
    f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "copyBuffer" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]]
    # the last search: from the last kernel whose predecessor ended more than 20 us earlier
    start = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 20000:
            start = i
    t0, prev = int(rows[start]["Start_Timestamp"]), None
    for r in rows[start:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev is None else (s - prev) / 1e3
        print(f'{r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]:70s} start {(s - t0) / 1e3:7.1f} us  dur {(e - s) / 1e3:6.2f}  gap {gap:5.2f}  grid {r.get("Grid_Size", "")}')
        prev = e
    print(f"# span {(prev - t0) / 1e3:.1f} us")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    import time

    import numpy as np

    from codesearch_amd import VectorStore
    from codesearch_amd.synth import synth_rows

    n, k, nq = int(os.environ.get("ROWS", 100000)), int(os.environ.get("K", 10)), int(os.environ.get("NQ", 1))
    st = VectorStore(None, 384)
    st.insert_synthetic(n, 77, 0)
    st.build_index()
    q = synth_rows(5, 0, nq, 384)
    for _ in range(30):
        st.search_raw(q if nq > 1 else q[0], k)
        time.sleep(0.0005)


if __name__ == "__main__":
    main()
