"""Per-kernel MFMA utilisation from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
GRBM_GUI_ACTIVE --kernel-trace --output-format csv` counter_collection.csv.

  python profiles/summarize_mfma_pmc.py profiles/r01d_encoder_pmc_mfma_busy.csv

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the fraction of all
SIMD-cycles of the dispatch in which the matrix pipe was executing (the counter adds 32 per
v_mfma_f32_32x32x16_f16 per SIMD, MI355X_MICROARCH.md cycle-constants table); effective clock =
GRBM_GUI_ACTIVE / 8 / dispatch wall time.
"""
import collections
import csv
import sys


def main(path):
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = disp[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["name"] = r["Kernel_Name"]
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0, 0.0, 0.0])
    for d in disp.values():
        a = agg[d["name"].split("(")[0][:80]]
        a[0] += 1
        a[1] += d["ns"]
        a[2] += d.get("GRBM_GUI_ACTIVE", 0.0)
        a[3] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print("kernel,calls,avg_us,effective_clock_GHz,mfma_busy_fraction")
    for name, (n, ns, gui, mfma) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if mfma == 0 or n < 2:
            continue
        cyc = gui / 8
        print(f"{name},{n},{ns / n / 1e3:.1f},{cyc / ns:.3f},{mfma / (cyc * 1024):.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
