#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc SQ_* counter_collection.csv: average per dispatch of every counter plus the
derived ratios MFMA-busy / busy cycles, LDS bank-conflict share, waves parked (WAIT_ANY) and issue-stalled (WAIT_INST_ANY).

usage: sq_summary.py <counter_collection.csv> <out.csv>
"""
import collections, csv, sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[name].add(r["Dispatch_Id"])
    cols = sorted({c for v in acc.values() for c in v})
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches"] + [c + "_avg" for c in cols] +
                   ["mfma_busy_over_busy_cycles", "mfma_busy_frac_of_simd_cycles", "lds_bank_conflict_share", "wait_any_over_wave_cycles",
                    "wait_inst_any_over_wave_cycles"])
        for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)):
            n = len(disp[k])
            v = acc[k]
            def ratio(a, b):
                return round(v.get(a, 0.0) / v[b], 4) if v.get(b) else ""
            w.writerow([k, n] + [round(v.get(c, 0.0) / n, 1) for c in cols] +
                       # SQ_VALU_MFMA_BUSY_CYCLES sums the matrix-pipe cycles of all 1024 SIMDs (checked: = MFMAs x 16 for the
                       # 16x16x32 form); SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles: busy share of one SIMD = ratio / 32
                       [ratio("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"),
                        round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / v["SQ_BUSY_CYCLES"] / 32.0, 4) if v.get("SQ_BUSY_CYCLES") else "",
                        ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
                        ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")])
    print(open(sys.argv[2]).read())


if __name__ == "__main__":
    main()
