#!/usr/bin/env python3
"""Build profiles/*_hbm_traffic_pmc.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
separate runs, as MI355X_MICROARCH.md prescribes): per kernel, the per-launch average of
(2 * FETCH_SIZE + WRITE_SIZE) KB -- FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE exact.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command profiled>"
"""
import csv, json, sys, collections


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        tot[name] += float(r["Counter_Value"])
        n[name].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(n[k]), len(n[k])) for k in tot}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (one run) and --pmc WRITE_SIZE (a second run) over `%s`; per-launch "
                  "average over all dispatches of each kernel; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 -- FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads), WRITE_SIZE exact." % sys.argv[4],
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out["kernels"][k] = {"FETCH_SIZE_KB_avg_per_launch": round(f, 2), "launches_FETCH_SIZE": nf,
                             "WRITE_SIZE_KB_avg_per_launch": round(w, 2), "launches_WRITE_SIZE": nw,
                             "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main()
