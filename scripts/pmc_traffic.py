"""rocprofv3 counter CSVs -> per-kernel HBM bytes per launch.

    python scripts/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [latest.json]

<fetch_dir>/<write_dir>: outputs of two SEPARATE passes
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -o p -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -o p -- python3 bench.py ...
FETCH_SIZE / WRITE_SIZE count KiB (MI355X_MICROARCH.md, HBM section).  On gfx950 FETCH_SIZE reports half the bytes of coalesced streaming
reads (same section: "double it before comparing with a byte count"); checked here on kernels whose bytes are known exactly
(radix_census reads n*12 B = 25.92 GB and reports 12.96 GB; item_scan<write> WRITES 25.92 GB and WRITE_SIZE says 25.92 GB).
Both the raw value and the corrected total (2 * fetch + write) are stored.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    tot = collections.defaultdict(float)
    launches = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mgta::", "")
            tot[name] += float(r["Counter_Value"]) * 1024.0
            launches[name].add(r["Dispatch_Id"])
    return {k: (tot[k], len(launches[k])) for k in tot}


def main():
    fd, wd, out = sys.argv[1:4]
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes per launch; hbm_bytes = 2 * FETCH_SIZE (gfx950 "
                    "correction of MI355X_MICROARCH.md, HBM section) + WRITE_SIZE"}
    latest = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, (0, 1))[0] + wr.get(k, (0, 1))[0])):
        f, nf = fe.get(k, (0.0, 0))
        w, nw = wr.get(k, (0.0, 0))
        n = max(nf, nw, 1)
        res[k] = {"launches": n, "fetch_bytes_per_launch_raw": f / max(nf, 1), "write_bytes_per_launch": w / max(nw, 1),
                  "hbm_bytes_per_launch": 2 * f / max(nf, 1) + w / max(nw, 1)}
        latest[k.split("<")[0] + "_bytes_per_launch"] = 2 * f / max(nf, 1) + w / max(nw, 1)
    json.dump(res, open(out, "w"), indent=1)
    if len(sys.argv) > 4:
        latest["_source"] = out + " (2 * FETCH_SIZE + WRITE_SIZE)"
        json.dump(latest, open(sys.argv[4], "w"), indent=1)
    for k, v in res.items():
        if k != "_note":
            print(f"{k:45s} x{v['launches']:<3d} fetch {v['fetch_bytes_per_launch_raw'] / 1e9:8.2f} GB  write {v['write_bytes_per_launch'] / 1e9:8.2f} GB")


if __name__ == "__main__":
    main()
