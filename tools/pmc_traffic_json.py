#!/usr/bin/env python3
"""gpurun_out/<tag>/pmc_<workload>_{FETCH,WRITE}_SIZE.csv (tools/gpu_call.sh pmc <workload>) -> profiles/<round>/pmc_dominant_kernel_traffic.json
{workload: {kernel, launches, fetch_size_kb_per_launch, write_size_kb_per_launch, hbm_bytes_per_launch}} for the dominant
kernel of each workload's bench line.  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the
128-byte requests of a 16 B/lane stream at 64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
usage: pmc_traffic_json.py <dir> <out.json>"""
import csv
import json
import os
import sys

DOMINANT = {"guided": "conv_kernel<2, 4, 8, 3, 2, 9, 324, 2, 1, false>", "unguided": "conv_kernel<2, 4, 8, 3, 2, 9, 324, 2, 1, false>",
            "adm256": "conv_kernel<2, 4, 8, 2, 2, 9, 324, 2, 1, false>", "adm128": "conv_kernel<2, 4, 8, 2, 2, 9, 324, 2, 1, false>", "sd": "conv_kernel<2, 4, 8, 2, 2, 9, 324, 2, 1, false>"}


def read(path, kernel, counter):
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["kernel"] == kernel and r["counter"] == counter:
                return float(r["mean_per_launch"]), int(r["launches"])
    return None, 0


def main():
    d, out = sys.argv[1], sys.argv[2]
    res = {}
    if os.path.exists(out):
        res = json.load(open(out))
    for w, k in DOMINANT.items():
        pf, pw = os.path.join(d, f"pmc_{w}_FETCH_SIZE.csv"), os.path.join(d, f"pmc_{w}_WRITE_SIZE.csv")
        if not (os.path.exists(pf) and os.path.exists(pw)):
            continue
        f, n = read(pf, k, "FETCH_SIZE")
        wv, _ = read(pw, k, "WRITE_SIZE")
        if f is None or wv is None:
            continue
        res[w] = {"kernel": k, "launches": n, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": wv,
                  "hbm_bytes_per_launch": (2 * f + wv) * 1024,
                  "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
                  "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python3 bench.py "
                            f"--workload {w} --steps 1 --warmup 0 (tools/gpu_call.sh pmc; {os.path.basename(d)})"}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in res.items()}))


if __name__ == "__main__":
    main()
