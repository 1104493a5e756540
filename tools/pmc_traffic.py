#!/usr/bin/env python3
"""Bytes behind L2 per GEMM launch from two rocprofv3 PMC passes of the same bench command (separate runs, as the
counter slots demand):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir_f> -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-roofline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir_w> -- python bench.py ... (same)
usage: pmc_traffic.py <dir_f> <dir_w> <out.json>
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests as 64 bytes -> doubled; WRITE_SIZE is
exact for 16-byte-per-lane stores.  Both are in KiB.  Infinity-Cache hits are included: an upper bound on HBM bytes."""
import csv
import glob
import gzip
import json
import shutil
import sys


def per_launch(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    tot, n = 0.0, 0
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter and ("gemm256_kernel" in r["Kernel_Name"] or "gemm_nt_kernel" in r["Kernel_Name"]):
                tot += float(r["Counter_Value"])
                n += 1
    return tot / max(n, 1), n, f


def main():
    df, dw, out = sys.argv[1:4]
    fkb, nf, ff = per_launch(df, "FETCH_SIZE")
    wkb, nw, fw = per_launch(dw, "WRITE_SIZE")
    res = {
        "workload": "bench.py default (mnist shapes, B=320, bf16), GEMM launches of 12 steps",
        "kernel": "gemm256_kernel + gemm_nt_kernel (all gpt_gemm_nt launches)",
        "launches": nf,
        "FETCH_SIZE_KB_per_launch_raw": fkb,
        "WRITE_SIZE_KB_per_launch": wkb,
        "correction": "gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes -> doubled (MI355X_MICROARCH.md, HBM); "
                      "Infinity-Cache hits are included, so this is fabric traffic behind L2, an upper bound on HBM bytes",
        "traffic_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0,
        "collected": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 10 --warmup 2 "
                     "--no-cpu-baseline --no-train --no-roofline",
    }
    json.dump(res, open(out, "w"), indent=1)
    for src, name in ((ff, "fetch"), (fw, "write")):
        with open(src, "rb") as a, gzip.open(out.replace("gemm_traffic.json", f"sample_pmc_{name}_size.csv.gz"), "wb") as b:
            shutil.copyfileobj(a, b)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
