#!/usr/bin/env python3
"""HBM traffic of the DP kernels from two rocprofv3 --pmc passes over the bench command
(FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv > profiles/r01_pmc_ksw_traffic.json

Units: rocprofv3 reports both counters in KiB-like units of 1 KB = 1024 B?  No: the raw value is in kilobytes as
derived by rocprof (TCC requests x 64 B / 1024); the guide's gfx950 correction doubles FETCH_SIZE for wide coalesced
reads (narrow reads are uncalibrated, so 2x is an upper bound) and takes WRITE_SIZE as is."""
import csv
import json
import sys


def total(path, counter, pat="ksw_extd2"):
    tot, disp = 0.0, set()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter or pat not in r["Kernel_Name"]:
                continue
            tot += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return tot, len(disp)


fetch_kb, n1 = total(sys.argv[1], "FETCH_SIZE")
write_kb, n2 = total(sys.argv[2], "WRITE_SIZE")
launches = max(n1, n2)
fetch_b, write_b = fetch_kb * 1024.0, write_kb * 1024.0
print(json.dumps({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0",
    "kernel": "ksw_extd2_reg_kernel<NW,NCH> (every register class; all DP launches of one cfg2 step)",
    "launches": launches,
    "fetch_bytes_raw": fetch_b,
    "fetch_bytes_x2_bound": 2 * fetch_b,
    "write_bytes": write_b,
    "traffic_bytes_per_launch": (2 * fetch_b + write_b) / max(launches, 1),
    "note": "WRITE_SIZE is the traceback scratch (1 B per computed DP cell, ~2e11 cells per step); FETCH_SIZE doubled per "
            "MI355X_MICROARCH.md (gfx950 halves wide coalesced reads; narrow reads are uncalibrated, so 2x is an upper bound)",
}, indent=1))
