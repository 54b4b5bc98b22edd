#!/usr/bin/env python3
"""Instruction-issue picture of the ksw_extd2 kernels from rocprofv3 --pmc passes (SQ counters; one pass holds 8 of them):

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \\
        --kernel-trace --output-format csv -d <dir> -o x -- python3 tools/bench_ksw.py [--long] N
    python tools/pmc_issue.py <dir>/x_counter_collection.csv [cells per launch set] > profiles/...json

Derivation (MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues in 2 cycles when another wave can fill in, 4 alone;
GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles):
  gpu_cycles           = GRBM_GUI_ACTIVE / 8
  valu_issue_capacity  = 1024 SIMDs x gpu_cycles / 2        wave-instructions
  valu_utilisation     = SQ_INSTS_VALU / valu_issue_capacity
  issue_per_simd_cycle = (SQ_INSTS_VALU + SQ_INSTS_SALU) / (1024 x gpu_cycles)
  waves_per_simd       = 4 x SQ_WAVE_CYCLES / (1024 x gpu_cycles)
"""
import csv, json, re, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in rows:
    m = re.search(r'(ksw_\w+(<[^>]*>)?)', r["Kernel_Name"])
    if not m or "gather" in m.group(1):
        continue
    k = m.group(1)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in disp[k]:
        disp[k].add(r["Dispatch_Id"])
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
out = {"source": "rocprofv3 --pmc (SQ issue counters) --kernel-trace -- python3 " + (sys.argv[2] if len(sys.argv) > 2 else "tools/bench_ksw.py"), "kernels": {}}
for k, c in acc.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    simd_cyc = 1024.0 * cyc
    d = {"launches": len(disp[k]), "total_ms": round(dur[k], 3), "counters": {a: b for a, b in c.items()},
         "gpu_cycles": cyc, "clock_ghz": round(cyc / (dur[k] * 1e6), 3) if dur[k] else None,
         "valu_utilisation": round(c["SQ_INSTS_VALU"] / (simd_cyc / 2.0), 4),
         "instructions_issued_per_simd_cycle": round((c["SQ_INSTS_VALU"] + c.get("SQ_INSTS_SALU", 0.0)) / simd_cyc, 4),
         "salu_per_valu": round(c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"], 3),
         "waves_per_simd": round(4.0 * c["SQ_WAVE_CYCLES"] / simd_cyc, 2),
         "wave_time_share": {"waiting_on_counters(s_waitcnt,barrier)": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                             "issue_stalled": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                             "valu_active": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 3),
                             "scalar_active": round(c.get("SQ_ACTIVE_INST_SCA", 0.0) / c["SQ_WAVE_CYCLES"], 3)}}
    out["kernels"][k] = d
print(json.dumps(out, indent=1))
