#!/usr/bin/env python3
"""Per-kernel stall-reason table from rocprofv3 PMC passes (rocpd .db files, one pass per 8 SQ counters):
    python tools/stall_table.py gpurun_out/r3a/st1/pmc_results.db gpurun_out/r3a/st2/pmc_results.db ... > profiles/r03_stalls.txt
Counters are summed over all SIMDs / shader engines by rocprofv3 and averaged per dispatch here.  Reading
(MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_WAVE_CYCLES = wave-resident cycles summed over waves;
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (has an instruction, cannot issue: MFMA operand dependency, pipe
busy) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES; WAIT_INST_LDS is the part of WAIT_INST_ANY stalled on LDS issue."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name).replace('v2x::', '').replace('void ', '')
    return name[:48]


def main(paths):
    data = {}
    for p in paths:
        c = sqlite3.connect(p)
        for n, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
            if not short(n).startswith('k_'):
                continue
            a = data.setdefault(short(n), {}).setdefault(cn, [0, 0.0])
            a[0] += 1
            a[1] += val
    kernels = sorted(data, key=lambda k: -data[k].get('SQ_WAVE_CYCLES', [1, 0])[1] / max(1, data[k].get('SQ_WAVE_CYCLES', [1, 0])[0]))
    avg = lambda k, c: (data[k][c][1] / data[k][c][0]) if c in data[k] and data[k][c][0] else None
    print("# stall reasons per kernel, headline step (20 links x 64 features x batch 4096, per-node weights), one MI355X, eager launches")
    print("# per-dispatch averages; fractions are of SQ_WAVE_CYCLES (wave-resident cycles summed over the launch's waves)")
    cols = [("WAIT_ANY", "SQ_WAIT_ANY"), ("WAIT_INST_ANY", "SQ_WAIT_INST_ANY"), ("of which LDS", "SQ_WAIT_INST_LDS"), ("of which VMEM", "SQ_WAIT_INST_VMEM"),
            ("ACTIVE_ANY", "SQ_ACTIVE_INST_ANY"), ("act VALU", "SQ_ACTIVE_INST_VALU"), ("act LDS", "SQ_ACTIVE_INST_LDS"),
            ("act VMEM", "SQ_ACTIVE_INST_VMEM"), ("act SCA", "SQ_ACTIVE_INST_SCA"), ("act MISC", "SQ_ACTIVE_INST_MISC")]
    print("%-42s %14s %12s " % ("kernel", "WAVE_CYCLES", "SQ_BUSY") + " ".join("%13s" % c[0] for c in cols))
    for k in kernels:
        wc = avg(k, 'SQ_WAVE_CYCLES')
        if not wc:
            continue
        row = "%-42s %14.0f %12.0f " % (k, wc, avg(k, 'SQ_BUSY_CYCLES') or 0)
        for _, c in cols:
            v = avg(k, c)
            row += " %12s" % ("-" if v is None else "%.1f%%" % (100.0 * v / wc))
        print(row)
    print("\n# instruction counts per dispatch (all waves) and matrix-pipe figures")
    ic = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS",
          "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INST_CYCLES_VMEM"]
    print("%-42s " % "kernel" + " ".join("%16s" % c.replace("SQ_", "")[:16] for c in ic) + "   mfma_busy/(32*SQ_BUSY)")
    for k in kernels:
        row = "%-42s " % k
        for c in ic:
            v = avg(k, c)
            row += " %16s" % ("-" if v is None else "%.0f" % v)
        mb, sb = avg(k, 'SQ_VALU_MFMA_BUSY_CYCLES'), avg(k, 'SQ_BUSY_CYCLES')
        row += "   %s" % ("-" if not mb or not sb else "%.1f%%" % (100.0 * mb / (32 * sb)))
        print(row)


if __name__ == "__main__":
    main(sys.argv[1:])
