"""Per-kernel summary (calls, total ms, average us, share) of a rocprofv3 --kernel-trace --stats run stored in rocpd
(sqlite) format:  python profiles/rocpd_stats.py gpurun_out/evprof/ev_results.db > profiles/r01_evalnet_kernel_stats.csv"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("""select s.kernel_name, count(*), sum(d.end - d.start) / 1e6, avg(d.end - d.start) / 1e3
                          from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                          group by s.kernel_name order by 3 desc"""))
tot = sum(r[2] for r in rows)
print("kernel,calls,total_ms,avg_us,share_pct")
for name, calls, ms, us in rows:
    short = name
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:   # Itanium mangling: <length><identifier>, then I<template args>E for templates
        n, p = int(m.group(1)), m.end()
        short, rest = name[p:p + n], name[p + n:]
        if rest.startswith("I"):
            short += "<" + ",".join(re.findall(r"L[ib](\d+)E", rest.split("EEv")[0] + "E")) + ">"
    print(f"\"{short[:90]}\",{calls},{ms:.3f},{us:.1f},{100 * ms / tot:.1f}")
