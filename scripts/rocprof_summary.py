#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 (rocpd sqlite) result into a small text summary.

    python scripts/rocprof_summary.py gpurun_out/prof_r01/ip1g_results.db profiles/r01_xxx.txt "<command line profiled>"
"""
import sqlite3
import sys

db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
c = sqlite3.connect(db)
lines = ["# rocprofv3 --kernel-trace --stats summary", f"# command: {cmd}", f"# source db: {db}", ""]
lines.append(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}")
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append(f"{name[:100]:100s} {calls:6d} {total:12.3f} {avg:12.3f} {pct:7.2f}")
lines.append("")
lines.append("# per-dispatch resources (first dispatch of each kernel)")
seen = set()
for row in c.execute("select name,grid_x,workgroup_x,lds_size,vgpr_count,accum_vgpr_count,sgpr_count,scratch_size,duration from kernels order by start"):
    if row[0] in seen:
        continue
    seen.add(row[0])
    lines.append(f"{row[0][:80]:80s} grid={row[1]} wg={row[2]} lds={row[3]} vgpr={row[4]} agpr={row[5]} sgpr={row[6]} scratch={row[7]} dur_ns={row[8]}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
