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
# the dominant kernel launch by launch: the average above covers every launch of the command (clock-settling passes, the count that sizes
# the output, warm-up, the timed steps, the asynchronous batches); bench.py's kernel_ms_avg is the HIP-event mean of the TIMED steps only
try:
    top = c.execute("select name from top_kernels order by total_duration desc limit 1").fetchone()[0]
    durs = [r[0] / 1000.0 for r in c.execute("select duration from kernels where name = ? order by start", (top,))]
    if len(durs) >= 8:
        srt = sorted(durs)
        lines.append("")
        lines.append(f"# {top[:90]}: {len(durs)} launches in start order, us")
        lines.append(f"#   min {srt[0]:.1f}  median {srt[len(srt) // 2]:.1f}  mean {sum(durs) / len(durs):.1f}  p90 {srt[int(0.9 * (len(srt) - 1))]:.1f}  max {srt[-1]:.1f}")
        lines.append(f"#   first 10: {' '.join('%.0f' % d for d in durs[:10])}")
        k = max(1, len(durs) // 4)
        lines.append(f"#   quarters (mean): {' '.join('%.1f' % (sum(durs[i * k:(i + 1) * k]) / k) for i in range(4))}")
except Exception as e:                                               # noqa: BLE001 (schema differences between rocprofv3 versions: the table above is the contract)
    lines.append(f"# (per-launch durations unavailable: {e})")
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
