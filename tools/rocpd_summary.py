#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 rocpd database (the `--kernel-trace --stats` output of ROCm 7.2) as a
small text table, so that the summary can be committed under profiles/ (the .db itself is scratch)."""
import sqlite3
import sys


def main(db_path, out_path=None, note=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations in microseconds)", "# source: " + db_path]
    if note:
        lines.append("# " + note)
    lines.append("calls,total_us,avg_us,percent,vgpr,sgpr,lds_bytes,grid,workgroup,kernel")
    for name, calls, total, avg, pct in rows:
        meta = cur.execute(
            "select vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x, workgroup_y, workgroup_z from kernels "
            "where name = ? limit 1", (name,)).fetchone() or (None,) * 7
        lines.append("%d,%.3f,%.3f,%.2f,%s,%s,%s,%s,%sx%sx%s,\"%s\"" % (calls, total, avg, pct, meta[0], meta[1], meta[2],
                                                                    meta[3], meta[4], meta[5], meta[6], name))
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, " ".join(sys.argv[3:]))
