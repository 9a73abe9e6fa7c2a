#!/usr/bin/env python3
"""Condenses a `rocprofv3 --kernel-trace --stats --output-format csv` output directory into the per-kernel summary kept
under profiles/ (durations in microseconds, launch geometry from the kernel trace).
Usage: stats_summary.py <rocprof output dir> <out.csv> "<command line that was profiled>" """
import csv
import glob
import sys


def main():
    src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    stats = sorted(glob.glob(src + "/*/*kernel_stats.csv"))[-1]
    trace = sorted(glob.glob(src + "/*/*kernel_trace.csv"))[-1]
    geo = {}
    for r in csv.DictReader(open(trace)):
        geo.setdefault(r["Kernel_Name"], (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                                          "x".join(r["Grid_Size_" + a] for a in "XYZ"),
                                          "x".join(r["Workgroup_Size_" + a] for a in "XYZ")))
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds), MI355X\n")
        o.write("# command: " + cmd + "\n")
        o.write("calls,total_us,avg_us,percent,min_us,max_us,vgpr,agpr,sgpr,lds_bytes,grid,workgroup,kernel\n")
        for r in csv.DictReader(open(stats)):
            g = geo.get(r["Name"], ("",) * 6)
            o.write('%s,%.3f,%.3f,%s,%.3f,%.3f,%s,%s,%s,%s,%s,%s,"%s"\n' %
                    (r["Calls"], int(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"],
                     int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3, g[0], g[1], g[2], g[3], g[4], g[5], r["Name"][:260]))
    print("wrote", out)


if __name__ == "__main__":
    main()
