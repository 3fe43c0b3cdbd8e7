#!/usr/bin/env python3
"""Reduce the rocprofv3 output directories of tools/r03_profile.sh to the small files kept under profiles/:
per section (dega / cfg3 / lzmh) `<tag>_<section>_kernel_stats.csv` (the dg:: rows of kernel_stats) and
`<tag>_<section>_pmc.csv` (every counter of every pass, one row per dispatch of a dg:: coder kernel), plus a text summary
on stdout: average duration, counters per launch, HBM traffic with the gfx950 correction (FETCH_SIZE x 2, KiB units:
/opt/skills/guides/MI355X_MICROARCH.md)."""
import csv
import glob
import os
import sys
from collections import defaultdict

KEEP = ("dega_encode_kernel", "dega_decode_kernel", "lzmh_encode_kernel", "lzmh_decode_kernel")


def short(name):
    for k in KEEP:
        if k in name:
            return k
    return None


def main():
    root = sys.argv[1]
    tag = os.path.basename(root.rstrip("/")).replace("prof", "")
    out_dir = os.path.join(root, "reduced")
    os.makedirs(out_dir, exist_ok=True)
    for section in ("dega", "cfg3", "lzmh"):
        stats = glob.glob(os.path.join(root, section + "_stats", "**", "*kernel_stats.csv"), recursive=True)
        if not stats:
            continue
        print("== %s ==" % section)
        dur = {}
        with open(stats[0]) as f, open(os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (tag, section)), "w") as g:
            rd = csv.reader(f)
            wr = csv.writer(g, quoting=csv.QUOTE_NONNUMERIC)
            for i, row in enumerate(rd):
                if i == 0 or "dg::" in row[0]:
                    wr.writerow(row)
                    if i and short(row[0]):
                        dur[short(row[0])] = (int(row[1]), float(row[3]) * 1e-6)
                        print("  %-20s calls %3d  avg %.3f ms   %s" % (short(row[0]), int(row[1]), float(row[3]) * 1e-6, row[0][:90]))
        rows = []
        agg = defaultdict(lambda: defaultdict(list))
        for pas in ("fetch", "write", "valu", "lds", "sq"):
            for path in glob.glob(os.path.join(root, "%s_%s" % (section, pas), "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for r in csv.DictReader(f):
                        k = short(r["Kernel_Name"])
                        if not k:
                            continue
                        rows.append([pas, r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                                     r["SGPR_Count"], r["Counter_Name"], r["Counter_Value"]])
                        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(os.path.join(out_dir, "%s_%s_pmc.csv" % (tag, section)), "w") as g:
            wr = csv.writer(g)
            wr.writerow(["Pass", "Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"])
            wr.writerows(rows)
        for k, cs in agg.items():
            print("  %s: per launch (mean over %s dispatches)" % (k, ",".join(sorted(set(str(len(v)) for v in cs.values())))))
            for name in sorted(cs):
                v = sum(cs[name]) / len(cs[name])
                print("    %-22s %.6g" % (name, v))
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                fe = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2
                wr_ = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
                print("    HBM traffic per launch: 2 x FETCH_SIZE = %.3f GB + WRITE_SIZE = %.3f GB = %.3f GB" % (fe / 1e9, wr_ / 1e9, (fe + wr_) / 1e9))
            if "SQ_INSTS_VALU" in cs and k in dur:
                iv = sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"])
                print("    SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / 2.25 GHz = %.2f ms of %.2f ms" % (iv / 1024 * 4 / 2.25e9 * 1e3, dur[k][1]))


if __name__ == "__main__":
    main()
