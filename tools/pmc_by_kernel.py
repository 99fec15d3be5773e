#!/usr/bin/env python3
"""Mean of every counter per (kernel, grid size, workgroup size) from rocprofv3 counter_collection CSVs, plus the mean
duration.  usage: python tools/pmc_by_kernel.py <pattern> file1.csv [file2.csv ...]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("dfe::", "")


def main():
    pat = re.compile(sys.argv[1])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[2:]:
        with open(path) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if not pat.search(k):
                    continue
                key = (k, int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"]))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[key]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key in sorted(acc):
        print("%s grid=%d wg=%d lds=%d vgpr=%d" % key)
        for c, v in sorted(acc[key].items()):
            print("    %-28s %14.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
