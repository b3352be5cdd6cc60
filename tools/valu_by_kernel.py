"""Summarises a rocprofv3 --pmc run (SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE ...) over tools/bench_ops.py:
per kernel, VALU wave-instructions per launch and the time the chip needs just to issue them
(4 cycles per wave64 instruction on each of 1024 SIMDs at 2.4 GHz), next to the kernel's duration from
a --kernel-trace run of the same command.  usage: valu_by_kernel.py <pmc_dir> <trace_dir>"""
import csv
import glob
import re
import sys
from collections import defaultdict

pmc_dir, trace_dir = sys.argv[1], sys.argv[2]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = {}
for f in glob.glob(trace_dir + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Name"]] = (float(row["AverageNs"]), int(row["Calls"]))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


print("%-70s %10s %12s %10s %8s" % ("kernel", "avg us", "VALU insts", "issue us", "share"))
for k in sorted(vals):
    if "mifc" not in k:
        continue
    c = vals[k]
    if "SQ_INSTS_VALU" not in c:
        continue
    n = len(c["SQ_INSTS_VALU"])
    insts = sum(c["SQ_INSTS_VALU"]) / n
    issue_us = insts * 4.0 / 1024.0 / 2400.0
    d = dur.get(k, (0.0, 0))[0] / 1000.0
    print("%-70s %10.1f %12.3e %10.1f %8.2f" % (short(k), d, insts, issue_us, issue_us / d if d else 0.0))
