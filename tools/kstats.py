#!/usr/bin/env python3
"""Prints a rocprofv3 kernel_stats.csv as name (shortened), calls, average us, total ms.  Usage: kstats.py <csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-90s %5s x %10.1f us = %9.3f ms" % (name[:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
