"""Prints a rocprofv3 *_kernel_stats.csv as a short table (name, calls, average us, share)."""
import csv
import sys

for row in csv.DictReader(open(sys.argv[1])):
    name = row["Name"].split("(")[0][-44:]
    print(f"{name:44s} calls={int(row['Calls']):6d} avg={float(row['AverageNs']) / 1e3:9.2f} us "
          f"share={float(row['Percentage']):6.2f}%")
