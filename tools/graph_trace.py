"""Prints the kernel sequence of one hipGraph replay of the bench step from a rocprofv3 kernel trace (CSV)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = idx[20], idx[21]
t0 = int(rows[a]["End_Timestamp"])
tot = 0
for r in rows[a + 1:b + 1]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot += d
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us +{d/1e3:6.1f}  {r['Kernel_Name'][:100]}")
print("kernels", b - a, "sum of kernel durations", tot / 1e3, "us; span", (int(rows[b]["End_Timestamp"]) - t0) / 1e3)
