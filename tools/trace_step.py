"""Scratch: print the kernel sequence of one bench step from a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# find the last-but-2 occurrence of combine_reduce and print from previous reduce end to this reduce end
idx = [i for i, n in enumerate(names) if "combine_reduce" in n]
a, b = idx[-4], idx[-3]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
print("step span", (int(rows[b]["End_Timestamp"]) - t0) / 1e3, "us")
