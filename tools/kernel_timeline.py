import csv, glob, sys, os
files = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)
f = max(files, key=os.path.getsize)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# print last 40 kernels with relative times
base = int(rows[-60]["Start_Timestamp"])
for r in rows[-60:-20]:
    s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
    print("%9.1f %9.1f  %6.1f  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
