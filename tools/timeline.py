"""prints the kernel timeline of the last Lambda-solve found in a rocprofv3 kernel_trace.csv"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
import os
anchor = os.environ.get("TL_ANCHOR", "cinv_kernel")
idx = [i for i, k in enumerate(ks) if anchor in k[2]]
i0 = idx[int(sys.argv[2]) if len(sys.argv) > 2 else -2]
t0 = ks[i0][0]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for k in ks[i0 + skip:i0 + skip + n]:
    name = k[2].replace("void spp::", "").replace("spp::", "")[:46]
    print("%9.1f %9.1f dur %7.1f q%s %s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], name))
