"""per-kernel totals of the LAST solve in a rocprofv3 kernel trace (between two gather_perm/cinv markers)"""
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "set_info_kernel"
rows = list(csv.DictReader(open(path)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
idx = [i for i, k in enumerate(ks) if marker in k[2]]
i0, i1 = idx[-2], idx[-1]
tot = collections.defaultdict(lambda: [0, 0.0])
for k in ks[i0:i1]:
    name = k[2].replace("void spp::", "").replace("spp::", "").split("(")[0]
    tot[name][0] += 1
    tot[name][1] += (k[1] - k[0]) / 1e3
print("span %.1f us, %d kernels" % ((ks[i1][0] - ks[i0][0]) / 1e3, i1 - i0))
for name, (n, t) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%8.1f us  %5d x  %s" % (t, n, name))
