"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
profiles/pmc_traffic.json and a per-kernel csv.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <workload> <kernel substring> [out.json] [out.csv] [round tag]

bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE / WRITE_SIZE are in KiB, and on
gfx950 FETCH_SIZE reports half of the 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    path = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    fdir, wdir, workload, sub = sys.argv[1:5]
    out_json = sys.argv[5] if len(sys.argv) > 5 else "profiles/pmc_traffic.json"
    out_csv = sys.argv[6] if len(sys.argv) > 6 else None
    source = sys.argv[7] if len(sys.argv) > 7 else "an earlier profiling run"
    f, w = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        rows.append((k, nf, sf / max(nf, 1), nw, sw / max(nw, 1), (2 * sf / max(nf, 1) + sw / max(nw, 1)) * 1024))
    if out_csv:
        with open(out_csv, "w") as fo:
            fo.write('"kernel","launches_fetch_pass","FETCH_SIZE_KiB_avg","launches_write_pass","WRITE_SIZE_KiB_avg","hbm_bytes_per_launch(2F+W)"\n')
            for r in rows:
                fo.write('"%s",%d,%.3f,%d,%.3f,%.0f\n' % r)
    dom = [r for r in rows if sub in r[0]]
    dom.sort(key=lambda r: -r[1] * r[5])   # the variant that moves the most bytes in total
    k = dom[0]
    cal = [r for r in rows if "ctile_rw_kernel" in r[0]]
    js = json.load(open(out_json)) if os.path.exists(out_json) else {}
    js[workload] = {
        "kernel": k[0], "bytes_per_launch": k[5], "fetch_size_kb_avg": k[2], "write_size_kb_avg": k[4],
        "launches_profiled": k[1], "source": "round " + source,
        "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (with --kernel-trace only); "
               "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE reports half of 16-B/lane streaming reads, "
               "MI355X_MICROARCH.md HBM section); tools/pmc_summary.py"}
    if cal:
        # known byte count in the kernel's own C-tile access pattern (8 B/lane): 8 * n * n each way, n = 8192
        known = 8.0 * 8192 * 8192
        js[workload]["calibration_ctile_8B_per_lane"] = {
            "known_bytes_each_way": known, "FETCH_SIZE_KiB_avg": cal[0][2], "WRITE_SIZE_KiB_avg": cal[0][4],
            "true_over_FETCH_SIZE": known / (cal[0][2] * 1024), "true_over_WRITE_SIZE": known / (cal[0][4] * 1024)}
    json.dump(js, open(out_json, "w"), indent=1)
    print(json.dumps(js[workload], indent=1))


if __name__ == "__main__":
    main()
