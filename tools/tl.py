"""annotated kernel timeline of the last dense factorization in a rocprofv3 kernel trace (tools/dense_time.py run)"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
idx = [i for i, k in enumerate(ks) if "set_info_kernel" in k[2]]
i0 = idx[-2]; i1 = idx[-1]; t0 = ks[i0][0]
short = {"potrf_diag": "POTRF", "gemm_tn_staged_kernel<128, 16": "panel", "gemm_tn_staged_kernel<64, 64": "upd64",
         "gemm_tn_staged_kernel<32, 32": "upd32", "gemm_tn_kernel<128, 128": "BULK128", "gemm_tn_kernel<64": "bulk64",
         "gemm_tn_mixed": "BULKMIX", "flag_wait": "wait", "flag_signal_wait": "sigwait", "flag_signal": "sig", "update_potrf": "FUSED"}
prev = None
for k in ks[i0:i1]:
    nm = k[2].replace("void spp::", "").replace("spp::", "")
    for a, b in short.items():
        if nm.startswith(a):
            nm = b
    nm = nm[:30]; extra = ""
    if nm == "POTRF":
        if prev is not None:
            extra = " step %.1f" % ((k[0] - prev) / 1e3)
        prev = k[0]
    print("%8.1f %8.1f %6.1f q%s %s%s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], nm, extra))
