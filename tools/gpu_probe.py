"""quick MI355X probe: measured copy bandwidth and MFMA f64 rate (used to sanity-check the peaks
quoted in bench.py's roofline)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_plus_plus_amd import api
ctx = api.Context(0, api.FLAG_PROFILE)
print("version", ctx.lib.spp_version().decode())
print("copy GB/s (1 GiB, read+write):", ctx.microbench_copy(1 << 30, 10))
print("mfma f64 16x16x4 TFLOP/s:", ctx.microbench_mfma_f64(4000))
