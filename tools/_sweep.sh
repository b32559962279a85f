for a in 0 3200 3600 4200 6000; do SPP_WAIT_MID_BELOW=$a timeout -k 5 120 python tools/dense_time.py 5226 40 2>&1 | grep -v amdgpu.ids; done
SPP_EARLY_ABOVE=3600 timeout -k 5 120 python tools/dense_time.py 5226 40 2>&1 | grep -v amdgpu.ids
SPP_EARLY_ABOVE=3600 SPP_WAIT_MID_BELOW=3600 timeout -k 5 120 python tools/dense_time.py 5226 40 2>&1 | grep -v amdgpu.ids
