#!/bin/bash
# counters of one kernel (substring $1) over an arbitrary python command: one rocprofv3 --pmc pass per counter group,
# averaged per launch.   tools/pmc_cmd.sh <kernel substring> "<groups>" <tag> tools/update_bench.py 5120
k=$1; sel=$2; tag=$3; shift 3
out=$PWD/gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
grp[1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
grp[2]="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
grp[3]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
grp[4]="FETCH_SIZE"
grp[5]="WRITE_SIZE"
grp[6]="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
grp[7]="GRBM_GUI_ACTIVE GRBM_COUNT"
for i in $sel; do
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc ${grp[$i]} -d $out/p$i -o run --output-format csv -- python3 "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; exit 1; }
done
python3 - "$out" "$k" <<'PY'
import sys,glob,csv,collections
out,k=sys.argv[1:3]
dur=collections.defaultdict(lambda:[0,0.0])
for f in sorted(glob.glob(out+"/p*/**/*_kernel_trace.csv",recursive=True))[:1]:
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            d=dur[r["Kernel_Name"][:60]]; d[0]+=1; d[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
for n,(c,s) in dur.items(): print("%-60s launches %3d avg %.1f us (under pmc)"%(n,c,s/c))
for f in sorted(glob.glob(out+"/p*/**/*_counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for c,(n,s) in acc.items(): print("%-36s launches %3d avg %.6g"%(c,n,s/n))
PY
