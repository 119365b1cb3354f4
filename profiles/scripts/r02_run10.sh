export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r2r; mkdir -p $out
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o t -- python3 $R/profiles/experiments/exp_blur_counters.py > $out/p$i.log 2>&1; echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, statistics
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/root/repo/gpurun_out/r2r/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "blur_sep" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("blur_sep_kernel")[1][:8], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k, {c: round(statistics.median(x)) for c, x in sorted(v.items())})
PY
rm -rf $out/p1 $out/p2 $out/p3
