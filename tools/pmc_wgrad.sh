set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${O:-gpurun_out/r3h}
export O
SHAPE=${SHAPE:-tower}
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -o "TCC_[A-Z0-9_]*\|SQ_[A-Z0-9_]*LDS[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*BUSY[A-Z0-9_]*" $O/counters_list.txt | sort -u | tr '\n' ' ' > $O/counters_short.txt
python3 tools/run_wgrad.py $SHAPE 10 > $O/plain.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT" ; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 tools/run_wgrad.py $SHAPE 10 > $O/p$i.log 2>&1 || echo "pass $i failed" >> $O/fail.txt
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("O", "gpurun_out/r3h")
for d in sorted(glob.glob(O + "/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        tot, calls = collections.defaultdict(float), collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
        for k in sorted(tot):
            print(d, k[0], k[1], "per launch %.4g over %d launches" % (tot[k] / len(calls[k]), len(calls[k])))
PY
