cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=1
for grp in "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE TCC_TAG_STALL_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  OUT=gpurun_out/pmc_l1/p$i; mkdir -p $OUT
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $OUT -- python3 tools/prof_spmv.py --reps 3 > $OUT.log 2>&1 || echo "pass $i failed"
  python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csr_spmv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print("%-44s %.4g" % (k, sum(v)/len(v)))
PY
  i=$((i+1))
done
