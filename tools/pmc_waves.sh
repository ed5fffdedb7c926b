#!/bin/bash
# Diagnostic: where the waves of the loop's kernels spend their cycles (one --pmc pass per group; gfx950 has 8 SQ slots).
# usage: bash tools/pmc_waves.sh [one_align arguments]   -> per kernel and counter: dispatches, average per dispatch
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc_w_$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_w_$i -o p -- python3 $GRAFT_REPO_ROOT/tools/one_align.py "$@" > /tmp/pmc_w_$i.log 2>&1
  echo "pass $i rc $?"
  f=$(find /tmp/pmc_w_$i -name '*counter_collection.csv' | head -n 1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "nn_grid" in k or "assoc_reduce" in k or "loop_step" in k or "p2l_reduce" in k:
        k = k.split("(")[0].replace("void icpk::", "")
        a = acc[(k, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print(f"{k:45s} {c:28s} n {n:4d} avg {v / n:14.1f}")
PY
done
