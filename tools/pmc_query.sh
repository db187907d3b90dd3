#!/bin/bash
# PMC passes over tools/bench_query (first kernel variant = the production kernel)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq_${1:-x}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
P2="SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P4="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CU_CYCLES"
P3="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1));
  rocprofv3 --output-format csv --pmc $P --kernel-trace -d $OUT/p$i -o pmc -- $ROOT/build/bench_query > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in acc:
        if "true, true, 0, 0>" not in k and "true, true, 0, 1>" not in k and "true, true, 0>" not in k: continue
        n=len(cnt[k]); print(k, "launches", n)
        for c,v in sorted(acc[k].items()): print("   %-26s %16.0f" % (c, v/n))
PY
