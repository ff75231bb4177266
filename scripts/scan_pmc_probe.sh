#!/bin/bash
# scripts/scan_pmc_probe.sh -- SQ counters of the packed scan alone (scripts/scan_alone.py), one rocprofv3 --pmc pass per
# counter group; summary -> gpurun_out/${PACK_PMC_TAG:-r6}/scan_pmc_summary.csv
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${PACK_PMC_TAG:-r6}/scanpmc
mkdir -p $OUT
B="python $ROOT/scripts/scan_alone.py 3.0"
i=0
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN")
for set in "${SETS[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --output-format csv --kernel-include-regex "scan_packed" -d $OUT/s$i -o p -- $B > $OUT/s$i.log 2>&1)
done
python scripts/pmc_summary.py $ROOT/gpurun_out/${PACK_PMC_TAG:-r6}/scan_pmc_summary.csv $OUT/s*/
find $OUT -name '*counter_collection.csv' -size +2M -delete
cat $ROOT/gpurun_out/${PACK_PMC_TAG:-r6}/scan_pmc_summary.csv
