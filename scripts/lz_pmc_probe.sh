#!/bin/bash
# scripts/lz_pmc_probe.sh TAG -- SQ / memory counters of the step's kernels (LZ parses, scan, key filter, split point), one
# rocprofv3 --pmc pass per counter group of `bench.py --steps 2 --warmup 1`; summary -> gpurun_out/TAG/lz_pmc_summary.csv
TAG=${1:-r4}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG/lzpmc
mkdir -p $OUT
B="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
if [ -n "$LZ_PMC_QUICK" ]; then
  SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE")
else
  SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_CYCLES")
fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --output-format csv --kernel-include-regex "${LZ_PMC_KERNELS:-lz_parse_kernel|lz_encode_grp|scan_packed|key_filter|split_point|idx_|ref_pack}" -d $OUT/s$i -o p -- $B > $OUT/s$i.log 2>&1)
  tail -c 300 $OUT/s$i.log | grep -o '"value": [0-9.]*' | head -1
done
python scripts/pmc_summary.py $ROOT/gpurun_out/$TAG/lz_pmc_summary.csv $OUT/s*/ 
find $OUT -name '*counter_collection.csv' -size +2M -delete
cat $ROOT/gpurun_out/$TAG/lz_pmc_summary.csv
