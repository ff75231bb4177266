# second counter set: translation (UTCL1/UTCL2), L1 stalls, L2 latencies -- same probe
export TMPDIR=/tmp
ROOT=$(pwd)
N=${1:-50000}
mkdir -p gpurun_out/zpmc2
i=0
for set in "TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_ACCESSES TCP_TOTAL_CACHE_ACCESSES" "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TCP_LATENCY" "TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_THRASHING_STALL TCP_UTCL1_SERIALIZATION_STALL" "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES" "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_CYCLES"; do
  i=$((i+1))
  rm -rf gpurun_out/zpmc2/s$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv --kernel-include-regex zstd_frames -d $ROOT/gpurun_out/zpmc2/s$i -o p -- python $ROOT/scripts/zstd_gpu_probe.py $N real > $ROOT/gpurun_out/zpmc2/s$i.log 2>&1)
  grep "run 1" gpurun_out/zpmc2/s$i.log | head -1
done
python scripts/pmc_summary.py gpurun_out/zpmc2/summary.csv gpurun_out/zpmc2/*/
cat gpurun_out/zpmc2/summary.csv
