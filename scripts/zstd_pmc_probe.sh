# scripts/zstd_pmc_probe.sh [n_frames] -- SQ / TCC counters of zstd_frames_kernel on the realistic pack mix
export TMPDIR=/tmp
ROOT=$(pwd)
N=${1:-6400}
mkdir -p gpurun_out/zpmc
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf gpurun_out/zpmc/$tag
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv --kernel-include-regex zstd_frames -d $ROOT/gpurun_out/zpmc/$tag -o p -- python $ROOT/scripts/zstd_gpu_probe.py $N real > $ROOT/gpurun_out/zpmc/$tag.log 2>&1)
  grep "run 1" gpurun_out/zpmc/$tag.log
done
python scripts/pmc_summary.py gpurun_out/zpmc/summary.csv gpurun_out/zpmc/*/
cat gpurun_out/zpmc/summary.csv
