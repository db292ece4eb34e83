# per-phase instruction counts of the embed kernel: one rocprofv3 --pmc pass per ablation mask
export TMPDIR=/tmp
rm -rf gpurun_out/pmca; mkdir -p gpurun_out/pmca
for m in 256 257 258 260 264 271; do
  timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmca -o m$m -- python tools/run_embed.py kitti00 2 $m > gpurun_out/pmca/m$m.log 2>&1 </dev/null
done
ls gpurun_out/pmca | head -30
