# usage: bash tools/exp/ab.sh [variant.so ...]   - kernel stats of bench.py for the default build and each variant
export TMPDIR=/tmp
for v in default "$@"; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$PWD/$v; fi
  rm -rf gpurun_out/kt
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o kt -- python bench.py --steps 40 --warmup 3 --no-cpu-baseline > gpurun_out/kt.log 2>&1 </dev/null
  echo "== $v"; python tools/kstats.py gpurun_out/kt/kt_kernel_stats.csv | head -3
done
