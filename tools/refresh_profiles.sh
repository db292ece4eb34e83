#!/bin/bash
# Regenerate the judged artefacts under profiles/ on the GPU box (run through gpurun; outputs land in gpurun_out/prof):
#   kernel stats of the default bench command, the bench line itself, HBM counters (separate --pmc passes), SQ counters
export TMPDIR=/tmp
O=gpurun_out/prof; rm -rf $O; mkdir -p $O
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err </dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err </dev/null
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- python tools/run_embed.py kitti00 3 > $O/fetch.log 2>&1 </dev/null
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O -o write -- python tools/run_embed.py kitti00 3 > $O/write.log 2>&1 </dev/null
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM --output-format csv -d $O -o sq1 -- python tools/run_embed.py kitti00 3 > $O/sq1.log 2>&1 </dev/null
timeout 100 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -o sq2 -- python tools/run_embed.py kitti00 3 > $O/sq2.log 2>&1 </dev/null
ls $O
