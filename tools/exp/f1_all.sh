export TMPDIR=/tmp
mkdir -p gpurun_out/f1t
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_f1 or device_roc or full_size or two_ranks or export or abi" ) > gpurun_out/f1t/pytest.log 2>&1; tail -15 gpurun_out/f1t/pytest.log
bash tools/exp/f1_time.sh
