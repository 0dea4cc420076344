#!/bin/bash
# round-2 GPU check A: new tests (sharded HIP branch, full-size configs, slices, chirps, transfers) then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_transfers.py -x -q -m gpu > gpurun_out/r02a/new_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r02a/new_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_sharded.py --deselect tests/test_gpu_transfers.py > gpurun_out/r02a/suite.log 2>&1
rc=$?
tail -8 gpurun_out/r02a/suite.log
exit $rc
