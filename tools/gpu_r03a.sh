#!/bin/bash
# round 3, first GPU call: the upload-once stream (new tests, config 3 at full size) + its numbers
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_transfers.py tests/test_readers.py \
   "tests/test_gpu_parity.py::test_stream_uploads_every_row_once" "tests/test_gpu_parity.py::test_stream_overlap_save" \
   tests/test_concatenate.py "tests/test_gpu_sharded.py::test_config3_full_size_stream" > $O/tests.log 2>&1
rc=$?; tail -15 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/bench_configs.py 4 4full > $O/configs3.jsonl 2> $O/configs3.err
rc=$?; cat $O/configs3.jsonl; tail -3 $O/configs3.err
exit $rc
