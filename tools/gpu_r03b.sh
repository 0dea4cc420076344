#!/bin/bash
# round 3, second GPU call: the sharded path (world 2 on one GPU, RCCL at world 1), ABI, incoherent; torchrun bench line
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_sharded.py tests/test_abi.py tests/test_incoherent_pol.py \
   --deselect tests/test_gpu_sharded.py::test_config3_full_size_stream > $O/tests.log 2>&1
rc=$?; tail -25 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
   bench.py --gpus 1 --gather all --no-cpu > $O/bench_torchrun_nccl_world1.json 2> $O/bench_torchrun_nccl_world1.err
rc=$?; cat $O/bench_torchrun_nccl_world1.json; tail -12 $O/bench_torchrun_nccl_world1.err
exit $rc
