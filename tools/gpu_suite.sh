#!/bin/bash
# whole GPU suite in one process + optional extra commands; logs under gpurun_out/$1
set -o pipefail
TAG=${1:-suite}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/suite.log 2>&1
rc=$?
tail -12 gpurun_out/$TAG/suite.log
exit $rc
