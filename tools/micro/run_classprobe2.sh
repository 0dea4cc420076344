#!/bin/bash
# classprobe2 on one box: [N] = N plain 2-GiB blocks in a row (two fresh processes), no argument = the candidate kinds (three processes)
if [ -n "${1:-}" ]; then for rep in 1 2; do echo "== process $rep"; tools/micro/bin/classprobe2 "$1" || exit 1; done
else for rep in 1 2 3; do echo "== process $rep"; tools/micro/bin/classprobe2 || exit 1; done; fi
