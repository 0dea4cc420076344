#!/bin/bash
# allocation mode study: the same patterns in fresh processes, hipMalloc vs VMM mappings at 2 MiB / 1 GiB aligned addresses
for rep in 1 2 3; do
  for mode in 0 21 30; do
    echo "== rep $rep vmm $mode"
    tools/micro/bin/pp4bench --quick --vmm $mode || exit 1
  done
done
