#!/bin/bash
# the forward column pattern's loads (rows 2 MiB apart) from hipMalloc memory and from hipMemCreate + hipMemMap mappings at
# 2-MiB / 1-GiB / 4-GiB aligned virtual addresses: does a better aligned mapping (larger page-table fragments) remove the page term?
for mode in 0 21 30 32; do
  echo "== vmm $mode"
  tools/micro/bin/pp4bench --widths --vmm $mode | grep -v "^fdq x2\|16-byte\|stores only\|KiB apart\|1 MiB apart" || exit 1
done
