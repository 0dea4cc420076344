#!/bin/bash
for rep in 1 2; do echo "== process $rep"; tools/micro/bin/bufprobe 4 arena || exit 1; done
