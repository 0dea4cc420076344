#!/bin/bash
for rep in 1 2 3 4; do echo "== process $rep"; tools/micro/bin/classprobe | grep "x -> work\|spacers\|^candidate  [0-9]" || exit 1; done
