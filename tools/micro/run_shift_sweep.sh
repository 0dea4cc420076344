#!/bin/bash
for rep in 1 2 3; do echo "== process $rep"; tools/micro/bin/pp4bench --shift-sweep || exit 1; done
