#!/bin/bash
# tree vs variants on several workloads, interleaved
set -u
export ROUNDS=1
for w in "--workload c3 --solid-method 2" "--workload c3 --solid-method 2 --max-seg 24" "--workload c3 --solid-method 2 --max-seg 40" "--workload c3 --solid-method 4" "--workload c3 --solid-method 6" "--workload c3 --solid-method 6 --max-seg 32" "--workload c2" "--workload c4"; do
  echo "== $w"; TAG=ab2 bash scripts/r02_ab.sh $w
done
