#!/bin/bash
# Builds tools/build/bench_select_* (cross-compiles without a GPU); run them on the GPU box.
set -e
cd "$(dirname "$0")"
for nt in 256 512 1024; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBENCH_NT=$nt bench_select.hip -o build/bench_select_$nt
done
for k in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DBENCH_NT=256 -DCI_SEL_STOP_AFTER=$k bench_select.hip -o build/bench_select_256_stop$k
done
