#!/bin/bash
# Full-library builds with a different PF_XCD mask (XCD-aware block order per kernel family, csrc/pf_common.h):
# tools/experiments/libpointflow_XCD<mask>.so, selected with PF_LIB_PATH.
set -e
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -Wno-pass-failed -Iinclude -Ipointmvsnet_amd/csrc"
for n in "$@"; do
  d=/tmp/pf_xcd$n; mkdir -p $d
  for s in pointmvsnet_amd/csrc/*.hip; do
    ( /opt/rocm/bin/hipcc $FLAGS -DPF_XCD=$n -c $s -o $d/$(basename $s .hip).o ) &
    while [ $(jobs -r | wc -l) -ge 7 ]; do sleep 0.2; done
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libpointflow_XCD$n.so $d/*.o
done
ls -la tools/experiments/*.so
