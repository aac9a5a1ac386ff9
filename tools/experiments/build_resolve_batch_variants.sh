#!/bin/bash
# Full-library builds with a different PF_RESOLVE_BATCH (rows of BatchNorm partials a consumer's resolve prologue keeps in
# flight per thread: 20 = one round trip for the tower layers, but 80 VGPRs -- the prologue then sets the kernel's
# occupancy): tools/experiments/libpointflow_RB<N>.so, selected with PF_LIB_PATH.
set -e
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -Wno-pass-failed -Iinclude -Ipointmvsnet_amd/csrc"
for n in "$@"; do
  d=/tmp/pf_rb$n; mkdir -p $d
  for s in pointmvsnet_amd/csrc/*.hip; do
    ( /opt/rocm/bin/hipcc $FLAGS -DPF_RESOLVE_BATCH=$n -c $s -o $d/$(basename $s .hip).o ) &
    while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libpointflow_RB$n.so $d/*.o
done
ls -la tools/experiments/*.so
