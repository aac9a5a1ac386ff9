#!/bin/bash
# Ablation builds of the tower kernels (csrc/conv2d_wide.hip's PF_DBG_NOLOAD / PF_DBG_NOMFMA / PF_DBG_NOSTORE switches):
# one library per switch under tools/experiments/ (git-ignored, travels with gpurun), selected with PF_LIB_PATH.
set -e
cd "$(dirname "$0")/../.."
B=pointmvsnet_amd/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -Wno-pass-failed -Iinclude -Ipointmvsnet_amd/csrc"
for v in NOLOAD NOMFMA NOSTORE; do
  /opt/rocm/bin/hipcc $FLAGS -DPF_DBG_$v -c pointmvsnet_amd/csrc/conv2d_wide.hip -o /tmp/conv2d_wide_$v.o
  objs=$(ls $B/*.o | grep -v conv2d_wide.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libpointflow_$v.so $objs /tmp/conv2d_wide_$v.o
done
ls -la tools/experiments/*.so
