#!/bin/bash
# Builds a variant of libposeidon_mi355x.so with extra device-compile flags into tools/ab/libposeidon_<name>.so
# (objects under sponge_amd/csrc/build_<name>/; the tree's own library and build/ are left alone).
# usage: tools/ab/build_variant.sh <name> "<extra flags, e.g. -DPMX_HYB_3WAVE_MAX_T=6>" [TUs to rebuild with the flags: "0 1 2 3 4"]
set -e
NAME=$1; EXTRA=$2; TUS=${3:-"0 1 2 3 4"}
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/sponge_amd/csrc
B=$C/build_$NAME
mkdir -p $B
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950"
DEV="-mllvm -opt-disable=reassociate"
pids=()
for tu in 0 1 2 3 4; do
  obj=$B/pmx_device_$tu.o
  if [[ " $TUS " == *" $tu "* ]]; then
    $HIPCC $FLAGS $DEV $EXTRA -DPMX_TU=$tu -c $C/pmx_device.hip -o $obj & pids+=($!)
  else
    case $tu in 0) src=$C/build/pmx_device.o;; 1) src=$C/build/pmx_device_hyb5.o;; 2) src=$C/build/pmx_device_hybg.o;; 3) src=$C/build/pmx_device_hyb5w.o;; 4) src=$C/build/pmx_device_hybgw.o;; esac
    cp $src $obj
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC -shared -fPIC --offload-arch=gfx950 $B/pmx_device_0.o $B/pmx_device_1.o $B/pmx_device_2.o $B/pmx_device_3.o $B/pmx_device_4.o $C/build/pmx_api.o $C/build/pmx_mgpu.o $C/build/pmx_params.o -ldl -o $R/tools/ab/libposeidon_$NAME.so
ls -la $R/tools/ab/libposeidon_$NAME.so
