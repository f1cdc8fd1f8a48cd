#!/bin/bash
# A/B build of the window size of the partial rounds (pmx_mfma.hpp: PMX_MFMA_WINDOW): the two alpha = 5 hybrid TUs (1: t <= 6, 3: t >= 7)
# and pmx_api.o (the host derives the window tables) with -DPMX_MFMA_WINDOW=K [+ extra flags], everything else from the tree's build.  For benches of
# alpha = 5 configs only (the generic-exponent TU keeps the tree's K).   usage: tools/ab/build_window_variant.sh K ["extra flags"]
set -e
K=$1; EXTRA=${2:-}; NAME=${3:-win$K}
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/sponge_amd/csrc
B=$C/build_$NAME
mkdir -p $B
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter --offload-arch=gfx950 -DPMX_MFMA_WINDOW=$K $EXTRA"
$HIPCC $FLAGS -mllvm -opt-disable=reassociate -DPMX_TU=1 -c $C/pmx_device.hip -o $B/pmx_device_1.o &
$HIPCC $FLAGS -mllvm -opt-disable=reassociate -DPMX_TU=3 -c $C/pmx_device.hip -o $B/pmx_device_3.o &
TU0=$C/build/pmx_device.o
if [ "${WITH_TU0:-0}" = "1" ]; then   # the public launchers as well (routing experiments)
  $HIPCC $FLAGS -mllvm -opt-disable=reassociate -DPMX_TU=0 -c $C/pmx_device.hip -o $B/pmx_device_0.o &
  TU0=$B/pmx_device_0.o
fi
$HIPCC $FLAGS -x hip -c $C/pmx_api.cpp -o $B/pmx_api.o &
wait
$HIPCC -shared -fPIC --offload-arch=gfx950 $TU0 $B/pmx_device_1.o $B/pmx_device_3.o $C/build/pmx_device_hybg.o $C/build/pmx_device_hybgw.o $B/pmx_api.o $C/build/pmx_mgpu.o $C/build/pmx_params.o -ldl -o $R/tools/ab/libposeidon_$NAME.so
ls -la $R/tools/ab/libposeidon_$NAME.so
