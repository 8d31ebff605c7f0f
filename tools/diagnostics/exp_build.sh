#!/bin/bash
# builds diagnostic variants of the library: tools/diagnostics/libs/lib_exp<N>.so with -DP4C_EXP=N in conv_bf16.hip
set -e
cd /root/repo/py4cast_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -simplifycfg-sink-common=false -fno-slp-vectorize -DP4C_EXP=$n $EXTRA -c conv_bf16.hip -o /root/repo/tools/diagnostics/libs/conv_bf16_exp$n$TAG.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 lib.o rollout.o losses.o conv_f32.o /root/repo/tools/diagnostics/libs/conv_bf16_exp$n$TAG.o norm_pool.o halfunet.o -o /root/repo/tools/diagnostics/libs/lib_exp$n$TAG.so
  rm /root/repo/tools/diagnostics/libs/conv_bf16_exp$n$TAG.o
done
