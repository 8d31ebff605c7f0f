#!/bin/bash
# builds a diagnostic variant of the library with extra flags for gemm.hip:
#   tools/diagnostics/gemm_build.sh <tag> <flags...>     -> tools/diagnostics/libs/lib_gemm_<tag>.so   (P4C_LIB_PATH=<that file>)
# e.g. -DP4C_NT_EXP=32: gemm_nt without the XCD-aware tile order (tile = blockIdx.x, the round-5 order)
set -e
tag=$1; shift
cd /root/repo/py4cast_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c gemm.hip -o /tmp/gemm_$tag.o
objs=$(ls obj/*.o | grep -v '/gemm.o$' | tr '\n' ' ')
mkdir -p /root/repo/tools/diagnostics/libs
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/gemm_$tag.o -o /root/repo/tools/diagnostics/libs/lib_gemm_$tag.so
