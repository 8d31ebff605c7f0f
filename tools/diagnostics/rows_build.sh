#!/bin/bash
# builds a diagnostic variant of the library with extra flags for conv_rows.hip:
#   tools/diagnostics/rows_build.sh <tag> <flags...>     -> tools/diagnostics/libs/lib_rows_<tag>.so   (P4C_LIB_PATH=<that file>)
set -e
mkdir -p /root/repo/tools/diagnostics/libs
tag=$1; shift
cd /root/repo/py4cast_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -simplifycfg-sink-common=false -fno-slp-vectorize "$@" -c conv_rows.hip -o /tmp/conv_rows_$tag.o
objs=$(ls obj/*.o | grep -v '/conv_rows.o$' | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/conv_rows_$tag.o -o /root/repo/tools/diagnostics/libs/lib_rows_$tag.so
