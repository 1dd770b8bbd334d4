#!/bin/bash
# Recompile ONE translation unit of libfqsx.so from the working tree and relink (the other objects are reused as they are:
# only valid when the edit does not touch what those units include: a change of fqsx_layout.h / fqsx_dev.h needs the full build).  usage: tools/rebuild_unit.sh fqsx_api.hip [out.so]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
U=${1:-fqsx_api.hip}; OUT=${2:-$R/fqsqueezer_amd/libfqsx.so}
B=$R/build/libfqsx
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -c $R/fqsqueezer_amd/csrc/$U -o $B/${U%.*}.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT.tmp $B/fqsx_api.o $B/fqsx_k_se.o $B/fqsx_k_pe.o $B/fqsx_k_dec.o $B/fqsx_host.o
mv $OUT.tmp $OUT
