#!/bin/bash
# dev aid: builds dump1090_amd/libmodes_hip_<tag>.so with extra -D flags for kernels_dc.hip (select with MODES_LIB=<path>)
tag=$1; shift
cd "$(dirname "$0")/../dump1090_amd/csrc" || exit 1
make -s || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result --offload-arch=gfx950 -ffp-contract=off "$@" -c kernels_dc.hip -o /tmp/kernels_dc_$tag.o || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result "$@" -c engine.cpp -o /tmp/engine_$tag.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libmodes_hip_$tag.so kernels.o kernels_2000.o kernels_ac.o kernels_8000.o /tmp/kernels_dc_$tag.o noalloc.o strip.o kernels_fields.o /tmp/engine_$tag.o live.o dropin.o accept.o tables.o
