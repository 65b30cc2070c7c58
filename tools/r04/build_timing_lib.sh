#!/bin/bash
# libjxl_hip.so with the DC-group timing prints (-DJXLHIP_DC_TIMING): entropy.cc recompiled, the rest from the build dir
# usage: tools/r04/build_timing_lib.sh out.so
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R/libjxl_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DJXLHIP_DC_TIMING -c entropy.cc -o /tmp/entropy_timing.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$1" $(ls build/*.o | grep -v "entropy.o\|runner.o") /tmp/entropy_timing.o
