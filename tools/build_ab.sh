#!/bin/bash
# A/B builds of the library for bench.py (SMX_BENCH_LIB=tools/ab/lib_<name>.so): tools/build_ab.sh name1=<hipcc flags> name2=<flags> ... (4 at a time)
cd /root/repo/spades_amd/csrc || exit 1
n=0
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread $flags -o /root/repo/tools/ab/lib_$name.so smx_api.hip 2>&1 | grep -i "error"; echo "$name built" ) &
  n=$((n+1)); [ $((n % 4)) -eq 0 ] && wait
done
wait
