#!/bin/bash
# usage: tools/mkvariant.sh <name> <file.hip replacement for csrc/X.hip> <X.hip> [extra flags]  -> tools/_variants/<name>.so
name=$1; src=$2; which=$3; shift 3
R=/root/repo/psi-release_amd
objs=""
for o in $R/lib/obj/*.o; do b=$(basename $o .o); case $b in *_fma) continue;; esac; [ "$b.hip" != "$which" ] && objs="$objs $o"; done
extra=""
[ "$which" = "nnindex.hip" ] && extra="-ffp-contract=off"
[ "$which" = "chamfer.hip" ] && extra="-ffp-contract=off -fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -I$R/csrc $extra "$@" -c $src -o /tmp/var_$name.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/tools/_variants/$name.so $objs /tmp/var_$name.o && echo built tools/_variants/$name.so
