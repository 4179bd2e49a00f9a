#!/bin/bash
# A/B builds of ONE translation unit: scratch/ab/<name>/libcxlspeckv.so = the current objects with <file>.o rebuilt under extra
# flags (scratch/ is git-ignored and travels to the GPU box).  usage: profiles/tools/ab_build.sh <name> <file-stem> <flags...>
# then: SPECKV_LIB_PATH=scratch/ab/<name>/libcxlspeckv.so python ...
set -e
name=$1; stem=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/scratch/ab/$name
mkdir -p $out/obj
make -s -C $root/cxl-speckv_amd/csrc -j8 >/dev/null
cp -p $root/cxl-speckv_amd/lib/obj/*.o $out/obj/
rm -f $out/obj/$stem.o
make -s -C $root/cxl-speckv_amd/csrc OUT=$out EXTRA="$*" >/dev/null
ls -la $out/libcxlspeckv.so
