#!/bin/bash
# Where do k_attend_int4_wg's microseconds go?  (bash profiles/tools/int4_where.sh on the GPU box; profiles/r03_int4_ablation.txt, third series)
# The read probe of the kernel's address pattern, then the shipped kernel and timing-only builds under rocprofv3 --kernel-trace
# (main kernel and merge kernel separately): "bare" = no arithmetic, no LDS reads; "plain_nt" = the same bytes by register loads;
# "nostore" = the shipped kernel without its partial stores.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd profiles/probes && hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libint4shape.so int4shape.hip) && python profiles/probes/int4shape.py 2>&1 | grep -E "^shipped"
build() { make -s -C cxl-speckv_amd/csrc OUT=/tmp/abl_$1 EXTRA="$2" -j8 > /dev/null 2>&1 || echo "$1: build failed"; }
build shipped ""
build bare "-DSPECKV_ABL_NO_QK -DSPECKV_ABL_NO_PV -DSPECKV_ABL_NO_LDSREAD"
build plain_nt "-DSPECKV_ABL_PLAIN_LOADS -DSPECKV_INT4_NT_LOADS"
build nostore "-DSPECKV_ABL_NO_STORE"
for v in shipped bare plain_nt nostore; do
  export SPECKV_LIB_PATH=/tmp/abl_$v/libcxlspeckv.so
  python profiles/tools/int4_bench.py 32768 80 2>/dev/null | grep "^int4" | sed "s/^/$v: /"
  (cd /tmp && rm -rf /tmp/prof_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/profiles/tools/int4_bench.py 32768 80 > /tmp/prof_$v.log 2>&1)
  for f in $(find /tmp/prof_$v -name "*kernel_stats.csv"); do grep -E "k_attend_int4_wg|k_attend_combine" $f | sed "s/^/$v: /" | cut -c1-150; done
done
