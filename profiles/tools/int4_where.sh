#!/bin/bash
# Where do k_attend_int4_wg's microseconds go?  timing-only builds under rocprofv3 --kernel-trace (main kernel and merge kernel separately).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
build() { make -s -C cxl-speckv_amd/csrc OUT=/tmp/abl_$1 EXTRA="$2" -j8 > /dev/null 2>&1 || echo "$1: build failed"; }
build shipped ""
build nostore "-DSPECKV_ABL_NO_STORE"
build store_nt "-DSPECKV_ABL_STORE_NT"
build store_lanemajor "-DSPECKV_ABL_STORE_LANEMAJOR"
for v in shipped nostore store_nt store_lanemajor shipped; do
  export SPECKV_LIB_PATH=/tmp/abl_$v/libcxlspeckv.so
  python profiles/tools/int4_bench.py 32768 80 2>/dev/null | grep "^int4" | sed "s/^/$v: /"
  (cd /tmp && rm -rf /tmp/prof_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/profiles/tools/int4_bench.py 32768 80 > /tmp/prof_$v.log 2>&1)
  for f in $(find /tmp/prof_$v -name "*kernel_stats.csv"); do grep -E "k_attend_int4_wg|k_attend_combine" $f | sed "s/^/$v: /" | cut -c1-150; done
done
