#!/bin/bash
# Timing-only ablation builds of k_attend_int4_wg (results are garbage): which part of the loop holds the kernel at 0.67?
#   bash profiles/tools/int4_ablate.sh            (on the GPU box; builds into /tmp, runs int4_bench.py 32768 80 on each)
cd $GRAFT_REPO_ROOT
run() {  # name, EXTRA flags
  out=/tmp/abl_$1
  make -s -C cxl-speckv_amd/csrc OUT=$out EXTRA="$2" -j8 > /dev/null 2>&1 || { echo "$1: build failed"; return; }
  for i in 1 2; do
    SPECKV_LIB_PATH=$out/libcxlspeckv.so python profiles/tools/int4_bench.py 32768 80 2>/dev/null | grep "^int4" | sed "s/^/$1: /"
  done
}
run shipped ""
run no_qk_no_pv "-DSPECKV_ABL_NO_QK -DSPECKV_ABL_NO_PV"
run no_arith_no_ldsread "-DSPECKV_ABL_NO_QK -DSPECKV_ABL_NO_PV -DSPECKV_ABL_NO_LDSREAD"
run no_arith_no_ldsread_no_barrier "-DSPECKV_ABL_NO_QK -DSPECKV_ABL_NO_PV -DSPECKV_ABL_NO_LDSREAD -DSPECKV_ABL_NO_BARRIER"
run no_barrier_only "-DSPECKV_ABL_NO_BARRIER"
run no_arith_4waves "-DSPECKV_ABL_NO_QK -DSPECKV_ABL_NO_PV -DSPECKV_INT4_WG_WAVES=4"
