#!/bin/bash
# Does the workgroup count's fit to the machine (768 resident workgroups = 3 per CU) matter for k_attend_int4_wg?
#   bash profiles/tools/int4_rounds.sh        (GPU box)   layers x 2 head quads x 8 splits workgroups: 48 -> 768 (1.0 rounds), 72 -> 1.5, 80 -> 1.67, 96 -> 2.0, 120 -> 2.5, 144 -> 3.0
cd $GRAFT_REPO_ROOT
make -s -C cxl-speckv_amd/csrc OUT=/tmp/abl_bare EXTRA="-DSPECKV_ABL_NO_QK -DSPECKV_ABL_NO_PV -DSPECKV_ABL_NO_LDSREAD" -j8 > /dev/null 2>&1 || echo "bare build failed"
for L in 48 72 80 96 120 144; do
  for lib in default /tmp/abl_bare/libcxlspeckv.so; do
    if [ $lib = default ]; then unset SPECKV_LIB_PATH; else export SPECKV_LIB_PATH=$lib; fi
    python profiles/tools/int4_bench.py 32768 $L 2>/dev/null | grep "^int4"
  done
done
