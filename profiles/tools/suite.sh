#!/bin/bash
cd $GRAFT_REPO_ROOT
n=${1:-2}
for i in $(seq 1 $n); do timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
