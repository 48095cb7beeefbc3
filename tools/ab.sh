#!/bin/bash
# A/B of library variants in ONE call (same box): tools/ab.sh "<variants>" <layers_isolated row substrings...>
R=$GRAFT_REPO_ROOT; cd $R
VARS=$1; shift
for rep in 1 2; do for v in $VARS; do
  echo "== $v (rep $rep)"
  REPO_HIP_LIB=$R/repo_amd/variants/lib_$v.so timeout 600 python3 tools/layers_isolated.py "$@" 2>&1 | grep -v "^#\|amdgpu.ids" | awk '{printf "%s ", $0; print ""}' | sed 's/  */ /g' | cut -c1-90
done; done
