#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in A B; do echo -n "$v: "; SYLOW_HIP_LIB=$PWD/tools/ab/lib_$v.so python3 tools/dbg/time_fp.py 2>&1 | grep "TB/s"; done; done
