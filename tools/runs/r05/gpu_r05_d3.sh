#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/dbg/lic_folds.py 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -30
