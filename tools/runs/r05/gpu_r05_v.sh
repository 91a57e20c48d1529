#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/dbg/lic_walk.py 2>&1 | grep -v amdgpu.ids | tail -40
