#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
