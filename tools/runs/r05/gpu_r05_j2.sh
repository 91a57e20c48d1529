#!/bin/bash
# last check of the round's final build: smoke, the frame pass / CU-state / decoder tests
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -x -q -k "frame_pass or cu_state or stream or rdoq or me_search" 2>&1 | grep -E "passed|failed" | tail -2
