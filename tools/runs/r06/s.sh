#!/bin/bash
# round 6, run s: the exact-16x16 search in its own 96-register kernel - alone time, PMC,
# parity subset, then the frame pass at the three sizes
R=${GRAFT_REPO_ROOT:-.}
cd $R
TAG=${TAG:-s} bash tools/runs/r06/o.sh
TAG=${TAG:-s} bash tools/runs/r06/f.sh
