#!/bin/bash
# Randomised GPU parity tests with shifted seeds (fresh inputs every round):
#   tools/soak.sh <first> <last> [pytest -k expression]
# (a test's input-coverage assertion - 'some coefficients are non-zero', 'the filter
# changed something' - may trip on an unlucky seed; parity assertions must not)
a=${1:-1}; b=${2:-5}; k=${3:-"not golden"}
for i in $(seq $a $b); do
  XVC_SOAK=$i timeout 900 python -m pytest tests -m gpu -q -k "$k" 2>&1 | grep -E "passed|failed|FAILED" | tr '\n' ' '
  echo " [soak $i]"
done
