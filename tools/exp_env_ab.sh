#!/bin/bash
# same-box A/B of an environment switch: tools/exp_env_ab.sh VAR=VALUE <python script> [args]
kv=$1; shift
for rep in 1 2; do
  echo "--- default"; python "$@" 2>&1 | grep -E "convert_into|bench layouts|->"
  echo "--- $kv"; env $kv python "$@" 2>&1 | grep -E "convert_into|bench layouts|->"
done
