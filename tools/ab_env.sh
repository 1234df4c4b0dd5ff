#!/bin/bash
# A/B of an environment switch on ONE box, interleaved: tools/ab_env.sh <VAR=value> [rounds] [bench args...]
sw=$1; rounds=${2:-5}; shift; shift
ARGS=${@:-"--steps 200 --warmup 10 --no-cpu-baseline --no-pile --plain"}
for r in $(seq 1 $rounds); do
  a=$(python3 bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  b=$(env $sw python3 bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  echo "round $r: default $a   $sw $b"
done
