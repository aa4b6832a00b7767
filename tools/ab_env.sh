#!/bin/bash
# A/B of an environment switch of the library on ONE box (the sweep kernel alone, tools/rank_scaling.py): tools/ab_env.sh VAR A B
for rep in 1 2; do
for v in "$2" "$3"; do
  export $1=$v
  python tools/rank_scaling.py 2>/dev/null | sed -n 4,5p | sed "s/^/$1=$v /"
  python tools/rank_scaling.py 123182 200 DistMult 2>/dev/null | sed -n 2,3p | sed "s/^/$1=$v /"
  python tools/rank_scaling.py 40943 400 ComplEx 2>/dev/null | sed -n 3,3p | sed "s/^/$1=$v /"
done; done
