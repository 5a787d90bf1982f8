#!/bin/bash
# tail split sweep: forward convs at 8 crops (one student pass) under different minimum piece lengths / overhead constants
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for shape in l3_1x1b l3_3x3 l4_3x3; do
  for ms in 8 12 16 21 32; do
    echo -n "$shape min_steps=$ms ovh=6: "; DASAC_TAIL_MINSTEPS=$ms python tools/one_conv.py $shape fwd 30 8 2>/dev/null | grep -o "us_per_call [0-9.]*"
  done
  for ov in 12 20; do
    echo -n "$shape min_steps=8 ovh=$ov: "; DASAC_TAIL_OVH=$ov python tools/one_conv.py $shape fwd 30 8 2>/dev/null | grep -o "us_per_call [0-9.]*"
  done
  echo -n "$shape B=16 default: "; python tools/one_conv.py $shape fwd 30 16 2>/dev/null | grep -o "us_per_call [0-9.]*"
  for ms in 16 32; do echo -n "$shape B=16 min_steps=$ms: "; DASAC_TAIL_MINSTEPS=$ms python tools/one_conv.py $shape fwd 30 16 2>/dev/null | grep -o "us_per_call [0-9.]*"; done
done
