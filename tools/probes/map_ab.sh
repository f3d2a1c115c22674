#!/bin/bash
# GPU box: tools/probes/map_ab.py over every library under tools/probes/ab/ (built here by tools/probes/map_ab_build.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for cfg in c3 c4; do
  for so in tools/probes/ab/*.so; do
    timeout -k 10 200 python3 -u tools/probes/map_ab.py "$so" $cfg || exit 1
  done
done
