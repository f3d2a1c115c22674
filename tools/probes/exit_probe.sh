#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}/tools/probes
for cfg in "0 0 0 0" "64 0 0 0" "2048 0 0 0" "2048 0 0 1" "64 1024 0 0" "64 1024 0 1" "64 0 64 0" "64 0 64 1" "2048 1024 64 0"; do
  s=$(date +%s.%N)
  out=$(timeout -k 5 30 ./exit_probe $cfg) || { echo "failed $cfg"; exit 1; }
  e=$(date +%s.%N)
  python3 -c "
import sys
s,e=float('$s'),float('$e'); o='$out'.split(); ll=float(o[-1])
print('dev_MiB pinned_MiB streams free = $cfg : wall %.3f  %s  after_last_line %.3f' % (e-s, ' '.join(o[:4]), e-ll))"
done
