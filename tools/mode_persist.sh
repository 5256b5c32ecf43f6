#!/bin/bash
# N fresh processes of tools/mode_persist (the persistent-grid / pad experiment on the timing modes) on one box; one JSON
# line per process.  Usage: tools/mode_persist.sh [N=10] [grid edge=512]
N=${1:-10}
E=${2:-512}
for i in $(seq 1 $N); do
  ./tools/mode_persist $E 4 $3 $4 $5 $6 $7 || exit 1
done
