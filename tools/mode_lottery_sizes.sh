#!/bin/bash
# the x / y allocation lottery of tools/mode_persist at several grid edges, three fresh processes each
for E in ${@:-512 500}; do for i in 1 2 3; do ./tools/mode_persist $E 4 sweep lottery || exit 1; done; done
