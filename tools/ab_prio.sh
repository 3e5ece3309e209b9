#!/bin/bash
# diagnostic: wave priority by grid quarter (builds with -DMNW_DIAG_SETPRIO=k in _ab/) against the weighted hand-out
for rnd in 1 2; do
for lib in base prio2 prio3; do for w in 1 0; do echo -n "$lib weighted=$w: "; MNW_LIB_PATH=_ab/$lib.so AB_VALUES=$w timeout -k 10 200 python tools/ab_option.py reg_weighted 20 1e9 2>/dev/null | tail -1; done; done
done
