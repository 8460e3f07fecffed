#!/bin/bash
export STRATS=tilesort2
run() { echo "== $*"; env "$@" timeout -k 10 200 python scripts/skew_profile.py $N $IDS 2>&1 | grep -v "^rows" ; }
N=1e9 IDS=1e8
run PLHIP_T2_HOT_CHUNK=16384 || exit 1
run SKEW_FRAC=0 PLHIP_T2_KB=76 PLHIP_T2_VERBOSE=1 || exit 1
run SKEW_FRAC=0 PLHIP_T2_KB=48 PLHIP_T2_VERBOSE=1 || exit 1
N=4e8 IDS=2e7
run SKEW_FRAC=0 PLHIP_T2_KB=32 PLHIP_T2_VERBOSE=1 || exit 1
run SKEW_FRAC=0 PLHIP_T2_KB=76 PLHIP_T2_VERBOSE=1 || exit 1
N=4e8 IDS=8e6
run SKEW_FRAC=0 PLHIP_T2_KB=32 PLHIP_T2_VERBOSE=1 || exit 1
run SKEW_FRAC=0 PLHIP_T2_KB=76 PLHIP_T2_VERBOSE=1 || exit 1
