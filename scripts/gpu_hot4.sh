#!/bin/bash
export STRATS=auto
PLHIP_T2_VERBOSE=1 timeout -k 10 200 python scripts/skew_profile.py 1e9 1e8 2>&1 | tail -2 && \
PLHIP_T2_VERBOSE=1 timeout -k 10 200 python scripts/skew_profile.py 3e8 3e7 2>&1 | tail -2 && \
PLHIP_T2_VERBOSE=1 SKEW_FRAC=0 timeout -k 10 200 python scripts/skew_profile.py 3e8 3e7 2>&1 | tail -2 && \
PLHIP_T2_SWITCH=9 PLHIP_T2_VERBOSE=1 SKEW_FRAC=0 timeout -k 10 200 python scripts/skew_profile.py 3e8 3e7 2>&1 | tail -2
