#!/bin/bash
export STRATS=auto
for c in 2048 8192 32768 131072; do echo "== chunk $c"; PLHIP_HOT_CHUNK=$c timeout -k 10 200 python scripts/skew_small.py 1e7 0.3 0 2>/dev/null | grep "hint  100000" || exit 1; done
for c in 2048 8192 32768; do echo "== chunk $c"; PLHIP_HOT_CHUNK=$c timeout -k 10 200 python scripts/skew_small.py 1e7 0 0.05 2>/dev/null | grep "hint  100000" || exit 1; done
echo "== no nulls but nullable key"; timeout -k 10 200 python scripts/skew_small.py 1e7 0 0.0000001 2>/dev/null | grep "hint  100000"
