#!/bin/bash
# samples GPU clock and power while the command given as arguments runs
out=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' ; echo; sleep 0.5; done ) > $out 2>&1 &
SAMPLER=$!
"$@"
rc=$?
kill $SAMPLER
exit $rc
