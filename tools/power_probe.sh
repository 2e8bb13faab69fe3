#!/bin/bash
# Samples rocm-smi power / clocks / temperature while a command runs:  tools/power_probe.sh out.txt <command...>
out=$1; shift
( while true; do /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)|fclk" | tr '\n' ';' ; echo; sleep 0.2; done ) > $out.smi &
smi=$!
"$@" > $out 2>&1
kill $smi
