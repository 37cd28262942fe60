#!/bin/bash
# Which limiter holds the clock down? amd-smi's throttle accumulators (MI300+: time the SMU spent limiting for socket power (PPT), socket /
# VR / HBM temperature, PROCHOT) read before and after a load, plus its violation-status monitor while the load runs:
#   tools/limiter_probe.sh mfma              a launch of nothing but bf16 MFMAs (tools/ubench_mfma_clock: zero and random operands)
#   tools/limiter_probe.sh <harness bench args...>   a loop of one harness bench line (e.g. 8 16 4096 4096 64 1 0 0 20000)
R=${GRAFT_REPO_ROOT:-/root/repo}; export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
SMI=/opt/rocm/bin/amd-smi
acc() { $SMI metric -g 0 -v 2>&1 | grep -vE "^ *$" | tr -s ' ' | tr '\n' ';'; echo; }   # (the per-XCD values follow their labels on lines of their own)
echo "-- accumulators before:"; acc
if [ "$1" = "mfma" ]; then $R/tools/ubench_mfma_clock > /tmp/limiter_load.log 2>&1 & else $R/tools/fasn_harness bench "$@" > /tmp/limiter_load.log 2>&1 & fi
LP=$!
for i in $(seq 1 120); do
  w=$(/opt/rocm/bin/rocm-smi -d 0 --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9]+" | grep -oE "[0-9]+$")
  [ "${w:-0}" -gt 500 ] && break
  kill -0 $LP 2>/dev/null || break
  sleep 0.25
done
for i in 1 2 3 4; do
  $SMI monitor -g 0 -p -u -V 2>&1 | tail -2 | tr -s ' '
  /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -E 's/.*sclk clock level: [^(]*\(([0-9]+)Mhz\).*/sclk \1 MHz/; s/.*Power \(W\): ([0-9.]+)/\1 W/' | tr '\n' ' '; echo
  kill -0 $LP 2>/dev/null || break
  sleep 0.5
done
wait $LP
echo "-- accumulators after:"; acc
tail -6 /tmp/limiter_load.log
