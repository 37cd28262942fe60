#!/bin/bash
# samples sclk / socket power (rocm-smi) while a harness bench line loops: tools/clock_probe.sh <harness bench args...>
# (waits until the socket draws more than 500 W - the harness builds its inputs on the host first - then takes six samples)
R=${GRAFT_REPO_ROOT:-/root/repo}; export LD_LIBRARY_PATH=$R/tools:$LD_LIBRARY_PATH
$R/tools/fasn_harness bench "$@" > /tmp/probe_bench.log 2>&1 &
BP=$!
smp() { /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -E 's/.*sclk clock level: [^(]*\(([0-9]+)Mhz\).*/sclk \1 MHz/; s/.*Power \(W\): ([0-9.]+)/\1 W/' | tr '\n' ' '; echo; }
for i in $(seq 1 120); do
  w=$(/opt/rocm/bin/rocm-smi -d 0 --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9]+" | grep -oE "[0-9]+$")
  [ "${w:-0}" -gt 500 ] && break
  kill -0 $BP 2>/dev/null || break
  sleep 0.5
done
for i in 1 2 3 4 5 6; do smp; sleep 0.4; done
wait $BP; tail -1 /tmp/probe_bench.log
