#!/bin/bash
# tools/collect_profiles.sh TAG PREFIX: copy what a tools/prof_round6.sh call left under gpurun_out/TAG into profiles/ under the names profiles/INDEX.md uses
# (PREFIX_<workload>_<pass>_rocprofv3_summary.txt, PREFIX_bench_all_workloads.jsonl, PREFIX_bench_default.json, PREFIX_clocks_and_power_under_load.log, pmc_latest.json)
R=/root/repo; T=$1; P=$2; O=$R/gpurun_out/$T
for d in $O/*_fwd $O/*_bwd; do
  n=$(basename $d); [ -s $d/summary.txt ] && cp $d/summary.txt $R/profiles/${P}_${n}_rocprofv3_summary.txt
done
cp $O/bench_all.jsonl $R/profiles/${P}_bench_all_workloads.jsonl
cp $O/bench_default.json $R/profiles/${P}_bench_default.json
[ -s $O/clocks.log ] && cp $O/clocks.log $R/profiles/${P}_clocks_and_power_under_load.log
cp $O/pmc_latest.json $R/profiles/pmc_latest.json
[ -s $R/gpurun_out/${T}_pytest.log ] && cp $R/gpurun_out/${T}_pytest.log $R/profiles/${P}_pytest_gpu_tail.log
[ -s $R/gpurun_out/${T}_dropout.log ] && cp $R/gpurun_out/${T}_dropout.log $R/profiles/${P}_dropout_cost.log
ls $R/profiles | grep "^${P}_" | wc -l
