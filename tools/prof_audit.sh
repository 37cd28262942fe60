#!/bin/bash
# tools/prof_audit.sh TAG "<audit_modes.py arguments>": rocprofv3 --kernel-trace --stats of one tools/audit_modes.py call; the per-kernel statistics (name, calls,
# average / min / max duration) of the library's kernels go to gpurun_out/TAG_kernels.txt (kernel names = those of fasn_launch_plan)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$1; A=$2; O=$R/gpurun_out/prof_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/tools/audit_modes.py $A > $O/run.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/audit_modes.py $A"; grep -v Warn $O/run.log | grep "ms_per_step"; echo; python3 $R/tools/pmc_summary.py $O fasn_; } > $R/gpurun_out/${T}_kernels.txt 2>&1
find $O -name "*.db" -delete; find $O -type f -size +2M -delete
cat $R/gpurun_out/${T}_kernels.txt | head -40
