#!/bin/bash
# One profiling pass on the MI355X box: tools/profile_pass.sh <tag>   (e.g. r02_a) -> gpurun_out/<tag>_* (copy into profiles/).
# Trace and counter runs are separate rocprofv3 invocations (never --pmc together with a sys/hip trace), the profiled
# program is python3 itself, each step is joined with && so nothing runs after a failed or timed-out one.
set -o pipefail
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R" || exit 1
mkdir -p gpurun_out profiles
B="bench.py --no-cpu-baseline --no-torch-baseline --no-roofline"
S=25   # 20 timed + 5 warm-up steps in the trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pp_trace -o t -- python3 $B --steps 20 --warmup 5 > gpurun_out/${tag}_trace.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pp_f -o t -- python3 $B --steps 3 --warmup 1 > gpurun_out/${tag}_pmcf.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pp_w -o t -- python3 $B --steps 3 --warmup 1 > gpurun_out/${tag}_pmcw.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc MfmaUtil --kernel-trace -d /tmp/pp_m -o t -- python3 $B --steps 3 --warmup 1 > gpurun_out/${tag}_pmcm.log 2>&1 &&
timeout -k 10 400 python3 bench.py --steps 60 --warmup 10 > gpurun_out/${tag}_bench.log 2> gpurun_out/${tag}_bench.err || { echo "profile pass failed"; tail -5 gpurun_out/${tag}_*.log; exit 1; }
db() { find "$1" -name "*.db" | head -1; }
python3 tools/export_profiles.py $tag "$(db /tmp/pp_trace)" "$(db /tmp/pp_f)" "$(db /tmp/pp_w)" gpurun_out/${tag}_bench.log > gpurun_out/${tag}_summary.txt 2>&1
python3 tools/queue_census.py "$(db /tmp/pp_trace)" $S > profiles/${tag}_queue_census.txt 2>&1
python3 tools/pmc_summary.py "$(db /tmp/pp_m)" profiles/${tag}_pmc_mfma_util.csv > /dev/null 2>&1
cp profiles/${tag}_* gpurun_out/
cat gpurun_out/${tag}_summary.txt
tail -1 gpurun_out/${tag}_bench.log
