# Kernel-trace `bench.py --mode train` and aggregate the dispatches of the timed steps only (tools/trace_window.py).
#   bash tools/scripts/trace_train_window.sh [STEPS] [OUT] [extra bench.py flags...]
set -e
STEPS=${1:-20}; OUT=${2:-gpurun_out/trace_window_train_b256.txt}; shift || true; shift || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/trk
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trk -- python3 bench.py --mode train --steps $STEPS --warmup 3 --no-verify --no-cpu-baseline "$@" > gpurun_out/trk_bench.json 2> gpurun_out/trk_bench.err
W=$(python -c "import json;d=json.loads(open('gpurun_out/trk_bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step']*$STEPS)")
python tools/trace_window.py /tmp/trk $W $STEPS > $OUT
head -3 $OUT
