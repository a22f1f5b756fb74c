# Kernel-trace the default captioning bench and aggregate the dispatches of the timed steps only (tools/trace_window.py).
#   bash tools/scripts/trace_caption_window.sh [STEPS] [OUT]
set -e
STEPS=${1:-20}; OUT=${2:-gpurun_out/trace_window_caption_b1024.txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/trc
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -- python3 bench.py --steps $STEPS --warmup 3 --no-verify --no-cpu-baseline --no-train-mode --no-decode-b1 --no-bf16-mode --no-plain-f32 > gpurun_out/trc_bench.json 2> gpurun_out/trc_bench.err
W=$(python -c "import json;d=json.loads(open('gpurun_out/trc_bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step']*$STEPS)")
python tools/trace_window.py /tmp/trc $W $STEPS > $OUT
head -3 $OUT
