set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trk -- python3 bench.py --mode train --steps 20 --warmup 3 --no-verify --no-cpu-baseline > gpurun_out/trk_bench.json 2> gpurun_out/trk_bench.err
W=$(python -c 'import json;d=json.loads(open("gpurun_out/trk_bench.json").read().strip().splitlines()[-1]);print(d["ms_per_step"]*20)')
python tools/trace_window.py /tmp/trk $W 20 > gpurun_out/trace_window_train_b256.txt
head -3 gpurun_out/trace_window_train_b256.txt
