set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/collect3; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gputest.log 2>&1
tail -2 $O/gputest.log
timeout -k 10 280 python bench.py > $O/bench_caption_b1024.json 2> $O/bench_caption_b1024.err
echo collectedA
