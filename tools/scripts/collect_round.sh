# End-of-round evidence: full GPU suite, the default bench line, the training-mode lines, breakdown tables, rocprofv3 kernel
# stats and the timed-window summaries.  Everything lands under gpurun_out/ (copy what is kept into profiles/).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/collect; mkdir -p $O
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gputest.log 2>&1
tail -2 $O/gputest.log
timeout -k 10 400 python bench.py > $O/bench_caption_b1024.json 2> $O/bench_caption_b1024.err
timeout -k 10 300 python bench.py --mode train > $O/bench_train_b256.json 2> $O/bench_train_b256.err
timeout -k 10 300 python bench.py --mode train --graph --batch 12 --steps 40 --no-cpu-baseline > $O/bench_train_b12_graph.json 2> $O/bench_train_b12_graph.err
timeout -k 10 300 python3 bench.py --mode train --dtype f32x3 --breakdown > $O/breakdown_train_b256.txt 2>&1
timeout -k 10 300 python3 bench.py --breakdown --no-train-mode --no-decode-b1 > $O/breakdown_caption_b1024.txt 2>&1
rm -rf /tmp/ks1 /tmp/ks2 /tmp/ks3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- python3 bench.py --mode train --no-cpu-baseline > $O/ks1.json 2> $O/ks1.err
cp $(find /tmp/ks1 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench_train_b256.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -- python3 bench.py --mode train --graph --batch 12 --steps 40 --no-cpu-baseline > $O/ks2.json 2> $O/ks2.err
cp $(find /tmp/ks2 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench_train_b12_graph.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks3 -- python3 bench.py --no-cpu-baseline --no-train-mode --no-decode-b1 --no-bf16-mode > $O/ks3.json 2> $O/ks3.err
cp $(find /tmp/ks3 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench_caption_b1024.csv
bash tools/scripts/trace_train_window.sh 20 $O/trace_window_train_b256.txt
bash tools/scripts/trace_train_window.sh 50 $O/trace_window_train_b12_graph.txt --graph --batch 12 --dtype f32
echo collected
