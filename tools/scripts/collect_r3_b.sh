# Profiles of the current tree: rocprofv3 kernel stats of the default bench command (captioning) and of the training
# bench, the timed-window summary, and the SQ / TCC counter passes (one pass per counter group).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/collect3; mkdir -p $O
CAP="--no-cpu-baseline --no-train-mode --no-decode-b1 --no-bf16-mode --no-plain-f32"
rm -rf /tmp/ks1 /tmp/ks3 /tmp/pm1 /tmp/pm2 /tmp/pm3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks3 -- python3 bench.py $CAP > $O/ks3.json 2> $O/ks3.err
cp $(find /tmp/ks3 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench_caption_b1024.csv
echo ks3 done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- python3 bench.py --mode train --no-cpu-baseline > $O/ks1.json 2> $O/ks1.err
cp $(find /tmp/ks1 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_bench_train_b256.csv
echo ks1 done
bash tools/scripts/trace_caption_window.sh 20 $O/trace_window_caption_b1024.txt
echo window done
PM="--no-cpu-baseline --no-bf16-mode --no-train-mode --no-decode-b1 --no-verify --no-plain-f32 --steps 3 --warmup 1"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pm1 -- python3 bench.py $PM > $O/pm1.json 2> $O/pm1.err
cp $(find /tmp/pm1 -name "*counter_collection.csv" | head -1) $O/pmc_sq.csv
echo pm1 done
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pm2 -- python3 bench.py $PM > $O/pm2.json 2> $O/pm2.err
cp $(find /tmp/pm2 -name "*counter_collection.csv" | head -1) $O/pmc_fetch.csv
echo pm2 done
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pm3 -- python3 bench.py $PM > $O/pm3.json 2> $O/pm3.err
cp $(find /tmp/pm3 -name "*counter_collection.csv" | head -1) $O/pmc_write.csv
echo collectedB
