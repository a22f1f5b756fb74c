# dense encoders in bf16 mode at B = 256: throughput table, per-layer configuration table, rocprofv3 kernel stats
set -e
O=$PWD/gpurun_out/enc3; mkdir -p $O
timeout -k 10 300 python -u tools/encoder_bench.py --dtype bf16 --which vgg16 resnet18 --top 16 > $O/encoder_bench_bf16.txt 2>&1
timeout -k 10 300 python -u tools/conv16_wide_bench.py > $O/conv16_wide.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/encp -- python3 $GRAFT_REPO_ROOT/tools/encoder_bench.py --dtype bf16 --which vgg16 resnet18 > $O/prof.log 2>&1
cp $(find /tmp/encp -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
