# Dynamic instruction counts of the fused MBConv layers (tools/mbconv_bench.py) per kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O; rm -rf /tmp/pmi
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d /tmp/pmi -- python3 tools/mbconv_bench.py 1024 > $O/pmi.txt 2>&1
cp $(find /tmp/pmi -name "*counter_collection.csv" | head -1) $O/pmi.csv
echo done
