# Evidence for the tiled MBConv kernels: issue cost of fp32 vector instructions, dynamic instruction counts per wave, the five
# layers standalone in each product mode
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/collect3; mkdir -p $O
timeout -k 10 60 tools/micro/valu_rate > $O/valu_rate.txt 2>&1
ICAP_MB_EARLY=1 timeout -k 10 200 python tools/mbconv_bench.py 1024 > $O/mbconv_bench.txt 2>&1
rm -rf /tmp/pmi
ICAP_MB_EARLY=1 timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d /tmp/pmi -- python3 tools/mbconv_bench.py 1024 > $O/pmi.txt 2>&1
cp $(find /tmp/pmi -name "*counter_collection.csv" | head -1) $O/pmi.csv
echo collectedC
