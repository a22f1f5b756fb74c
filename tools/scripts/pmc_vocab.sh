# Counter passes over the vocabulary projection alone (tools/x3_vocab_pmc.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/vocab_pmc; mkdir -p $O
timeout -k 10 120 python3 tools/x3_vocab_pmc.py 3 > $O/plain.txt 2>&1
rm -rf /tmp/v1 /tmp/v2 /tmp/v3
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/v1 -- python3 tools/x3_vocab_pmc.py > $O/v1.txt 2>&1
cp $(find /tmp/v1 -name "*counter_collection.csv" | head -1) $O/v1.csv
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d /tmp/v2 -- python3 tools/x3_vocab_pmc.py > $O/v2.txt 2>&1
cp $(find /tmp/v2 -name "*counter_collection.csv" | head -1) $O/v2.csv
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d /tmp/v3 -- python3 tools/x3_vocab_pmc.py > $O/v3.txt 2>&1
cp $(find /tmp/v3 -name "*counter_collection.csv" | head -1) $O/v3.csv
echo collectedV
