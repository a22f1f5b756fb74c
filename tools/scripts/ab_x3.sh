# Same-box A/B of the decode-step GEMMs: the tree's library, tools/ab/libicap_hip_old.so, the tree's again
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O; rm -f $O/x3.txt
timeout -k 10 100 python3 tools/x3_vocab_time.py >> $O/x3.txt 2>&1
cp image_captioning_amd/libicap_hip.so /tmp/new.so
cp tools/ab/libicap_hip_old.so image_captioning_amd/libicap_hip.so
timeout -k 10 100 python3 tools/x3_vocab_time.py >> $O/x3.txt 2>&1
cp /tmp/new.so image_captioning_amd/libicap_hip.so
timeout -k 10 100 python3 tools/x3_vocab_time.py >> $O/x3.txt 2>&1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "x3 or split or linear or bf16 or greedy or caption" > $O/x3_tests.log 2>&1
tail -2 $O/x3_tests.log
