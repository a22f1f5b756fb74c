# Same-box A/B of the fused MBConv layers: GPU tests, then tools/mbconv_bench.py with the library in the tree, with
# tools/ab/libicap_hip_old.so (a build of the revision to compare against: check it out, make, copy the .so there), and with the tree's again
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/test.txt 2>&1 || { tail -30 $O/test.txt; exit 1; }
tail -1 $O/test.txt
ICAP_MB_EARLY=1 timeout -k 10 200 python tools/mbconv_bench.py 1024 > $O/new.txt 2>&1

cp image_captioning_amd/libicap_hip.so /tmp/new.so
cp tools/ab/libicap_hip_old.so image_captioning_amd/libicap_hip.so
ICAP_MB_EARLY=1 timeout -k 10 200 python tools/mbconv_bench.py 1024 > $O/old.txt 2>&1
cp /tmp/new.so image_captioning_amd/libicap_hip.so
ICAP_MB_EARLY=1 timeout -k 10 200 python tools/mbconv_bench.py 1024 > $O/new2.txt 2>&1
echo done
