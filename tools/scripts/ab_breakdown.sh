# Same-box A/B of the whole captioning step: bench.py --breakdown with the library in the tree, with tools/ab/libicap_hip_old.so
# (a build of the revision to compare against: check it out, make, copy the .so there), and with the tree's again
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
F="--breakdown --no-cpu-baseline --no-train-mode --no-decode-b1 --no-bf16-mode --no-plain-f32"
timeout -k 10 200 python bench.py $F > $O/bd_new.txt 2>&1
cp image_captioning_amd/libicap_hip.so /tmp/new.so
cp tools/ab/libicap_hip_old.so image_captioning_amd/libicap_hip.so
timeout -k 10 200 python bench.py $F > $O/bd_old.txt 2>&1
cp /tmp/new.so image_captioning_amd/libicap_hip.so
timeout -k 10 200 python bench.py $F > $O/bd_new2.txt 2>&1
echo done
