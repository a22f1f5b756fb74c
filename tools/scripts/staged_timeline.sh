cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3
for B in 12 64; do
REH_TIMELINE=1 REH_SAME=0 REH_GRAPH=1 REH_STEPS=6 REH_BATCH=$B REH_VOCAB=6756 REH_BUCKET_MB=16 timeout -k 10 250 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2957$((B%10)) tools/train_rehearse.py 2>&1 | grep "^\[r" > gpurun_out/r3/staged_timeline_b$B.txt || exit 1
cat gpurun_out/r3/staged_timeline_b$B.txt
done
