"""cProfile of the host side of one training step at a small batch (where the step is bound by launch overhead).
usage: python tools/host_profile.py [--batch 12] [--steps 10]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, step = bench.build_train_step(torch, dev, 0, args.batch)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"batch {args.batch}: host enqueue {t_enq / args.steps * 1e3:.2f} ms/step, with sync {t_all / args.steps * 1e3:.2f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(args.top)


if __name__ == "__main__":
    main()
