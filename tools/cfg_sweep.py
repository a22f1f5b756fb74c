"""Time ops.linear for (M, K, N) under chosen tile configurations (ICAP_FORCE_CFG), one subprocess each."""
import os, subprocess, sys
M, K, N = sys.argv[1:4]
cfgs = [int(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else list(range(20))
here = os.path.dirname(os.path.abspath(__file__))
for c in cfgs:
    env = dict(os.environ, ICAP_FORCE_CFG=str(c))
    r = subprocess.run([sys.executable, os.path.join(here, "linear_bench.py"), M, K, N, "200"], env=env, capture_output=True, text=True, timeout=120)
    print(f"cfg {c:2d}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr.strip()[-200:]}", flush=True)
