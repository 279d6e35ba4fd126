import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in sys.argv[1:]:
    env = dict(os.environ, BICG_SPMV_VARIANT=v)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spmv_only.py")], env=env, capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-300:], flush=True)
