"""Lab: bench.py twice in one call on the same box with a module attribute of singa_amd.ops flipped (A/B of a host-side
choice without an environment switch in the product):  python tools/lab/ab_bench.py DW_QUEUE [bench args...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
attr, rest = sys.argv[1], sys.argv[2:]
for val in ("True", "False", "True"):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.argv = ['bench.py'] + {rest!r}; import singa_amd.ops as o; o.{attr} = {val}; "
            f"import runpy; runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    d = json.loads(line[-1]) if line else {}
    print(f"{attr}={val}: step {d.get('ms_per_step_median')} ms, proxy {d.get('strong_proxy_ms')} ms", flush=True)
