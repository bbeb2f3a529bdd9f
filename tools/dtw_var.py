"""Runs tools/dtw_time.py once per library in tools/variants/ (each in its own process)."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sorted(glob.glob(os.path.join(root, 'tools', 'variants', 'lib_*.so'))):
    env = dict(os.environ, ABNET3_HIP_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dtw_time.py')] + sys.argv[1:], env=env,
                       capture_output=True, text=True)
    print(os.path.basename(lib), [l for l in r.stdout.splitlines() if 'dtw' in l][-1:] or r.stderr[-300:], flush=True)
