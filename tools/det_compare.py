import os, subprocess, sys, numpy as np
HERE = "tests"
FALLBACK_ENV = {"SLAMHIP_IGEMM256": "0", "SLAMHIP_IGEMM_ROW3": "0", "SLAMHIP_WGRAD_ROWS64": "0",
                "SLAMHIP_WGRAD_ROWGROUP": "0", "SLAM_FUSE_FIRST_BWD": "0", "SLAM_FIRST_RECOMPUTE": "0"}
def run(out, env):
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, "tests/encoder_worker.py", "fullsize", out], env=e, check=True, capture_output=True)
    return np.load(out)
def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
d = run("/tmp/d.npz", {"SLAMHIP_DETERMINISTIC": "1"})
f = run("/tmp/f.npz", dict(FALLBACK_ENV, SLAMHIP_DETERMINISTIC="1"))
for k in d.files:
    if k.startswith(("z_", "g_", "n_", "loss", "output")):
        print(k, round(rel(d[k], f[k]), 5))
