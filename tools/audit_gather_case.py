import os, sys, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["SLAMHIP_LIB"] = "build/variants/audit.so"
import numpy as np, torch
from slam_amd import capi, frame_pairs as fp
lib = capi.load()
lib.slamhip_audit_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
H, W, factor = 32, 320, 5000.0
K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
for direction in (1, -1):
    B = 3
    rs = np.random.RandomState(11 + direction)
    c1 = rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    c2 = rs.randint(0, 256, size=(B, 1, H, W, 3)).astype(np.uint8)
    d2 = rs.randint(2500, 25001, size=(B, 1, H, W)).astype(np.uint16)
    d1 = np.zeros((B, H, W), np.uint16)
    rel = np.tile(np.eye(4), (B, 1, 1, 1))
    if direction > 0:
        rel[:, :, 0, 3], rel[:, :, 1, 3] = 0.5, 0.004
        cols = np.arange(0, 128); d1[:, 20, cols] = 25000
        d1[:, H - 2, cols] = np.round(K[0] * 0.5 / ((W - 1.5) - cols) * factor).astype(np.uint16)
    else:
        rel[:, :, 0, 3], rel[:, :, 1, 3] = -0.5, -0.1
        cols = np.arange(W - 64, W); d1[:, 20, cols] = 25000
        d1[:, H - 1, cols] = 2250 + 40 * np.arange(64)
    out = (ctypes.c_ulonglong * 8)()
    lib.slamhip_audit_read(out, 1)
    t = [torch.from_numpy(x).cuda() for x in (d1, c1, d2, c2, rel)]
    s, c = fp.pair_loss(*t, K, factor)
    torch.cuda.synchronize()
    lib.slamhip_audit_read(out, 1)
    print(direction, "inside, certified, left to phase 2, mismatches:", list(out)[:7], c.cpu().numpy().reshape(-1, 4)[0])
