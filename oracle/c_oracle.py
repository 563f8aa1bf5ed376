"""ctypes front end of oracle/libslam_oracle.so (the C restatement of the
oracle).  TEST INFRASTRUCTURE ONLY - see oracle/slam_oracle.c."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslam_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s missing: run `make -C oracle`" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.slam_oracle_pair.restype = None
        _lib.slam_oracle_pair_loss.restype = None
        _lib.slam_oracle_pair_loss_mt.restype = None
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def pair_loss(d1, c1, d2, c2, rel, K, factor, threads=1):
    """Same contract as slam_oracle.pair_loss (numpy in, numpy out); `threads` host threads
    share the pairs (the results do not depend on it)."""
    d1 = np.ascontiguousarray(d1, np.uint16)
    c1 = np.ascontiguousarray(c1, np.uint8)
    B, H, W = d1.shape
    d2 = np.ascontiguousarray(d2, np.uint16).reshape(B, -1, H, W)
    k = d2.shape[1]
    c2 = np.ascontiguousarray(c2, np.uint8).reshape(B, k, H, W, 3)
    rel = np.ascontiguousarray(rel, np.float64).reshape(B, k, 16)
    Kc = (ctypes.c_double * 4)(*[float(x) for x in K])
    sums = np.zeros((B, k, 4))
    counts = np.zeros((B, k, 4), np.uint32)
    load().slam_oracle_pair_loss_mt(_p(d1), _p(c1), _p(d2), _p(c2), _p(rel), Kc, ctypes.c_double(float(factor)),
                                    B, k, H, W, _p(sums), _p(counts), int(threads))
    return sums, counts


def warp_pair(d1, c1, d2, c2, T, K, factor):
    """One pair, every per-pixel output; dict with the keys of slam_oracle.warp_pair
    that the C side produces."""
    d1 = np.ascontiguousarray(d1, np.uint16)
    H, W = d1.shape
    c1 = np.ascontiguousarray(c1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint16)
    c2 = np.ascontiguousarray(c2, np.uint8)
    T = np.ascontiguousarray(T, np.float64).reshape(16)
    Kc = (ctypes.c_double * 4)(*[float(x) for x in K])
    o = dict(sums=np.zeros(4), counts=np.zeros(4, np.uint32),
             warped_rgb=np.zeros((H, W, 3), np.float32), warped_depth=np.zeros((H, W), np.float32),
             r_photo=np.zeros((H, W, 3), np.float32), r_geo=np.zeros((H, W), np.float32),
             mask=np.zeros((H, W), np.uint8), iu_iv=np.zeros((H, W, 2), np.int32),
             uvd=np.zeros((H, W, 3)), xyz=np.zeros((H, W, 3)), xyz2=np.zeros((H, W, 3)))
    load().slam_oracle_pair(_p(d1), _p(c1), _p(d2), _p(c2), _p(T), Kc, ctypes.c_double(float(factor)), H, W,
                            _p(o["sums"]), _p(o["counts"]), _p(o["warped_rgb"]), _p(o["warped_depth"]),
                            _p(o["r_photo"]), _p(o["r_geo"]), _p(o["mask"]), _p(o["iu_iv"]), _p(o["uvd"]),
                            _p(o["xyz"]), _p(o["xyz2"]))
    return o
