"""Sweep for the audit build of the two kernels that decide pixels with a cheap, certified projection
(pair_loss_fused_kernel, overlap_fused_kernel): every pixel the cheap form decides is decided again inside the
kernel with the reference-order functions (test_import.py:153-155,160,165-167,174) and disagreements are counted.

  tools/build_audit.sh                      -> build/variants/audit.so   (-DSLAMHIP_AUDIT)
  SLAMHIP_LIB=build/variants/audit.so python tools/audit_certificate.py OUT.json

Sweep: rotations 0 .. 40 degrees about random axes, translations 0 .. 5 m, raw depths 1 .. 65535 (uniform, near,
plane, constant), depth divisor 1 and 5000, the three TUM intrinsics and doubled ones, 640x480 and 1280x960."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd import capi                       # noqa: E402
from slam_amd import frame_pairs as fp          # noqa: E402

NAMES = ["pixels", "decided_by_cheap_form", "left_to_exact_form", "inside_window_differs", "taps_exist_differs",
         "floor_differs", "kept_by_point_cloud_differs", "taps_gathered_outside_window"]


def rotation(axis, deg):
    a = np.asarray(axis, np.float64)
    a /= np.linalg.norm(a)
    t = np.deg2rad(deg)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * Kx + (1 - np.cos(t)) * Kx @ Kx


def depth_batch(kind, B, H, W, gen, dev):
    if kind == "uniform":
        d = torch.randint(1, 65536, (B, H, W), generator=gen, device=dev)
    elif kind == "near":
        d = torch.randint(1, 300, (B, H, W), generator=gen, device=dev)
    elif kind == "plane":
        yy = torch.arange(H, device=dev).view(1, H, 1).float()
        xx = torch.arange(W, device=dev).view(1, 1, W).float()
        d = (4000 + 30 * yy + 11 * xx + torch.randint(0, 40, (B, H, W), generator=gen, device=dev)).clamp(1, 65535).long()
    else:
        d = torch.full((B, H, W), 5000, device=dev, dtype=torch.long)
    d[:, ::13, ::7] = 0
    return (d.to(torch.int32) & 0xFFFF).to(torch.int16).view(torch.uint16)     # (values above 32767 wrap: same bits)


def main():
    out = sys.argv[1]
    lib = capi.load()
    for name in ("slamhip_audit_read", "slamhip_audit_ovl_read"):
        fn = getattr(lib, name)     # AttributeError: not the audit build
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1)
    rs = np.random.RandomState(2)
    buf = (ctypes.c_ulonglong * 8)()
    lib.slamhip_audit_read(buf, 1)
    lib.slamhip_audit_ovl_read(buf, 1)
    intr = {"fr1": (517.3, 516.5, 318.6, 255.3), "fr2": (520.9, 521.0, 325.1, 249.7), "fr3": (535.4, 539.2, 320.1, 247.6)}
    rows, B, launches = [], 4, 0
    for (H, W) in ((480, 640), (960, 1280)):
        for iname, K0 in intr.items():
            for scale in (1.0, 2.0):
                if (W == 1280) != (scale == 2.0) and not (W == 640 and iname == "fr1"):
                    continue      # doubled intrinsics on 640x480 once (fr1); 1280x960 runs with doubled ones
                K = tuple(v * scale for v in K0)
                for factor in (1.0, 5000.0):
                    for kind in ("uniform", "near", "plane", "constant"):
                        tot = np.zeros(8, np.uint64)
                        tot_o = np.zeros(8, np.uint64)
                        for deg in (0.0, 0.5, 3.0, 10.0, 20.0, 40.0):
                            for t_m in (0.0, 0.05, 0.5, 2.0, 5.0):
                                rel = np.tile(np.eye(4), (B, 1, 1))
                                for b in range(B):
                                    rel[b, :3, :3] = rotation(rs.normal(size=3), deg)
                                    tv = rs.normal(size=3)
                                    rel[b, :3, 3] = tv / np.linalg.norm(tv) * t_m * (5000.0 / factor)   # metres in depth units
                                d1 = depth_batch(kind, B, H, W, gen, dev)
                                d2 = depth_batch("uniform", B, H, W, gen, dev).reshape(B, 1, H, W)
                                c1 = torch.randint(0, 256, (B, H, W, 3), generator=gen, device=dev, dtype=torch.uint8)
                                c2 = torch.randint(0, 256, (B, 1, H, W, 3), generator=gen, device=dev, dtype=torch.uint8)
                                r = torch.from_numpy(rel).to(dev)
                                fp.pair_loss(d1, c1, d2, c2, r.reshape(B, 1, 4, 4), K, factor)
                                fp.overlap_fused(d1, r, K, (float(W), float(H)), factor)
                                launches += 2
                        lib.slamhip_audit_read(buf, 1)
                        tot += np.array(list(buf), np.uint64)
                        lib.slamhip_audit_ovl_read(buf, 1)
                        tot_o += np.array(list(buf), np.uint64)
                        rows.append({"size": "%dx%d" % (W, H), "intrinsics": "%s x%g" % (iname, scale), "factor": factor,
                                     "depth": kind, "pair_loss_fused": dict(zip(NAMES[:8], [int(v) for v in tot[:8]])),
                                     "overlap_fused": {NAMES[i]: int(tot_o[i]) for i in (0, 1, 2, 3, 6)}})
                        print(rows[-1], flush=True)
    def total(key, name):
        return int(sum(r[key][name] for r in rows))
    summary = {}
    for key in ("pair_loss_fused", "overlap_fused"):
        px, dec, rest = total(key, "pixels"), total(key, "decided_by_cheap_form"), total(key, "left_to_exact_form")
        mism = {n: total(key, n) for n in rows[0][key] if n.endswith("differs")}
        summary[key] = {"pixels": px, "decided_by_cheap_form": dec, "left_to_exact_form": rest,
                        "exact_fraction": rest / max(px, 1), "mismatches": mism, "mismatches_total": int(sum(mism.values()))}
        if key == "pair_loss_fused":
            summary[key]["taps_gathered_outside_window"] = total(key, "taps_gathered_outside_window")
    json.dump({"what": "in-kernel audit of the certified cheap projection against the reference-order projection",
               "sweep": "rotation 0/0.5/3/10/20/40 deg x translation 0/0.05/0.5/2/5 m x depth uniform 1..65535 / near 1..299 / "
                        "plane / constant x factor 1 / 5000 x fr1, fr2, fr3 (and doubled) x 640x480, 1280x960; %d pairs per point" % B,
               "launches": launches, "summary": summary, "groups": rows}, open(out, "w"), indent=1)
    print(json.dumps(summary, indent=1))
    return 0 if all(s["mismatches_total"] == 0 for s in summary.values()) else 1


if __name__ == "__main__":
    sys.exit(main())
