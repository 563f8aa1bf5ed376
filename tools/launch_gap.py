import sys, torch
sys.path.insert(0, "/root/repo")
from slam_amd import capi
lib = capi.load()
buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
s = capi.current_stream()
def run(n):
    for _ in range(n):
        lib.slamhip_zero_fill(buf.data_ptr(), 16, s)
run(200); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(2000); e1.record(); torch.cuda.synchronize()
print("back-to-back 16-byte clear kernels on one stream: %.2f us per launch" % (e0.elapsed_time(e1) * 1e3 / 2000))
