"""What a HIP-event pair measures with nothing (or a trivial kernel on either side) in between: the floor of the
per-kernel HIP-event timings bench.py reports next to the rocprofv3 averages."""
import torch, numpy as np
torch.cuda.init()
s = torch.cuda.current_stream()
x = torch.zeros(1, device="cuda")
def overhead(n=300):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); b.record()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) * 1e3 for a, b in evs])
o = overhead()
print("empty bracket us: mean %.2f median %.2f min %.2f max %.2f" % (o.mean(), np.median(o), o.min(), o.max()))
# with preceding kernel activity (like in a stream of kernels)
evs=[]
for i in range(300):
    x.add_(1.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); b.record(); evs.append((a,b))
    x.add_(1.0)
torch.cuda.synchronize()
o = np.array([a.elapsed_time(b)*1e3 for a,b in evs])
print("bracket between kernels us: mean %.2f median %.2f min %.2f" % (o.mean(), np.median(o), o.min()))
