"""Does HIP stream priority change how fast a small kernel gets through while a long kernel saturates the GPU?"""
import time, torch
dev = torch.device("cuda", 0)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
big = torch.randn(1 << 28, device=dev)       # 1 GiB: elementwise chain = long, many-workgroup kernels
small = torch.randn(1 << 20, device=dev)
def run(p_long, p_short):
    s_long, s_short = torch.cuda.Stream(device=dev, priority=p_long), torch.cuda.Stream(device=dev, priority=p_short)
    torch.cuda.synchronize()
    lat = []
    for rep in range(20):
        with torch.cuda.stream(s_long):
            y = big
            for _ in range(6):
                y = torch.sin(y) * 1.0001       # ~6 long kernels back to back
        time.sleep(0.0005)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s_short):
            e0.record()
            z = small
            for _ in range(8):
                z = torch.sin(z) * 1.0001       # 8 short dependent kernels (a latency-bound chain)
            e1.record()
        torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1))
    lat.sort()
    return lat[len(lat) // 2]
torch.cuda.synchronize()
alone = []
s = torch.cuda.Stream(device=dev)
for rep in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(); z = small
        for _ in range(8): z = torch.sin(z) * 1.0001
        e1.record()
    torch.cuda.synchronize(); alone.append(e0.elapsed_time(e1))
print("short chain alone: %.3f ms" % sorted(alone)[10])
print("long normal(0), short normal(0): %.3f ms" % run(0, 0))
print("long normal(0), short high(-1): %.3f ms" % run(0, -1))
try:
    print("long low(+1?), short normal: %.3f ms" % run(1, 0))
except Exception as e:
    print("low priority not available:", e)
