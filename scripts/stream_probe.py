"""Which torch streams share a hardware queue?  Two spin kernels (torch.cuda._sleep, one thread each) on two streams take
T if the streams are independent and 2T if they serialise."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("Q", "8"))
import torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
n = int(os.environ.get("NS", "12"))
null = torch.cuda.default_stream(dev)
streams = [null] + [torch.cuda.Stream(device=dev) for _ in range(n)]
hi = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(3)]
streams += hi
names = ["null"] + [f"s{i}" for i in range(n)] + [f"h{i}" for i in range(3)]
CYC = 4_000_000
def pair(a, b):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.cuda.stream(a): torch.cuda._sleep(CYC)
    with torch.cuda.stream(b): torch.cuda._sleep(CYC)
    torch.cuda.synchronize(); return time.perf_counter() - t
base = pair(null, null) / 2
print("GPU_MAX_HW_QUEUES", os.environ["GPU_MAX_HW_QUEUES"], "one spin kernel %.2f ms" % (base * 1e3))
print("      " + " ".join(f"{x:>4}" for x in names))
for i, a in enumerate(streams):
    row = []
    for j, b in enumerate(streams):
        row.append("  = " if i == j else ("SER " if pair(a, b) > 1.6 * base else "  . "))
    print(f"{names[i]:>5} " + " ".join(row))
