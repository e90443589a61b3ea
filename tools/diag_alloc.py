import torch, time
dev = torch.device("cuda", 0)
def T(n, sz):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        x = torch.empty(sz, dtype=torch.uint8, device=dev)
        del x
    return (time.perf_counter() - t0) / n * 1e6
for sz in (1 << 20, 50 << 20, 400 << 20, 800 << 20):
    T(3, sz)
    print(sz >> 20, "MiB: us per empty+del", T(20, sz), "device allocs", torch.cuda.memory_stats()["num_device_alloc"])
# keep-two-alive pattern
a = None
t0 = time.perf_counter()
for i in range(20):
    b = torch.empty(800 << 20, dtype=torch.uint8, device=dev); a = b
print("pingpong us", (time.perf_counter() - t0) / 20 * 1e6, torch.cuda.memory_stats()["num_device_alloc"])
# with a kernel using the buffer in between (stream use)
t0 = time.perf_counter()
for i in range(20):
    b = torch.empty(800 << 20, dtype=torch.uint8, device=dev); b[:1024].zero_(); a = b
torch.cuda.synchronize()
print("pingpong+kernel us", (time.perf_counter() - t0) / 20 * 1e6, torch.cuda.memory_stats()["num_device_alloc"])
import os
print(os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"), os.environ.get("PYTORCH_NO_HIP_MEMORY_CACHING"), os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING"))
