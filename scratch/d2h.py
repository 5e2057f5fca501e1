import torch, time
n = 1920*1080*3
d = torch.empty(n, dtype=torch.uint8, device='cuda')
h = torch.empty(n, dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream()
for sz in (n, n//2):
    for rep in range(3):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            e0.record()
            for k in range(20): h[:sz].copy_(d[:sz], non_blocking=True)
            e1.record()
        torch.cuda.synchronize()
        print(sz, 'bytes', round(e0.elapsed_time(e1)/20*1000, 1), 'us per copy', round(sz/(e0.elapsed_time(e1)/20*1e-3)/1e9, 1), 'GB/s')
# two halves on two streams
s2 = torch.cuda.Stream()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20):
        with torch.cuda.stream(s): h[:n//2].copy_(d[:n//2], non_blocking=True)
        with torch.cuda.stream(s2): h[n//2:].copy_(d[n//2:], non_blocking=True)
    torch.cuda.synchronize()
    print('two streams', round((time.perf_counter()-t0)/20*1e6, 1), 'us per frame')
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20):
        with torch.cuda.stream(s): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    print('one stream', round((time.perf_counter()-t0)/20*1e6, 1), 'us per frame')
