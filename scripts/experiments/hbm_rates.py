import torch, time
for mb in (256, 1024):
    a = torch.empty(mb * 1024 * 1024, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print('copy %d MB: %.1f us  %.2f TB/s (read + write)' % (mb, dt * 1e6, 2 * mb * 1048576 / dt / 1e12))
    for _ in range(3): a.fill_(1)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): a.fill_(1)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print('fill %d MB: %.1f us  %.2f TB/s (write)' % (mb, dt * 1e6, mb * 1048576 / dt / 1e12))
    for _ in range(3): s = a.view(torch.int64).sum()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): s = a.view(torch.int64).sum()
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print('sum %d MB: %.1f us  %.2f TB/s (read)' % (mb, dt * 1e6, mb * 1048576 / dt / 1e12))
