import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n, c, hw = 16, 64, 256
x = torch.randn(n, c, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
w = (torch.randn(c, c, 3, 3, device='cuda') * 0.06).contiguous(memory_format=torch.channels_last)
b = torch.zeros(c, device='cuda')
for d in (1, 2, 3, 4):
    t = timeit(lambda: F.conv2d(x, w, b, padding=d, dilation=d))
    flop = 2.0 * n * hw * hw * c * c * 9
    print('dilation %d at [%d, %d, %d, %d]: %.3f ms %.1f TF %.3f' % (d, n, c, hw, hw, t, flop / t / 1e9, flop / t / 1e9 / 157.3))
