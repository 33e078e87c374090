"""Stand-alone probe of the NULL-stream hazard: a captured graph with INDEPENDENT root branches, each
reading a device buffer that a pinned non_blocking host-to-device copy refreshes right before every replay -- all on
the legacy NULL stream, or all on one created stream ('created').  After a device-to-host copy at step 20, does every
branch of every replay still see the values of ITS step?     python tools/repro_null_stream_graph.py [created]"""
import sys
import torch

if len(sys.argv) > 1 and sys.argv[1] == 'created':
    torch.cuda.set_stream(torch.cuda.Stream())
dev, N, R = torch.device('cuda'), 1 << 16, 4
bufs = [torch.zeros(N, dtype=torch.int64, device=dev) for _ in range(3)]
ring = [[torch.zeros(N, dtype=torch.int64).pin_memory() for _ in range(3)] for _ in range(R)]
table = torch.arange(4096, device=dev, dtype=torch.float32)
big = torch.randn(29_000_000, device=dev)
w = torch.randn(2048, 2048, device=dev)


def work():
    outs = [table[b % 4096].sum() for b in bufs]                  # three independent roots
    heavy = (w @ w).sum() * 0                                      # a fourth, long one
    return torch.stack(outs) + heavy


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    work()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = work()
bad, results = 0, []
for i in range(200):
    slot = ring[i % R]
    for j, (b, h) in enumerate(zip(bufs, slot)):
        h.fill_((i * 3 + j) % 4096)
        b.copy_(h, non_blocking=True)
    g.replay()
    results.append((i, out.clone()))
    if i == 20:
        z = big.cpu()                                              # the poke
    if i % 4 == 3:                                                 # the host may run at most R steps ahead
        torch.cuda.current_stream().synchronize()
torch.cuda.synchronize()
for i, o in results:
    want = [float(N * ((i * 3 + j) % 4096)) for j in range(3)]
    if o.tolist() != want:
        bad += 1
print('created stream' if len(sys.argv) > 1 else 'NULL stream', ':', bad, 'of 200 replays saw stale / wrong draws')
