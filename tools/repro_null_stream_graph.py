"""Does NULL-stream work between replays disturb a hipGraph made of plain PyTorch ops (no pdae kernels)?
   python tools/repro_null_stream_graph.py [created]      'created' = everything on one created stream"""
import sys
import torch

one = len(sys.argv) > 1 and sys.argv[1] == 'created'
dev = torch.device('cuda')
if one:
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
torch.manual_seed(0)
x = torch.randn(262144, 256, device=dev)
w1 = torch.randn(512, 256, device=dev) * 0.05
w2 = torch.randn(384, 512, device=dev) * 0.05
big = torch.randn(29_000_000, device=dev)


def work():
    h = torch.relu(x @ w1.t())
    h = torch.nn.functional.layer_norm(h, (512,))
    y = (h @ w2.t()).reshape(8192, 32, 384).max(1)[0]
    return y.square().mean()


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        work()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = work()
vals = []
for i in range(6):
    g.replay()
    torch.cuda.current_stream().synchronize()
    vals.append(out.item())
    if i == 2:
        z = big.cpu()                      # 116 MB device-to-host on the current stream (NULL unless 'created')
        zz = big.clone()
print('created stream' if one else 'NULL stream', vals)
