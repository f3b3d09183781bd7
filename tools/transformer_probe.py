"""The nine Transformer layers alone (tokens of B faces), eager and as a captured graph; CF_LIB_PATH picks the build."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codeformer_amd import ops
from codeformer_amd.archs.codeformer_arch import TransformerSALayer
torch.manual_seed(0)
layers = [TransformerSALayer(512, 8, 1024).cuda().eval() for _ in range(9)]
pos = torch.randn(256, 512, device='cuda') * 0.02
code = ops.GSPLIT if os.environ.get('LN_SPLIT', '1') == '1' else 0
for B in (1, 16):
    X0 = torch.randn(B * 256, 512, device='cuda')
    def fwd():
        X = X0
        for l in layers:
            X = l.forward_tokens(X, pos, B, code=0, code_ln=code)
        return X
    for _ in range(3):
        y = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        y = fwd()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 30
    t0 = time.perf_counter()
    for _ in range(30):
        y = fwd()
    cpu_only = (time.perf_counter() - t0) / 30      # host time to enqueue one pass (queue not drained)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            y = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 30
    print(f'B={B:2d}: eager {eager * 1e3:.3f} ms  (host enqueue {cpu_only * 1e3:.3f} ms)  graph {graph * 1e3:.3f} ms  checksum {float(y.double().sum()):.6f}', flush=True)
