"""PyTorch's global-reduce kernels (memset of a semaphore + kernel) replayed from a hipGraph"""
import torch
dev = "cuda"
for shape, dims in (((32768, 96), (0,)), ((32, 96, 32, 32), (0, 2, 3))):
    for cl in (False, True):
        if cl and len(shape) != 4: continue
        x = torch.randn(*shape, device=dev)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        ref = x.double().sum(dim=dims)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): y = x.sum(dim=dims)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = x.sum(dim=dims)
        errs = []
        for _ in range(4):
            g.replay(); torch.cuda.synchronize()
            errs.append(float((y.double() - ref).abs().max() / ref.abs().max()))
        eager = float((x.sum(dim=dims).double() - ref).abs().max() / ref.abs().max())
        print("RESULT sum", shape, dims, "channels_last" if cl else "contiguous", "eager %.1e replays" % eager, " ".join("%.1e" % e for e in errs), flush=True)
