import os, sys; sys.path.insert(0, '.')
os.environ.setdefault("MIFC_LIB_PATH", os.path.join(os.getcwd(), "mi-fieldcalc_amd", "libmifc_measure.so"))  # measurement build
import torch, mi_fieldcalc_amd as fc
ctx = fc.Context(0)
for n in (1440*720*3, 1000*4+ 4*77, 256*4*4096*2 + 4*5):
    a = torch.randn(n, device='cuda'); b = torch.randn(n, device='cuda')
    for blocks in (0, 7, 1024):
        x = torch.zeros_like(a); y = torch.zeros_like(b)
        assert ctx.bench_stream2(3, blocks, x, y, a, b)
        torch.cuda.synchronize()
        assert torch.equal(x, a) and torch.equal(y, b), (n, blocks)
print("split-role copy correct")
for n in (1440*720*3, 1000*4 + 4*77, 384*4*50 + 4*5):
    a = torch.randn(n, device='cuda'); b = torch.randn(n, device='cuda')
    for blocks in (1, 7, 1024):
        x = torch.zeros_like(a); y = torch.zeros_like(b)
        assert ctx.bench_stream2(4, blocks, x, y, a, b)
        torch.cuda.synchronize()
        assert torch.equal(x, a) and torch.equal(y, b), (n, blocks)
print("band copy correct")
# write-only yardsticks: every element of both outputs gets the repeating (1, 2, 3, 4)
n = 1440 * 720 * 3
a = torch.randn(n, device='cuda'); b = torch.randn(n, device='cuda')
want = torch.tensor([1.0, 2.0, 3.0, 4.0], device='cuda').repeat(n // 4)
for variant in (5, 6, 7, 8):
    x = torch.zeros_like(a); y = torch.zeros_like(b)
    assert ctx.bench_stream2(variant, 0, x, y, a, b)
    torch.cuda.synchronize()
    assert torch.equal(x, want) and torch.equal(y, want), variant
print("fill yardsticks write every element")
