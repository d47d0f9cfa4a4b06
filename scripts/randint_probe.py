import torch
dev = torch.device("cuda:0")
n = 111_000_000; nnz = n * 30
gen = torch.Generator(device=dev); gen.manual_seed(4)
col = torch.randint(0, n, (nnz,), generator=gen, device=dev, dtype=torch.int32)
for name, sl in (("head", col[:50_000_000]), ("mid", col[nnz//2: nnz//2 + 50_000_000]), ("tail", col[-50_000_000:])):
    print(name, int(sl.min()), int(sl.max()), float(sl.float().mean()), int(torch.unique(sl[:5_000_000]).numel()))
print("equal head/2^32 offset?", bool(torch.equal(col[:1000], col[2**32 - 2**32 % 1: 2**32 - 2**32 % 1 + 1000])) if nnz > 2**32 + 1000 else None)
print("eq col[k] vs col[k+2^31]:", float((col[:10_000_000] == col[2**31:2**31 + 10_000_000]).float().mean()))
print("eq col[k] vs col[k+2^32]:", float((col[:10_000_000] == col[2**32 - 4294967296 + 2**32 - 2**32: 10_000_000]).float().mean()))
