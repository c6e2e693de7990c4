#!/usr/bin/env python3
"""The Huffman literal kernels alone: streams of a given alphabet and size through naf_gpu_zstd_compress / _decompress, one lane
per stream (NAF_GPU_HUF_PAR=0) against 2^N parts per stream (NAF_GPU_HUF_PAR=N), for the block sizes of this build (32 KiB) and of
libzstd (128 KiB).  Prints the kernel's device time per configuration.   tools/perf_huf.py [sizes in MB ...]   (GPU box, repo root)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import capi

sizes = [int(a) for a in sys.argv[1:]] or [16, 128, 1024, 4096]
ctx = capi.Context(0)
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(1)


def make(kind, n):
    if kind == "pairs+rare":       # pair codes of a GC-poor genome and a rare 17th symbol: codes up to 11 bits (the compact table form)
        p = torch.tensor([.0826, .0826, .0854, .0574, .0826, .0574, .0126, .0604, .0604, .042, .042, .0574, .0854, .0604, .0604, .0826, 2e-5], device=dev)
    elif kind == "pairs":
        p = torch.tensor([.0826, .0826, .0854, .0574, .0826, .0574, .0126, .0604, .0604, .042, .042, .0574, .0854, .0604, .0604, .0826], device=dev)
    else:                           # 41 equally likely quality values
        p = torch.ones(41, device=dev)
    cum = torch.cumsum(p / p.sum(), 0)[:-1].contiguous()
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    for a in range(0, n, 1 << 28):
        e = min(n, a + (1 << 28))
        out[a:e] = torch.bucketize(torch.rand(e - a, device=dev, generator=g), cum).to(torch.uint8) + 33
    return out


def kernel_ms(frame, n, env):
    for k, v in env.items():
        os.environ[k] = v
    ctx.zstd_decompress(frame, n + 64)
    ctx.set_timing(True)
    out = ctx.zstd_decompress(frame, n + 64)
    kt = {nm: ms for nm, ms, k in ctx.get_timing()}
    ctx.set_timing(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ctx.zstd_decompress(frame, n + 64); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    for k in env:
        del os.environ[k]
    return kt.get("zstd_huf_literals", 0.0), kt.get("zstd_build_huf", 0.0), wall, out


for kind in ("pairs", "pairs+rare", "qual41"):
    for mb in sizes:
        n = mb << 20
        data = make(kind, n)
        for blog in ("15", "17"):
            os.environ["NAF_GPU_BLOCK_LOG"] = blog
            frame = ctx.zstd_compress(data).clone()
            del os.environ["NAF_GPU_BLOCK_LOG"]
            res = []
            for par in ("0", "1", "2", "3", "4", "5", "6"):
                h, b, w, out = kernel_ms(frame, n, {"NAF_GPU_HUF_PAR": par, "NAF_GPU_HUF_PART": "128"})
                if par == "0":
                    assert torch.equal(out, data)
                res.append("P=%-2d %7.3f" % (1 << int(par), h))
            h, b, w, out = kernel_ms(frame, n, {})
            print("%-10s %5d MB  blocks of 2^%s  (%.3f of raw)  huf_literals ms: %s | auto %7.3f (tables %.3f, call %.3f)" % (kind, mb, blog, frame.numel() / n, "  ".join(res), h, b, w), flush=True)
        del data
