"""Are hipGraph MEMSET nodes ordered against their neighbour kernel nodes when a captured graph is replayed (ROCm 7.2, gfx950)?

The framework's multi-block reduction (at::native::reduce_kernel with a global reduce) zeroes its semaphores with hipMemsetAsync before the
launch -- inside a stream capture that is a memset NODE.  Round 2 saw history-dependent bias gradients from exactly those reductions in
the replayed 880-node train step; round 1 saw our own hipMemsetAsync-before-accumulate race in ~1 of 3 replays.  This probe isolates the
node type:

  part 0 (bare): [ hipMemsetAsync(buf, byte) | out = buf + 0 ] and nothing else in the graph, buf pre-filled with 5.0 before every replay:
         out must be the byte pattern everywhere.  A wrong value here is not an ordering question: the node itself writes the wrong thing.
  part 1 (explicit): R x [ heavy GEMM | K0: buf = 5 | hipMemsetAsync(buf, 0) | K2: buf += 1 | out_i = buf + 0 ]   -> every out_i must be 1
         6 = the memset ran before K0 (or not at all), 0 = it ran after K2, 5/other = it overlapped;  control: the memset as a fill kernel
  part 2 (framework): R x [ heavy GEMM | y_i = x_i.sum(0) over (16384, 96) ] with x_i rewritten between replays, against the fp64 sums;
         a wrong row is classified as stale (equals the previous replay's row) or other.

    python tools/memset_node_probe.py [replays]
"""
import ctypes
import sys

import torch

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
R = 12


def capture(body):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(s)                                                   # warm-up (allocator, library heuristics)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body(s)
    return g


def part0(nbytes, byte):
    n = nbytes // 4
    buf = torch.empty(n, device=dev)
    out = torch.empty(n, device=dev)

    def body(s):
        rc = hip.hipMemsetAsync(buf.data_ptr(), byte, nbytes, s.cuda_stream)
        assert rc == 0, rc
        torch.add(buf, 0.0, out=out)
    g = capture(body)
    bad, seen = 0, {}
    for _ in range(replays):
        buf.fill_(5.0)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        iv = out.view(torch.int32)
        ok = iv == (byte * 0x01010101 if byte < 0x80 else byte * 0x01010101 - (1 << 32))
        if not bool(ok.all()):
            bad += 1
            for v in torch.unique(iv[~ok])[:4].tolist():
                seen[hex(v & 0xFFFFFFFF)] = seen.get(hex(v & 0xFFFFFFFF), 0) + 1
    print(f"part 0, {nbytes:>8d} B, memset byte {byte:#04x} alone in the graph: {bad} of {replays} replays wrong; wrong words seen {dict(list(seen.items())[:6])}", flush=True)


def part1(nbytes, memset_node):
    n = nbytes // 4
    a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    c = torch.empty_like(a)
    buf = torch.empty(n, device=dev)
    outs = [torch.empty(n, device=dev) for _ in range(R)]

    def body(s):
        for i in range(R):
            torch.mm(a, a, out=c)
            buf.fill_(5.0)
            if memset_node:
                rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, s.cuda_stream)
                assert rc == 0, rc
            else:
                buf.fill_(0.0)
            buf.add_(1.0)
            torch.add(buf, 0.0, out=outs[i])
    g = capture(body)
    hist = {}
    for _ in range(replays):
        g.replay()
        torch.cuda.synchronize()
        for o in outs:
            vals, cnt = torch.unique(o, return_counts=True)
            key = tuple(sorted((float(v), int(k)) for v, k in zip(vals, cnt)))
            if key != ((1.0, n),):
                hist[key] = hist.get(key, 0) + 1
    kind = "hipMemsetAsync node" if memset_node else "fill kernel (control)"
    print(f"part 1, {nbytes:>8d} B, {kind:24s}: {sum(hist.values())} of {replays * R} segments wrong {dict(list(hist.items())[:4])}", flush=True)


def part2():
    a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    c = torch.empty_like(a)
    xs = [torch.randn(16384, 96, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    ys = [None] * R

    def body(s):
        for i in range(R):
            torch.mm(a, a, out=c)
            ys[i] = xs[i].sum(0, dtype=torch.float32)
    g = capture(body)
    wrong = stale = 0
    prev = None
    for r in range(replays):
        for x in xs:
            x.normal_()
        torch.cuda.synchronize()
        want = [x.double().sum(0) for x in xs]
        g.replay()
        torch.cuda.synchronize()
        got = [y.clone() for y in ys]
        for i in range(R):
            bad = (got[i].double() - want[i]).abs() > 1e-3 * want[i].abs() + 0.05
            if bad.any():
                wrong += 1
                if prev is not None and torch.equal(got[i][bad], prev[i][bad]):
                    stale += 1
        prev = got
    print(f"part 2, framework sum(0) of (16384, 96) inside a {2 * R}-kernel graph: {wrong} of {replays * R} reductions wrong, {stale} of them equal to the "
          f"previous replay's values at the wrong elements", flush=True)


for nbytes, byte in ((64, 0), (4096, 0), (4096, 0x3F), (1 << 22, 0)):
    part0(nbytes, byte)
for nbytes in (64, 4096, 1 << 22):
    part1(nbytes, True)
part1(4096, False)
part2()
print("probe done")
