"""Do kernels on two HIP streams of one process overlap on this box, and what does a launch cost the host then?"""
import sys
import time

import torch

dev = torch.device("cuda", 0)


def run(label, plan, xs, n, streams, mats=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        s = streams[plan(i)]
        with torch.cuda.stream(s):
            if mats is None:
                xs[plan(i)].add_(1.0)
            else:
                torch.mm(mats[plan(i)][0], mats[plan(i)][1], out=mats[plan(i)][2])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{label:34s} host {1e6 * th / n:6.2f} us/launch   total {1e6 * t / n:6.2f} us/launch", flush=True)


def main():
    n = 2000
    mats = [(torch.randn(64, 16384, device=dev), torch.randn(16384, 64, device=dev), torch.empty(64, 64, device=dev)) for _ in range(2)]
    null = torch.cuda.current_stream()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    print("small-grid long kernel (64x16384 @ 16384x64)")
    for rep in range(2):
        run("null stream only", lambda i: 0, None, n, [null, null], mats)
        run("side stream A only", lambda i: 0, None, n, [a, a], mats)
        run("A/B alternating", lambda i: i & 1, None, n, [a, b], mats)
        run("null/B alternating", lambda i: i & 1, None, n, [null, b], mats)
        run("A/B blocks of 100", lambda i: (i // 100) & 1, None, n, [a, b], mats)
    for numel in ():
        print("numel", numel)
        xs = [torch.zeros(numel, device=dev) for _ in range(2)]
        null = torch.cuda.current_stream()
        a, b = torch.cuda.Stream(), torch.cuda.Stream()
        for rep in range(2):
            run("null stream only", lambda i: 0, xs, n, [null, null])
            run("side stream A only", lambda i: 0, xs, n, [a, a])
            run("A then B (halves)", lambda i: 0 if i < n // 2 else 1, xs, n, [a, b])
            run("A/B alternating", lambda i: i & 1, xs, n, [a, b])
            run("null/B alternating", lambda i: i & 1, xs, n, [null, b])
            run("A/B blocks of 100", lambda i: (i // 100) & 1, xs, n, [a, b])
            run("null/B blocks of 100", lambda i: (i // 100) & 1, xs, n, [null, b])


if __name__ == "__main__":
    main()
