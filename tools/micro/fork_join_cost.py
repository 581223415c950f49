#!/usr/bin/env python3
"""What does a fork/join onto a second HIP stream cost between small kernels? (GPU box.)
A chain of small elementwise kernels on the main stream with and without a side chain forked off by events:
serial time, forked time, and the time of the main chain alone."""
import time
import torch

dev = 'cuda'
a = torch.randn(64000, 32, device=dev)
bufs = [torch.empty_like(a) for _ in range(8)]
big = torch.randn(64000 * 3, 32, device=dev)
bigo = torch.empty_like(big)
side = torch.cuda.Stream()


def main_chain(n):
    x = a
    for i in range(n):
        torch.mul(x, 1.0001, out=bufs[i % 4])
        x = bufs[i % 4]


def side_chain(n):
    for i in range(n):
        torch.add(big, 1.0, out=bigo)


def run(mode, reps=200, n_main=4, n_side=2):
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if mode == 'serial':
            main_chain(n_main)
            side_chain(n_side)
        elif mode == 'main_only':
            main_chain(n_main)
        elif mode == 'side_only':
            side_chain(n_side)
        else:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                side_chain(n_side)
                ev2 = torch.cuda.Event()
                ev2.record(side)
            main_chain(n_main)
            main.wait_event(ev2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for mode in ('serial', 'main_only', 'side_only', 'forked', 'serial', 'forked'):
    run(mode, 20)
    print(f'eager {mode:10s} {run(mode):8.1f} us per repeat')

# the same under hipGraph replay (no host cost at all)
for mode in ('serial', 'main_only', 'side_only', 'forked'):
    g = torch.cuda.CUDAGraph()
    s0 = torch.cuda.Stream()
    with torch.cuda.stream(s0):
        run(mode, 3)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s0):
            main = torch.cuda.current_stream()
            for _ in range(20):
                if mode == 'serial':
                    main_chain(4); side_chain(2)
                elif mode == 'main_only':
                    main_chain(4)
                elif mode == 'side_only':
                    side_chain(2)
                else:
                    ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
                    with torch.cuda.stream(side):
                        side_chain(2)
                        ev2 = torch.cuda.Event(); ev2.record(side)
                    main_chain(4)
                    main.wait_event(ev2)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f'graph {mode:10s} {(time.perf_counter() - t0) / 400 * 1e6:8.1f} us per repeat')
