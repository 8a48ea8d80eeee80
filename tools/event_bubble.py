"""What does a hipEventRecord between two dependent kernels of one stream cost on the GPU?  (The backward records one event per
layer on the main stream so that the weight-gradient stream can start; tools/trace_gaps.py shows a 7-8 us hole after each.)

    python tools/event_bubble.py > profiles/r04_event_bubble.txt
"""
import time
import torch

dev = torch.device("cuda", 0)
x = torch.zeros(1 << 26, device=dev)
y = torch.zeros(1 << 26, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def run(mode, n=200):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        x.add_(1.0)
        if mode in ("record", "record+wait", "record+wait+work"):
            ev = torch.cuda.Event()
            ev.record(main)
            if mode != "record":
                side.wait_event(ev)
                if mode == "record+wait+work":
                    with torch.cuda.stream(side):
                        y.add_(1.0)
        x.mul_(0.5)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for mode in ("plain", "record", "record+wait", "record+wait+work"):
    run(mode, 50)
    t = [run(mode) for _ in range(3)]
    print(f"{mode:18s} {min(t):7.2f} us per (kernel, kernel) pair")
