#!/usr/bin/env python3
"""Does a pending RCCL work item make the watchdog thread abort a hipGraph capture that starts right after it?
One forced rank.  argv[1]: 'side' = collective issued from the stream that then captures, 'main' = from the default stream,
'drain' = as 'side' but sleep 0.5 s (five watchdog periods) between the collective and the capture."""
import os, sys, time, torch, torch.distributed as dist
mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
buf = torch.ones(1 << 20, device="cuda")
dist.all_reduce(buf); torch.cuda.synchronize(); time.sleep(0.5)
side = torch.cuda.Stream(priority=-1)
y = torch.zeros(1 << 20, device="cuda")
for trial in range(6):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side if mode in ("side", "drain") else torch.cuda.current_stream()):
        for _ in range(4):
            dist.all_reduce(buf)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if mode == "drain":
        time.sleep(0.5)
    time.sleep(0.013 * trial)                      # sweep the phase against the watchdog's 100 ms period
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        for _ in range(50):
            y.add_(1.0)
            time.sleep(0.005)                      # hold the capture open ~250 ms
    g.replay(); torch.cuda.synchronize()
    print("trial", trial, "ok", float(y[0]), flush=True)
dist.destroy_process_group()
print("PROBE", mode, "PASSED")
