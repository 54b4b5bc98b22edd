"""Does a host thread burn CPU while it waits for the GPU?  (Decides whether waiting threads eat the cgroup CPU quota.)
Measures process CPU time across a ~1 s device-side sleep for: the default wait, a blocking-sync event, the same two
after hipSetDeviceFlags(hipDeviceScheduleBlockingSync), and an event-query + nanosleep poll."""
import ctypes
import time
import torch

torch.cuda.init()
x = torch.zeros(1, device="cuda")
torch.cuda.synchronize()


def loaded_hip():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("HIP runtime not loaded")


def probe(tag):
    for label, mode in (("stream sync", 0), ("blocking event", 1), ("query + sleep 50us", 2)):
        ev = torch.cuda.Event(blocking=(mode == 1))
        w0, c0 = time.perf_counter(), time.process_time()
        torch.cuda._sleep(int(2.0e9))
        ev.record()
        if mode == 1:
            ev.synchronize()
        elif mode == 2:
            while not ev.query():
                time.sleep(50e-6)
        else:
            torch.cuda.current_stream().synchronize()
        w1, c1 = time.perf_counter(), time.process_time()
        print(f"[{tag}] {label}: wall {w1 - w0:.2f} s, process CPU {c1 - c0:.2f} s")


probe("default flags")
hip = loaded_hip()
rc = hip.hipSetDeviceFlags(ctypes.c_uint(0x04))       # hipDeviceScheduleBlockingSync
print("hipSetDeviceFlags(hipDeviceScheduleBlockingSync) ->", rc)
probe("blocking-sync device flag")
