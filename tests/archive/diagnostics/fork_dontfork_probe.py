"""Probe of the mechanism behind round 2's decode-worker SIGSEGV (gpu_augment.decode_context): does a child fork()ed
while another thread of the parent is inside a host->device copy from PAGEABLE memory lose pages?

Parent: one thread copies a large pageable buffer to the GPU in a loop.  Main thread: fork() repeatedly; each child
reads one byte of every page of that buffer (and of a bystander array allocated next to it) and exits 0; a child that
dies on a signal is counted.  Run on the GPU box:  python tests/archive/diagnostics/fork_dontfork_probe.py
Not a test (no asserts): prints the counts."""
import os
import signal
import sys
import threading
import time

import numpy as np
import torch


def main():
    n_forks = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    buf = np.ones(64 << 20, dtype=np.uint8)            # pageable, 64 MiB
    bystander = np.ones(1 << 20, dtype=np.uint8)
    t_buf = torch.from_numpy(buf)
    dev = torch.empty(buf.size, dtype=torch.uint8, device="cuda:0")
    stop = threading.Event()

    def copier():
        while not stop.is_set():
            dev.copy_(t_buf)
            torch.cuda.synchronize()

    results = {"ok": 0}
    for phase, with_copy in (("idle parent", False), ("parent copying", True)):
        results = {"ok": 0}
        th = None
        if with_copy:
            stop.clear()
            th = threading.Thread(target=copier, daemon=True)
            th.start()
            time.sleep(0.2)
        for i in range(n_forks):
            pid = os.fork()
            if pid == 0:
                s = 0
                for arr in (buf, bystander):
                    s += int(arr[::4096].sum())
                os._exit(0 if s > 0 else 3)
            _, status = os.waitpid(pid, 0)
            if os.WIFSIGNALED(status):
                key = signal.Signals(os.WTERMSIG(status)).name
            else:
                key = "ok" if os.WEXITSTATUS(status) == 0 else f"exit {os.WEXITSTATUS(status)}"
            results[key] = results.get(key, 0) + 1
            time.sleep(0.003)
        if th is not None:
            stop.set()
            th.join()
        print(f"{phase}: {n_forks} forks -> {results}", flush=True)


def finalizer_probe():
    """Second mechanism: a fork()ed child that FINALIZES GPU objects it inherited (Python's garbage collector running in
    a DataLoader worker frees whatever only the parent's other threads referenced: CPython drops the other threads'
    frames in the child).  Each child destroys one kind of inherited object and exits; a signal is the finding."""
    import gc
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "syke-pic_amd"))
    from sykepic_hip.net import HipNet
    net = HipNet("resnet18", 10, weights=None)
    x = torch.zeros(2, 3, 64, 64, device="cuda:0")
    net.eval()
    net(x)
    objs = {
        "cuda tensor": lambda: torch.ones(1 << 20, device="cuda:0"),
        "cuda event": lambda: torch.cuda.Event(),
        "cuda stream": lambda: torch.cuda.Stream(),
        "pinned tensor": lambda: torch.ones(1 << 20).pin_memory(),
        "HipNet handle": lambda: net,
    }
    for name, make in objs.items():
        o = make()
        if name == "cuda event":
            o.record()
        torch.cuda.synchronize()
        pid = os.fork()
        if pid == 0:
            try:
                if name == "HipNet handle":
                    o.__del__() if hasattr(o, "__del__") else None
                del o
                gc.collect()
            finally:
                os._exit(0)
        _, status = os.waitpid(pid, 0)
        res = signal.Signals(os.WTERMSIG(status)).name if os.WIFSIGNALED(status) else f"exit {os.WEXITSTATUS(status)}"
        print(f"child finalizes inherited {name}: {res}", flush=True)
        if name != "HipNet handle":
            del o


if __name__ == "__main__":
    main()
    finalizer_probe()
