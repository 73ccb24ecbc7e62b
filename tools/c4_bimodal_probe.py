#!/usr/bin/env python
"""Why does `row_stats_tile2d_kernel` measure 0.90 ms in some processes and 1.10 ms in others (same binary, same inputs)?
(EXPERIMENTS.md 11: offsets INSIDE an allocation do not move it, where the allocation landed does.)

  python tools/c4_bimodal_probe.py run            ONE process, the c4 batch placed in several ways -- torch's allocator, behind
                                                  fillers of 40 / 120 GB (other physical pages), raw hipMalloc / hipFree three
                                                  times over, 2 MiB-aligned inside an over-allocation -- and per placement the
                                                  stage times of compute_rnnt_loss_async (4 calls, the first dropped)
  rocprofv3 --pmc <counters> --kernel-trace -d DIR -o pmc -- python tools/c4_bimodal_probe.py run
  python tools/c4_bimodal_probe.py table DB [...] per dispatch of the statistics kernel (in launch order = placement order):
                                                  duration and every counter of the pass, so that a counter that moves with
                                                  the kernel's time can be named
tools/c4_bimodal_session.sh runs the whole set on one box."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "warp-transducer_amd"))


def table(paths):
    import sqlite3
    for path in paths:
        db = sqlite3.connect(path)
        rows = db.execute("select dispatch_id, name, duration, counter_name, sum(counter_value), count(*), min(counter_value), max(counter_value) "
                          "from pmc_events where name like '%row_stats_tile2d%' or name like '%grad_flat%' group by 1, 4 order by 1").fetchall()
        by = {}
        for did, name, dur, cname, val, cnt, lo, hi in rows:
            rec = by.setdefault(did, {"kernel": "stats" if "row_stats" in name else "grad", "us": dur / 1e3})
            rec[cname] = val
            if cnt > 1:                                           # one row per instance (channel, XCD ...): the spread shows an imbalance
                rec[cname + " max/min over %d instances" % cnt] = hi / lo if lo else float("inf")
        names = sorted({k for r in by.values() for k in r} - {"kernel", "us"})
        print("# %s" % path)
        print("| # | kernel | us | " + " | ".join(names) + " |")
        print("|---|---|---|" + "---|" * len(names))
        for i, did in enumerate(sorted(by)):
            r = by[did]
            print("| %d | %s | %.1f | " % (i, r["kernel"], r["us"]) + " | ".join("%.6g" % r.get(n, float("nan")) for n in names) + " |")


def run():
    import torch
    from warprnnt_pytorch import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    N, T, U, A = 64, 1500, 301, 50
    E = N * T * U * A
    nbytes = E * 4
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lab = torch.randint(1, A, (N, U - 1), device=dev, dtype=torch.int32)
    tl = torch.full((N,), T, dtype=torch.int32, device=dev)
    ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    costs = torch.empty(N, device=dev)
    wsb = _lib.workspace_bytes(T, U, N, True, 4)
    src = torch.rand(E // 64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=stream, blank_label=0, maxT=T, maxU=U, batch_first=True)

    def fill(ptr):
        for i in range(64):
            assert hip.hipMemcpy(ptr + i * (nbytes // 64), src.data_ptr(), nbytes // 64, 3) == 0

    def measure(label, a_ptr, g_ptr, w_ptr, reps=4):
        fill(a_ptr)
        lib.rnnt_profile_reset(); lib.rnnt_profile_enable(1)
        for i in range(reps):
            if i == 1:
                torch.cuda.synchronize(); lib.rnnt_profile_collect(); lib.rnnt_profile_reset()
            st = lib.compute_rnnt_loss_async(a_ptr, g_ptr, lab.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N, costs.data_ptr(), None,
                                             w_ptr, opt, 0)
            assert st == 0
            torch.cuda.synchronize(); lib.rnnt_profile_collect()
        ms = (C.c_double * 5)()
        calls = lib.rnnt_profile_read(ms, 5)
        lib.rnnt_profile_enable(0)
        rec = {"placement": label, "acts": hex(a_ptr), "grads": hex(g_ptr), "ws": hex(w_ptr),
               "acts_mod_2MiB": a_ptr % (2 << 20), "stats_ms": round(ms[0] / calls, 4), "lattice_ms": round(ms[1] / calls, 4),
               "coef_ms": round(ms[2] / calls, 4), "grad_ms": round(ms[3] / calls, 4), "loss0": float(costs[0])}
        print(json.dumps(rec), flush=True)

    def torch_placement(label, filler_gb=0):
        filler = torch.empty(filler_gb << 30, dtype=torch.uint8, device=dev) if filler_gb else None
        a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        g = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        w = torch.empty(wsb, dtype=torch.uint8, device=dev)
        measure(label, a.data_ptr(), g.data_ptr(), w.data_ptr())
        del a, g, w, filler
        torch.cuda.empty_cache()

    def raw_placement(label, align=0):
        ptrs = []
        for size in (nbytes + align, nbytes + align, wsb + align):
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), size) == 0
            ptrs.append(p.value)
        al = [(p + align - 1) // align * align if align else p for p in ptrs]
        measure(label, *al)
        for p in ptrs:
            hip.hipFree(p)

    steps = [lambda: torch_placement("torch allocator, fresh process"),
             lambda: torch_placement("torch allocator, again (freed and re-allocated)"),
             lambda: torch_placement("torch allocator behind a 40 GB filler", 40),
             lambda: torch_placement("torch allocator behind a 120 GB filler", 120),
             lambda: raw_placement("hipMalloc #1"), lambda: raw_placement("hipMalloc #2"), lambda: raw_placement("hipMalloc #3"),
             lambda: raw_placement("hipMalloc, pointers rounded up to 2 MiB", 2 << 20),
             lambda: raw_placement("hipMalloc, pointers rounded up to 1 GiB", 1 << 30),
             lambda: torch_placement("torch allocator, last")]
    for s in steps:
        try:
            s()
        except Exception as e:                                  # noqa: BLE001 -- a placement that cannot be made is reported, not fatal
            print(json.dumps({"placement": "FAILED", "error": repr(e)}), flush=True)


def matrix():
    """WHICH allocation carries the state?  Three activation buffers and three workspaces, allocated behind fillers of
    different sizes (different physical pages), all kept alive; every combination is measured.  With the development library
    (WARP_RNNT_PATH=.../lib/dev, RNNT_TUNE_LIVE=1) each combination is measured with every tile order of the statistics
    kernel (RNNT_TUNE=t2ord=0|1|2) -- same buffers, same process, alternating."""
    import torch
    from warprnnt_pytorch import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    N, T, U, A = 64, 1500, 301, 50
    E = N * T * U * A
    lab = torch.randint(1, A, (N, U - 1), device=dev, dtype=torch.int32)
    tl = torch.full((N,), T, dtype=torch.int32, device=dev)
    ll = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    costs = torch.empty(N, device=dev)
    wsb = _lib.workspace_bytes(T, U, N, True, 4)
    stream = torch.cuda.current_stream().cuda_stream
    opt = _lib.rnntOptions(loc=1, num_threads=0, stream=stream, blank_label=0, maxT=T, maxU=U, batch_first=True)
    src = torch.rand(E // 64, device=dev)
    acts, wss, keep = [], [], []
    for gb in (0, 7, 33):                                          # fillers stay allocated: the next buffers come from other pages
        if gb:
            keep.append(torch.empty(gb << 30, dtype=torch.uint8, device=dev))
        a = torch.empty(E, device=dev)
        for i in range(64):
            a.view(-1)[i * (E // 64):(i + 1) * (E // 64)].copy_(src)
        acts.append(a)
        wss.append(torch.empty(wsb, dtype=torch.uint8, device=dev))
    grads = torch.empty(E, device=dev)
    live = bool(os.environ.get("RNNT_TUNE_LIVE"))
    variants = (os.environ.get("VARIANTS", "t2ord=0;t2ord=1;t2ord=2").split(";")) if live else [""]
    for rnd in range(2):
        for ia, a in enumerate(acts):
            for iw, w in enumerate(wss):
                rec = {"round": rnd, "acts": ia, "ws": iw, "acts_ptr": hex(a.data_ptr()), "ws_ptr": hex(w.data_ptr())}
                for v in variants:
                    if live:
                        os.environ["RNNT_TUNE"] = v
                    lib.rnnt_profile_reset(); lib.rnnt_profile_enable(1)
                    for i in range(4):
                        if i == 1:
                            torch.cuda.synchronize(); lib.rnnt_profile_collect(); lib.rnnt_profile_reset()
                        st = lib.compute_rnnt_loss_async(a.data_ptr(), grads.data_ptr(), lab.data_ptr(), ll.data_ptr(), tl.data_ptr(), A, N,
                                                         costs.data_ptr(), None, w.data_ptr(), opt, 0)
                        assert st == 0
                        torch.cuda.synchronize(); lib.rnnt_profile_collect()
                    ms = (C.c_double * 5)()
                    calls = lib.rnnt_profile_read(ms, 5)
                    lib.rnnt_profile_enable(0)
                    rec["stats_ms " + v] = round(ms[0] / calls, 4)
                    rec["grad_ms " + v] = round(ms[3] / calls, 4)
                rec["loss0"] = float(costs[0])
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "table":
        table(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "matrix":
        matrix()
    else:
        run()
