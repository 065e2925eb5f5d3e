#!/usr/bin/env python3
"""Does the LIBRARY's captured single-pass launch replay correctly when the state is zeroed by hipMemsetAsync (round 1)
instead of k_state_clear (round 2)?  Needs the experiment build:
    make -C disparity_to_point_cloud_amd/csrc variant NAME=memset DEFS=-DD2PC_CLEAR_WITH_MEMSET=1
    D2PC_LIBRARY_VARIANT=memset python tools/graph_memset_probe.py
Matrix: capture through torch.cuda.CUDAGraph or through raw HIP calls (ctypes) x host reads of the state header between
replays (d2pc_check_async_error) or none.  Prints per replay whether the counts came out and the timeout flag.
GPU box only."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import disparity_to_point_cloud_amd as d2pc  # noqa: E402
from disparity_to_point_cloud_amd.synth import synth_disparity  # noqa: E402
from disparity_to_point_cloud_amd.torch_api import DeviceBatch  # noqa: E402

print("library:", d2pc.capi.library_path())
hip = ctypes.CDLL("libamdhip64.so")
vp = ctypes.c_void_p


class MemsetParams(ctypes.Structure):   # hipMemsetParams (hip/driver_types.h)
    _fields_ = [("dst", ctypes.c_void_p), ("elementSize", ctypes.c_uint), ("height", ctypes.c_size_t),
                ("pitch", ctypes.c_size_t), ("value", ctypes.c_uint), ("width", ctypes.c_size_t)]


def memset_nodes(graph):
    """[(dst, bytes, value)] of the graph's memset nodes as the runtime recorded them (hipGraphMemsetNodeGetParams)."""
    n = ctypes.c_size_t(0)
    ck(hip.hipGraphGetNodes(graph, None, ctypes.byref(n)), "hipGraphGetNodes")
    nodes = (vp * n.value)()
    ck(hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)), "hipGraphGetNodes")
    out = []
    for nd in nodes:
        t = ctypes.c_int()
        ck(hip.hipGraphNodeGetType(vp(nd), ctypes.byref(t)), "hipGraphNodeGetType")
        if t.value == 2:   # hipGraphNodeTypeMemset
            p = MemsetParams()
            ck(hip.hipGraphMemsetNodeGetParams(vp(nd), ctypes.byref(p)), "hipGraphMemsetNodeGetParams")
            out.append((p.dst, p.width * p.elementSize * max(p.height, 1), p.value))
    return out, n.value


class CaptureStderr:
    """fd 2 into a file for the duration (the experiment build prints `d2pc: hipMemsetAsync(ptr, 0, bytes)` under D2PC_TRACE_MEMSET)."""
    def __enter__(self):
        import tempfile
        sys.stderr.flush()
        self.f = tempfile.TemporaryFile()
        self.saved = os.dup(2)
        os.dup2(self.f.fileno(), 2)
        return self

    def __exit__(self, *a):
        os.dup2(self.saved, 2)
        os.close(self.saved)
        self.f.seek(0)
        self.text = self.f.read().decode(errors="replace")
        self.f.close()


def ck(e, what):
    if e != 0:
        raise RuntimeError(f"{what}: hip error {e}")


def dirty(ctx):
    """(16-byte pieces of the state found non-zero by the verify kernel behind the memset, launches verified)"""
    buf = (ctypes.c_ulonglong * 1028)()  # CompactStats: 64 slots of 128 bytes, then launches, timeouts, dbg[2]
    L = d2pc.load_library()
    L.d2pc_debug_read_stats.argtypes = [vp, vp]
    L.d2pc_debug_read_stats(ctx.handle, buf)
    return f"[verify kernel: {buf[1026]} dirty pieces over {buf[1027]} launches]"


def flag(ctx):
    try:
        ctx.check_async_error()
        return "clear"
    except d2pc.D2pcError:
        return "TIMEOUT FLAG SET"


q = d2pc.make_q()
w, h, n = 640, 480, 6
frames = [synth_disparity(2, f, w, h, "holes") for f in range(n)]
for capture in ("torch", "torch_side", "hip", "hip_null"):
    for host_reads in (True, False):
        with d2pc.Context(q=q, mode=d2pc.MODE_COMPACT, compact_algo=2) as ctx:
            ctx.set_tuning("spin_timeout_ms", 100)
            b = DeviceBatch(ctx, n, h, w, want_index=True)
            b.disp.copy_(torch.from_numpy(np.stack(frames)))
            b.launch()
            torch.cuda.synchronize()
            want = b.counts.cpu().numpy().copy()
            if capture in ("torch", "torch_side"):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    b.launch()
                # torch replays on its CURRENT stream: the legacy default stream, or a side stream
                stream = torch.cuda.current_stream() if capture == "torch" else torch.cuda.Stream()
                replay = g.replay
            else:
                raw = vp()
                ck(hip.hipStreamCreateWithFlags(ctypes.byref(raw), 1), "hipStreamCreateWithFlags")  # non-blocking
                stream = torch.cuda.ExternalStream(raw.value)
                graph, gexec = vp(), vp()
                os.environ["D2PC_TRACE_MEMSET"] = "1"
                with CaptureStderr() as cap:
                    ck(hip.hipStreamBeginCapture(raw, 0), "hipStreamBeginCapture")  # hipStreamCaptureModeGlobal
                    b.launch(stream=stream)
                    ck(hip.hipStreamEndCapture(raw, ctypes.byref(graph)), "hipStreamEndCapture")
                os.environ.pop("D2PC_TRACE_MEMSET")
                import re
                lib_calls = [(int(m.group(1), 16), int(m.group(2))) for m in re.finditer(r"hipMemsetAsync\((0x[0-9a-f]+), 0, (\d+)\)", cap.text)]
                recorded, n_nodes = memset_nodes(graph)
                node_report = (f"library asked for {[(hex(p), n_) for p, n_ in lib_calls]}; graph of {n_nodes} nodes holds memset "
                               f"{[(hex(p or 0), n_, v) for p, n_, v in recorded]}: "
                               + ("MATCH" if [(p, n_) for p, n_, _ in recorded] == lib_calls and all(v == 0 for *_, v in recorded) else "DIFFER"))
                ck(hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_size_t(0)), "hipGraphInstantiate")

                if capture == "hip_null":   # captured on `raw`, launched on the legacy default stream
                    stream = torch.cuda.default_stream()

                    def replay():
                        ck(hip.hipGraphLaunch(gexec, None), "hipGraphLaunch")
                else:
                    def replay():
                        ck(hip.hipGraphLaunch(gexec, raw), "hipGraphLaunch")
            results = []
            with torch.cuda.stream(stream):
                for r in range(4):
                    b.counts.fill_(0)   # stream-ordered with the replay
                    replay()
                    if host_reads:
                        torch.cuda.synchronize()
                        ok = np.array_equal(b.counts.cpu().numpy(), want)
                        results.append(f"replay {r}: counts {'ok' if ok else 'WRONG'}, {flag(ctx)} {dirty(ctx)}")
            torch.cuda.synchronize()
            if not host_reads:
                ok = np.array_equal(b.counts.cpu().numpy(), want)
                results.append(f"after 4 replays without a host read: counts {'ok' if ok else 'WRONG'}, {flag(ctx)} {dirty(ctx)}")
            if capture in ("hip", "hip_null"):
                again, _ = memset_nodes(graph)   # ... and once more after the replays
                node_report += "; after the replays " + ("unchanged" if again == recorded else f"CHANGED to {again}")
                results.append(node_report)
            print(f"capture via {capture:10s}, host reads between replays {host_reads}: " + "; ".join(results), flush=True)
