"""Measurement context of include/imk.h (imk_prof_*): per-launch HIP-event timing of the library's kernel families, owned by
the caller and bound to the calling thread.  Used by bench.py and the probes; the product path never needs it."""
import ctypes

from ._lib import check, lib

N_FAMILIES = 18     # IMK_PROF_VARIANTS


class Profiler:
    def __init__(self, period=0):
        self._p = ctypes.c_void_p()
        check(lib.imk_prof_create(int(period), ctypes.byref(self._p)), "imk_prof_create")
        check(lib.imk_prof_bind(self._p), "imk_prof_bind")

    def set_period(self, period):
        """time every `period`-th hooked launch of this thread (0 = off)"""
        check(lib.imk_prof_set_period(self._p, int(period)), "imk_prof_set_period")

    def collect(self):
        """-> (launches, ms, algorithmic bytes, flops) per family; synchronises the recorded events and resets them"""
        c = (ctypes.c_int64 * N_FAMILIES)()
        ms, by, fl = ((ctypes.c_double * N_FAMILIES)() for _ in range(3))
        check(lib.imk_prof_collect(self._p, c, ms, by, fl), "imk_prof_collect")
        return [int(v) for v in c], [float(v) for v in ms], [float(v) for v in by], [float(v) for v in fl]

    def totals(self, on):
        """start (clearing) / stop summing every hooked launch per kernel name (imk_prof_totals_enable)"""
        check(lib.imk_prof_totals_enable(self._p, 1 if on else 0), "imk_prof_totals_enable")

    def totals_dump(self):
        """-> {kernel name: {"launches", "bytes", "flops"}} since totals(True)"""
        n = int(lib.imk_prof_totals_dump(self._p, None, 0))
        buf = ctypes.create_string_buffer(n + 16)
        lib.imk_prof_totals_dump(self._p, buf, n + 16)
        out = {}
        for line in buf.value.decode().splitlines():
            name, c, b, f = line.rsplit(";", 3)
            out[name] = {"launches": int(c), "bytes": float(b), "flops": float(f)}
        return out

    @staticmethod
    def mark(ident, stream=None):
        """marker dispatch (imk_mark_kernel, 64 * ident work-items) on `stream` (default: torch's current stream)"""
        import torch
        check(lib.imk_prof_mark(int(ident), ctypes.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)),
              "imk_prof_mark")

    def close(self):
        """Unbind (only if THIS context is the one bound to the calling thread) and destroy.  Call it on the thread that
        created the Profiler: a context still bound on another thread would dangle there."""
        if self._p:
            lib.imk_prof_unbind(self._p)
            lib.imk_prof_destroy(self._p)
            self._p = ctypes.c_void_p()

    def __del__(self):
        # a garbage-collected Profiler must not unbind whatever ANOTHER live Profiler has bound since (imk_prof_unbind compares);
        # destroy clears the binding of the destroying thread if it still points here
        try:
            self.close()
        except Exception:
            pass
