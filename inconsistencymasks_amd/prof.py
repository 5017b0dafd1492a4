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

    def close(self):
        if self._p:
            lib.imk_prof_bind(None)
            lib.imk_prof_destroy(self._p)
            self._p = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
