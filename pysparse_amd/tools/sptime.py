"""CPU-time stopwatch of the reference's scripts (pysparse/tools/sptime.py): user CPU seconds of this
process.  GPU work does not show up in it; the examples here print wall time next to it."""
import time

try:
    import resource

    def cputime():
        return resource.getrusage(resource.RUSAGE_SELF)[0]
except ImportError:  # no resource module on this platform
    def cputime():
        return time.process_time()
