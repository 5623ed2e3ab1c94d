"""Fill the code-object cache of the run-time instantiated kernels ahead of time (deployment step; no GPU needed):

    python -m gfdm_amd.precompile <timeslots> <subcarriers> <overlap> [--parts rx,ic,pre,mod,est] [more shapes: M K L ...]

Shapes that are compiled into the library need nothing and are reported as such; shapes only the generic family serves are refused.
See gfdm_hip_precompile / gfdm_hip_set_jit in include/gfdm_hip.h.
"""
import sys
import time

from . import capi

PARTS = {"rx": 1, "ic": 2, "pre": 4, "mod": 8, "est": 16}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    parts = 0
    if "--parts" in argv:
        i = argv.index("--parts")
        parts = sum(PARTS[p] for p in argv[i + 1].split(","))
        del argv[i:i + 2]
    if not argv or len(argv) % 3:
        print(__doc__)
        return 2
    rc = 0
    for i in range(0, len(argv), 3):
        M, K, L = (int(v) for v in argv[i:i + 3])
        t0 = time.perf_counter()
        try:
            capi.precompile(M, K, L, parts)
            print("timeslots %d subcarriers %d overlap %d: ready (%.1f s)" % (M, K, L, time.perf_counter() - t0))
        except capi.GfdmHipError as e:
            print("timeslots %d subcarriers %d overlap %d: %s" % (M, K, L, e))
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
