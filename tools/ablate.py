#!/usr/bin/env python3
"""Measurement aid (was NEMO_ABLATE inside the product loader until round 4):

    python tools/ablate.py name1,name2,... [bench.py arguments]

runs bench.py with the named C entry points of libnemo_hip.so turned into no-ops that return 0.  The step then computes
garbage -- the only meaningful output is its TIME: the difference to the full step is what the kernel contributes to the
un-profiled critical path (kernel traces over-state cross-queue latencies inside replayed graphs).
`nemo_gemm_f32@<M>x<N>x<K>` ablates one GEMM shape.  Nothing in nemo_cvpr2023_amd/ knows about this."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Ablated:
    def __init__(self, lib, names):
        self._lib, self._names = lib, set(names)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in self._names:
            return lambda *a, **k: 0
        shapes = {n.split('@')[1] for n in self._names if n.startswith(name + '@')}
        if shapes:
            def gated(*a, **k):
                return 0 if f'{a[2]}x{a[3]}x{a[4]}' in shapes else fn(*a, **k)
            return gated
        return fn


def main():
    names = [n for n in sys.argv[1].split(',') if n]
    from nemo_cvpr2023_amd import _lib
    real = _lib.load
    _lib.load_for_engine = lambda: Ablated(real(), names)
    import nemo_cvpr2023_amd.engine as engine          # (binds _lib.load_for_engine at call time)
    sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
    import bench
    bench.main()


if __name__ == '__main__':
    main()
